"""Training losses (counterparts of the reference's loss/losses.py:16-355).  The loss math is stock PyTorch; what
matters for the hot path is that mask_loss_rast and orientation_loss_rast each run one more rasterizer
forward+backward through render(override_color=...), so one training iteration = 3 raster passes."""
import math

import numpy as np
import torch
import torch.nn.functional as F

from gaussian_renderer import render
from scene.hair_gaussian_model import HairGaussianModel

_WINDOWS = {}
fused_losses = True  # GPU: fused HIP kernels for L1+SSIM and the orientation loss (hgs_runtime.fused); False: torch ops


def l1_loss(network_output, gt):
    return torch.abs(network_output - gt).mean()


def l2_loss(network_output, gt):
    return ((network_output - gt) ** 2).mean()


def gaussian(window_size, sigma):
    g = torch.tensor([math.exp(-((x - window_size // 2) ** 2) / float(2 * sigma ** 2)) for x in range(window_size)])
    return g / g.sum()


def create_window(window_size, channel):
    w1 = gaussian(window_size, 1.5).unsqueeze(1)
    return (w1 @ w1.t()).float()[None, None].expand(channel, 1, window_size, window_size).contiguous()


def _window(window_size, channel, like):
    key = (window_size, channel, like.device, like.dtype)
    if key not in _WINDOWS:  # the reference rebuilds + uploads the 11x11 window every call (losses.py:78-82)
        _WINDOWS[key] = create_window(window_size, channel).to(device=like.device, dtype=like.dtype)
    return _WINDOWS[key]


def ssim(img1, img2, window_size=11, size_average=True):
    """Gaussian-window SSIM exactly as losses.py:43-84 (5 grouped 11x11 convolutions, C1=0.01^2, C2=0.03^2)."""
    channel = img1.size(-3)
    window = _window(window_size, channel, img1)
    pad = window_size // 2
    mu1 = F.conv2d(img1, window, padding=pad, groups=channel)
    mu2 = F.conv2d(img2, window, padding=pad, groups=channel)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = F.conv2d(img1 * img1, window, padding=pad, groups=channel) - mu1_sq
    sigma2_sq = F.conv2d(img2 * img2, window, padding=pad, groups=channel) - mu2_sq
    sigma12 = F.conv2d(img1 * img2, window, padding=pad, groups=channel) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    return ssim_map.mean() if size_average else ssim_map.mean(1).mean(1).mean(1)


def bidirectional_angle_difference(angle1, angle2):
    """min(|a1-a2|, pi-|a1-a2|) for line orientations in [0, pi)."""
    half_pi = np.pi / 2
    ab = torch.abs if isinstance(angle1, torch.Tensor) else np.abs
    return half_pi - ab(ab(angle1 - angle2) - half_pi)


def angle_smoothness_loss(gaussians: HairGaussianModel, threshold: float = 30, eps: float = 1e-6):
    """Mean squared bending angle over consecutive strand segments whose angle exceeds `threshold` degrees
    (losses.py:175-221).  The index table comes from the model's cache instead of a per-step CPU rebuild."""
    cos_th = np.cos(threshold * np.pi / 180)
    idx = gaussians.smoothness_index_pairs()
    if idx.shape[0] == 0:
        return 0
    if fused_losses and gaussians._endpoints.is_cuda:
        from hgs_runtime.fused import smoothness_loss
        return smoothness_loss(gaussians._endpoints, idx, float(cos_th), float(eps))
    # op-by-op form (CPU tensors, or fused_losses = False: what the fused kernel is tested against): the reference's statements
    # as they stand, boolean indexing included -- no selected pair is the python number 0 (no gradient), and a pair with a
    # zero-length segment (0 / 0 direction) is never selected but, like in the reference, turns the gradient of a non-empty
    # selection into NaN through the normalisation's backward (tests/test_ref_loss_pins.py; the fused kernel skips such pairs)
    pos = gaussians._endpoints[idx]                      # (N, 2, 2, 3)
    d = pos[:, :, 1] - pos[:, :, 0]
    d = d / torch.norm(d, dim=2, keepdim=True)
    dot = torch.sum(d[:, 0] * d[:, 1], dim=1)
    if dot.is_cuda:
        # fused_losses = False on the GPU (the op-by-op iteration, which may be captured in a graph): the same selection without
        # boolean indexing -- no host synchronisation, no data-dependent shape, always a tensor (0 when no pair is selected).
        # Same value and gradient as the statements below wherever those are finite; with a zero-length segment in a strand the
        # reference's form gives NaN gradients (0 / 0 through the normalisation's backward), and so does this one -- unlike the
        # fused kernel, which skips such pairs (tests/test_ref_loss_pins.py).
        sel = dot <= cos_th
        ang2 = torch.acos(torch.clamp(dot, -1 + eps, 1 - eps)) ** 2
        return torch.where(sel, ang2, torch.zeros_like(ang2)).sum() / sel.sum().clamp(min=1)
    dot = dot[dot <= cos_th]                             # only bends sharper than the threshold are penalised
    if dot.shape[0] == 0:
        return 0
    return torch.mean(torch.acos(torch.clamp(dot, -1 + eps, 1 - eps)) ** 2)


def knn3_self(points):
    """(squared distances [n,3], indices [n,3]) of the 3 nearest points of `points` within the set itself, ascending, the
    point itself included -- what pytorch3d.ops.knn_points(p, p, K=3, return_sorted=True) returns for one cloud
    (reference loss/losses.py:139-144).  Indices only (no autograd): GPU tensors go through hgs_knn3, CPU tensors through
    a chunked distance matrix.  Fewer than 3 points: index -1, distance +inf."""
    pts = points.detach().to(torch.float32).contiguous()
    n = pts.shape[0]
    if pts.is_cuda:
        import hgs_runtime as rt
        idx = torch.empty((n, 3), dtype=torch.int32, device=pts.device)
        d2 = torch.empty((n, 3), dtype=torch.float32, device=pts.device)
        with torch.cuda.device(pts.device):
            rt.check(rt.lib().hgs_knn3(rt.current_stream(), n, rt.ptr(pts), rt.ptr(idx), rt.ptr(d2)))
        return d2, idx.to(torch.long)
    d2 = torch.full((n, 3), float("inf"))
    idx = torch.full((n, 3), -1, dtype=torch.long)
    k = min(3, n)
    for s in range(0, n, 4096):
        diff = pts[s:s + 4096, None, :] - pts[None, :, :]
        dist = diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1] + diff[..., 2] * diff[..., 2]
        # ascending by (distance, index): a stable sort of the distances keeps equal ones in index order
        order = torch.sort(dist, dim=1, stable=True)
        d2[s:s + 4096, :k], idx[s:s + 4096, :k] = order.values[:, :k], order.indices[:, :k]
    return d2, idx


def strand_joints_magnet_loss(gaussians: HairGaussianModel):
    """Pulls every strand end towards the nearest end of another strand (reference loss/losses.py:106-172; enabled by
    --lambda_magnet, 0 by default): mean over the strand ends of the FOURTH power of the distance to that neighbour
    (the reference squares the squared distance, :170).  Follows the reference statement by statement, including what
    looks unintended there: the second neighbour is compared with the GLOBAL id of the end's segment partner although the
    neighbour indices are positions in the list of ends (:150-152), and the neighbour's direction is looked up with those
    positions in the global endpoint table (:159-161) -- it only feeds the validity mask.  The neighbour search runs
    without autograd; the distances are then formed differentiably from the selected pairs, which gives knn_points'
    gradient (2 (p - q) to the query, -2 (p - q) to the neighbour)."""
    ep = gaussians._endpoints
    u, c = torch.unique(gaussians.endpoint_pairs, return_counts=True)
    ends = u[c == 1]
    comp, _ = gaussians.get_complementary_endpoint_idx(ends)
    mapping = torch.zeros(ep.shape[0], device=ep.device, dtype=torch.long)
    mapping[ends] = comp
    det = ep.detach()
    self_dir = det[ends] - det[comp]
    valid = torch.norm(self_dir, dim=1) > gaussians.min_val                 # collapsed end segments are left out
    self_dir, ends, comp = self_dir[valid], ends[valid], comp[valid]
    pts = ep[ends]
    n = pts.shape[0]
    if n < 3:
        return pts.sum() * 0.0
    _, nn = knn3_self(pts)
    # (an end whose coordinates are not finite compares closer to nothing: its neighbour slots stay -1, which would index
    # the LAST end below instead of failing -- such rows, and rows whose neighbour is one, are left out of the mean)
    found = (nn >= 0).all(dim=1)
    nn = nn.clamp(min=0)
    sq = ((pts[:, None, :] - pts[nn]) ** 2).sum(dim=-1)                     # [n, 3], differentiable
    self_idx = torch.arange(n, device=ep.device)
    second_ok = (nn[:, 1] != self_idx) & (nn[:, 1] != comp)
    sq = torch.where(second_ok, sq[:, 1], sq[:, 2])
    nn = torch.where(second_ok, nn[:, 1], nn[:, 2])
    self_mask = torch.norm(self_dir, dim=1, keepdim=True) > gaussians.min_val
    nn_dir = det[nn] - det[mapping[nn]]
    nn_mask = torch.norm(nn_dir, dim=1, keepdim=True) > gaussians.min_val
    final = (self_mask & nn_mask).reshape(-1) & found & torch.isfinite(sq.detach())
    sq = sq[final]
    if sq.numel() == 0:
        return pts.sum() * 0.0
    return torch.mean(sq * sq)


_BLACK = {}
_BG_HOST = {}


def _black(device):
    """Black background tensor, one per device (the reference allocates a fresh one per call, losses.py:228,296)."""
    key = str(device)
    if key not in _BLACK:
        _BLACK[key] = torch.zeros(3, dtype=torch.float32, device=device)
    return _BLACK[key]


def _bg_host(bg, needed=True):
    """Host copy of a background colour for the fused orientation kernel, which uses it only when the camera has no mask
    (mask = any(o != bg), losses.py:277).  Read back once per tensor OBJECT and version: the entry keeps the tensor alive,
    so the id cannot be handed to another tensor while the entry exists."""
    if not needed:
        return [0.0, 0.0, 0.0]
    hit = _BG_HOST.get(id(bg))
    if hit is None or hit[0] is not bg or hit[1] != bg._version:
        if len(_BG_HOST) > 16:
            _BG_HOST.clear()
        hit = _BG_HOST[id(bg)] = (bg, bg._version, [float(x) for x in bg.detach().cpu()])
    return hit[2]


def orientation_loss_rast(gaussians, camera, args, bg=None):
    """Render world-space segment directions, rotate to view space, convert to an angle in [0, pi) w.r.t. the image
    y axis and compare bidirectionally with the GT orientation field, confidence-weighted (losses.py:224-289).
    On the GPU everything after the render is one fused kernel pair (hgs_orientation_loss_*); `fused_losses=False`
    or CPU tensors run the op-by-op restatement below (masked mean written as sum/count: no host sync)."""
    bg = _black(gaussians.get_xyz.device) if bg is None else bg
    omap = render(camera, gaussians, bg, override_color=gaussians.get_orientation)["render"]      # [3,H,W]
    if fused_losses and omap.is_cuda:
        from hgs_runtime.fused import orientation_loss
        bg3 = _bg_host(bg, camera.mask is None)
        return orientation_loss(omap, camera.world_view_transform, bg3, gaussians.min_val, camera.orientation_field,
                                camera.orientation_confidence, camera.mask)
    omap = omap.permute(1, 2, 0)
    h, w = omap.shape[:2]
    pix = (omap.flatten(0, 1) @ camera.world_view_transform[:3, :3])[:, :2]
    pix = pix / (torch.norm(pix, dim=1, keepdim=True) + gaussians.min_val)
    x, y = pix[:, 0], pix[:, 1]
    y = torch.where(y < gaussians.min_val, y + gaussians.min_val, y)
    theta = torch.atan2(x, y)
    theta = torch.where(theta < 0, theta + np.pi, theta).reshape(h, w)
    mask = torch.any(omap != bg, dim=2) if camera.mask is None else camera.mask
    diff = bidirectional_angle_difference(theta, camera.orientation_field) * camera.orientation_confidence
    m = mask.to(diff.dtype)
    return (diff * m).sum() / m.sum()


def mask_loss_rast(gaussians, camera, args, bg=None):
    """BCE-with-logits between the rasterized per-Gaussian mask value and the GT mask (losses.py:292-316)."""
    bg = _black(gaussians.get_xyz.device) if bg is None else bg
    rendered = render(camera, gaussians, bg, override_color=gaussians.get_mask.repeat(1, 3))["render"][0]
    return F.binary_cross_entropy_with_logits(rendered, camera.float_mask)


def _orientation_term(omap, gaussians, camera, bg):
    """Everything of orientation_loss_rast after the render, on a [3,H,W] direction image."""
    if fused_losses and omap.is_cuda:
        from hgs_runtime.fused import orientation_loss
        return orientation_loss(omap, camera.world_view_transform, _bg_host(bg, camera.mask is None), gaussians.min_val,
                                camera.orientation_field, camera.orientation_confidence, camera.mask)
    o = omap.permute(1, 2, 0)
    h, w = o.shape[:2]
    pix = (o.flatten(0, 1) @ camera.world_view_transform[:3, :3])[:, :2]
    pix = pix / (torch.norm(pix, dim=1, keepdim=True) + gaussians.min_val)
    x, y = pix[:, 0], pix[:, 1]
    y = torch.where(y < gaussians.min_val, y + gaussians.min_val, y)
    theta = torch.atan2(x, y)
    theta = torch.where(theta < 0, theta + np.pi, theta).reshape(h, w)
    mask = torch.any(o != bg, dim=2) if camera.mask is None else camera.mask
    diff = bidirectional_angle_difference(theta, camera.orientation_field) * camera.orientation_confidence
    m = mask.to(diff.dtype)
    return (diff * m).sum() / m.sum()


def loss_function_single_pass(gaussians, viewpoint_cam, args, bg, black_background=False):
    """loss_function with ONE rasterizer traversal: RGB, the mask value and the world-space direction are blended
    together (gaussian_renderer.render_multi) instead of three render() calls.  Same terms, same weights; returns
    (loss, terms, render_pkg) where render_pkg is what render() would have returned for the RGB pass."""
    from gaussian_renderer import render_multi
    extra = torch.cat((gaussians.get_mask, gaussians.get_orientation), dim=1)      # [P, 1 + 3]
    pkg = render_multi(viewpoint_cam, gaussians, bg, extra, splits=(1, 3), black_background=black_background)
    image, (mask_img, omap) = pkg["render"], pkg["extra"]
    gt = viewpoint_cam.original_image
    if fused_losses and image.is_cuda:
        from hgs_runtime.fused import ssim_l1
        ssim_mean, l1_mean = ssim_l1(image, gt)
        terms = {"l1": l1_mean, "dssim": 1.0 - ssim_mean}
    else:
        terms = {"l1": l1_loss(image, gt), "dssim": 1.0 - ssim(image, gt)}
    loss = max(0, 1.0 - args.lambda_dssim) * terms["l1"] + args.lambda_dssim * terms["dssim"]
    black = _black(image.device)
    if args.lambda_mask > 0 and viewpoint_cam.mask is not None:
        terms["mask"] = F.binary_cross_entropy_with_logits(mask_img, viewpoint_cam.float_mask)
        loss = loss + args.lambda_mask * terms["mask"]
    if args.lambda_orientation > 0:
        terms["orientation"] = _orientation_term(omap, gaussians, viewpoint_cam, black)
        loss = loss + args.lambda_orientation * terms["orientation"]
    if isinstance(gaussians, HairGaussianModel):
        if args.lambda_smooth > 0:
            terms["smooth"] = angle_smoothness_loss(gaussians)
            loss = loss + args.lambda_smooth * terms["smooth"]
        if args.lambda_magnet > 0:
            terms["magnet"] = strand_joints_magnet_loss(gaussians)
            loss = loss + args.lambda_magnet * terms["magnet"]
    return loss, terms, pkg


def loss_function(gaussians, image, viewpoint_cam, args):
    """(1-l)L1 + l(1-SSIM) + l_mask BCE + l_orient orientation [+ l_smooth smoothness] (losses.py:319-355)."""
    gt = viewpoint_cam.original_image
    if fused_losses and image.is_cuda:
        from hgs_runtime.fused import ssim_l1
        ssim_mean, l1_mean = ssim_l1(image, gt)
        terms = {"l1": l1_mean, "dssim": 1.0 - ssim_mean}
    else:
        terms = {"l1": l1_loss(image, gt), "dssim": 1.0 - ssim(image, gt)}
    loss = max(0, 1.0 - args.lambda_dssim) * terms["l1"] + args.lambda_dssim * terms["dssim"]
    if args.lambda_mask > 0 and viewpoint_cam.mask is not None:
        terms["mask"] = mask_loss_rast(gaussians, viewpoint_cam, args)
        loss = loss + args.lambda_mask * terms["mask"]
    if args.lambda_orientation > 0:
        terms["orientation"] = orientation_loss_rast(gaussians, viewpoint_cam, args)
        loss = loss + args.lambda_orientation * terms["orientation"]
    if isinstance(gaussians, HairGaussianModel):
        if args.lambda_smooth > 0:
            terms["smooth"] = angle_smoothness_loss(gaussians)
            loss = loss + args.lambda_smooth * terms["smooth"]
        if args.lambda_magnet > 0:
            terms["magnet"] = strand_joints_magnet_loss(gaussians)
            loss = loss + args.lambda_magnet * terms["magnet"]
    return loss, terms
