"""Deterministic synthetic scenes for tests and benchmarks: camera rigs as the reference's
dataset scripts build them (utils/camera.py generate_cameras; 90 degree FoV: focal = W/2), random Gaussian
clouds (Stage I) and random-walk strand sets (Stage III), plus self-rendered training targets."""
import math

import numpy as np
import torch

from scene.cameras import Camera
from utils.camera import generate_cameras
from utils.graphics import focal2fov

WORKLOADS = {
    # name: (kind, primitives, views, W, H)   -- BASELINE.json configs
    "north_star": ("strands", dict(n_strands=1000, n_seg=100), 32, 1920, 1080),   # 100k strand-Gaussians / 1080p / 32 views
    "c2": ("cloud", dict(P=50000), 16, 800, 800),
    "c3": ("strands", dict(n_strands=2000, n_seg=100), 32, 1920, 1080),
    "c4": ("strands", dict(n_strands=10000, n_seg=100, curly=True), 48, 1920, 1080),
    "c5": ("strands", dict(n_strands=5000, n_seg=100), 64, 1920, 1080),
    "tiny": ("strands", dict(n_strands=40, n_seg=30), 4, 256, 144),
}


def make_cameras(n_views, W, H, device="cuda", dist=0.5, anchor=(0.0, 0.0, 0.0)):
    """n_views-1 cameras on a circle of radius `dist` about the y axis + one top view; f = W/2 px."""
    # camera 1 as the reference's dataset scripts place it (scripts/parse_usc_hairsalon.py:172-177): y-up world,
    # camera at +z looking back at the anchor, OpenCV axes (y and z flipped)
    pose = np.eye(4)
    pose[:3, 3] = np.asarray(anchor) + np.array([0.0, 0.0, dist])
    pose[:3, 1:3] *= -1
    focal = W / 2.0
    _, Es = generate_cameras(n_views, H, W, cam_pose=pose, anchor_pos=np.asarray(anchor, float), offset=dist,
                             focal_length_px=focal)
    cams = []
    for uid, cid in enumerate(sorted(Es)):
        w2c = Es[cid]
        R = w2c[:3, :3].T  # camera-to-world rotation, as the COLMAP reader stores it
        T = w2c[:3, 3]
        cams.append(Camera(colmap_id=cid, R=R, T=T, FoVx=focal2fov(focal, W), FoVy=focal2fov(focal, H), image=None,
                           gt_alpha_mask=None, image_name=f"view_{cid:03d}", uid=uid, data_device=device,
                           image_width=W, image_height=H))
    return cams


def cameras_extent(cams):
    """1.1 x max distance of a camera centre from their mean (data/dataset_readers.py:57-78) = spatial_lr_scale."""
    c = torch.stack([cam.camera_center for cam in cams]).double()
    return float(1.1 * (c - c.mean(0, keepdim=True)).norm(dim=1).max())


def strand_polylines(n_strands, n_seg, seed=0, radius=0.10, step=0.0025, jitter_deg=5.0, curly=False):
    """[S, n_seg+1, 3] random-walk strands rooted on a sphere (USC-HairSalon-like: 100 vertices, mm-scale steps)."""
    rng = np.random.default_rng(seed)
    roots = rng.normal(size=(n_strands, 3))
    roots = radius * roots / np.linalg.norm(roots, axis=1, keepdims=True)
    d = roots / np.linalg.norm(roots, axis=1, keepdims=True)
    pts = [roots]
    phase = rng.uniform(0, 2 * np.pi, n_strands)
    for k in range(n_seg):
        d = d + rng.normal(size=d.shape) * math.radians(jitter_deg) + np.array([0.0, -0.02, 0.0])  # y-up world: gravity pulls towards -y
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        p = pts[-1] + step * d
        pts.append(p)
    pts = np.stack(pts, 1)
    if curly:  # helical offset, radius 5 mm, period 20 segments
        t = np.arange(n_seg + 1)[None, :] * (2 * np.pi / 20.0) + phase[:, None]
        pts = pts + 0.005 * np.stack([np.cos(t), np.zeros_like(t), np.sin(t)], -1) * np.minimum(1.0, np.arange(n_seg + 1) / 10.0)[None, :, None]
    return pts.astype(np.float32)


def make_strand_model(n_strands, n_seg, seed=0, device="cuda", spatial_lr_scale=1.0, curly=False, sh_degree=0):
    from scene.hair_gaussian_model import HairGaussianModel
    pts = strand_polylines(n_strands, n_seg, seed=seed, curly=curly)
    rng = np.random.default_rng(seed + 1)
    P = n_strands * n_seg
    hue = rng.uniform(0, 1, n_strands)
    col = np.stack([0.5 + 0.4 * np.cos(2 * np.pi * (hue + s)) for s in (0.0, 1 / 3, 2 / 3)], 1)
    col = np.repeat(col, n_seg, axis=0).astype(np.float32)
    opacity = rng.uniform(0.3, 0.95, (P, 1)).astype(np.float32)
    return HairGaussianModel.from_strands(pts, width=1e-4, opacity=opacity, mask_prob=0.9, colors=col,
                                          sh_degree=sh_degree, spatial_lr_scale=spatial_lr_scale, device=device)


def make_cloud_model(P, seed=0, device="cuda", spatial_lr_scale=1.0, radius=0.12, sh_degree=0):
    """Stage-I initial state exactly as create_from_pcd builds it (distCUDA2 scales, opacity 0.1, mask 0.5)."""
    from scene.gaussian_model import GaussianModel
    from utils.graphics import BasicPointCloud
    rng = np.random.default_rng(seed)
    v = rng.normal(size=(P, 3))
    v = v / np.linalg.norm(v, axis=1, keepdims=True) * radius * rng.uniform(0, 1, (P, 1)) ** (1 / 3)
    m = GaussianModel(sh_degree=sh_degree, spatial_lr_scale=spatial_lr_scale, device=device)
    m.create_from_pcd(BasicPointCloud(points=v.astype(np.float32), colors=rng.uniform(0, 1, (P, 3)).astype(np.float32),
                                      normals=np.zeros((P, 3), np.float32)))
    return m


def perturbed_copy(model, seed=0, perturb=0.001):
    """A shallow copy of the model whose positions (strand endpoints / cloud centres) are moved by N(0, perturb): the
    synthetic ground truth.  The model itself is not touched (no `.data` writes on its parameters)."""
    import copy
    g = torch.Generator(device="cpu").manual_seed(seed)
    pos_attr = "_endpoints" if hasattr(model, "_endpoints") and model._endpoints.numel() else "_xyz"
    pos = getattr(model, pos_attr)
    gt = copy.copy(model)
    setattr(gt, pos_attr, (pos.detach() + torch.randn(pos.shape, generator=g).to(pos.device) * perturb).requires_grad_(False))
    return gt, g


def orientation_angles(omap, view, min_val):
    """World-space direction image [3,H,W] -> angle in [0, pi) w.r.t. the image y axis, the map the orientation term
    compares with the target field (reference loss/losses.py:250-268)."""
    o = omap.permute(1, 2, 0).reshape(-1, 3)
    pix = (o @ view[:3, :3])[:, :2]
    pix = pix / (torch.norm(pix, dim=1, keepdim=True) + min_val)
    x, y = pix[:, 0], pix[:, 1]
    y = torch.where(y < min_val, y + min_val, y)
    theta = torch.atan2(x, y)
    return torch.where(theta < 0, theta + math.pi, theta).reshape(omap.shape[1], omap.shape[2])


@torch.no_grad()
def attach_targets(cams, model, seed=0, perturb=0.001, consistent=False):
    """GT image = render of a perturbed copy of the model, GT mask = (alpha > 0), orientation ~ U[0,pi),
    confidence ~ U[0,1] (SURVEY.md 8d 'Targets for the training step').
    consistent=True: the orientation field is the projected direction of the GROUND-TRUTH strands (the perturbed copy's
    blended direction image turned into angles exactly as the loss does) with confidence 1 -- targets a model can
    converge to, for training-curve PSNR (SURVEY.md 8d 'PSNR')."""
    from gaussian_renderer import render
    dev = model.get_xyz.device
    gt, g = perturbed_copy(model, seed=seed, perturb=perturb)
    bg = torch.zeros(3, device=dev)
    ones = torch.ones((model.get_xyz.shape[0], 3), device=dev)
    for cam in cams:
        H, W = cam.image_height, cam.image_width
        cam.original_image = render(cam, gt, bg)["render"].clamp(0, 1).detach()
        alpha = render(cam, gt, bg, override_color=ones)["render"][0]
        cam.mask = alpha > 0
        cam.float_mask = cam.mask.to(torch.float32)
        if consistent and hasattr(gt, "get_orientation"):
            omap = render(cam, gt, bg, override_color=gt.get_orientation)["render"]
            cam.orientation_field = orientation_angles(omap, cam.world_view_transform, gt.min_val).contiguous()
            cam.orientation_confidence = torch.ones((H, W), device=dev)
        else:
            cam.orientation_field = (torch.rand((H, W), generator=g) * math.pi).to(dev)
            cam.orientation_confidence = torch.rand((H, W), generator=g).to(dev)


def build_workload(name, device="cuda", seed=0, with_targets=True, n_views=None, consistent=False):
    kind, kw, views, W, H = WORKLOADS[name]
    views = views if n_views is None else n_views
    cams = make_cameras(views, W, H, device=device)
    extent = cameras_extent(cams)
    if kind == "strands":
        model = make_strand_model(seed=seed, device=device, spatial_lr_scale=extent, **kw)
        model.compute_strands_info(only_foreground=True)
    else:
        model = make_cloud_model(seed=seed, device=device, spatial_lr_scale=extent, **kw)
    if with_targets:
        attach_targets(cams, model, seed=seed, consistent=consistent)
    return model, cams, extent


# ---- the three-stage workflow on ONE synthetic capture (reference README.md:136-153: train.py -> merge.py -> train.py) ------
# name: (ground-truth strands, curly, views, W, H).  c3_capture = BASELINE config 3's frame and views with 200 k ground-truth
# segments; c4_capture = BASELINE config 4 as written ("Cem-Yuksel 'curly' full 3-stage, ~1M Gaussians, 48 views @ 1080p").
CAPTURES = {
    "c3_capture": (2000, False, 32, 1920, 1080),
    "c4_capture": (10000, True, 48, 1920, 1080),
    "small_capture": (500, False, 16, 800, 800),
    "tiny_capture": (30, False, 4, 256, 144),
}


def build_capture(name_or_spec, device="cuda", seed=0, n_seg=100):
    """Ground-truth strands rendered to image / mask / orientation targets for every view from the ground-truth strand model
    itself (consistent targets).  Returns (gt_pts [S, n_seg+1, 3], gt_model, cams, extent)."""
    S, curly, views, W, H = CAPTURES[name_or_spec] if isinstance(name_or_spec, str) else name_or_spec
    gt_pts = strand_polylines(S, n_seg, seed=seed, curly=curly)
    cams = make_cameras(views, W, H, device=device)
    extent = cameras_extent(cams)
    gt_model = make_strand_model(S, n_seg, seed=seed, device=device, spatial_lr_scale=extent, curly=curly)
    attach_targets(cams, gt_model, seed=seed, perturb=0.0, consistent=True)
    return gt_pts, gt_model, cams, extent


def stage1_cloud(gt_pts, gt_model, extent, device="cuda", seed=1, jitter=0.002, colour_noise=0.1):
    """The Stage-I initial state of a capture: a Gaussian cloud of as many points as the ground truth has segments -- the
    segments' midpoints + N(0, jitter), colours of the ground truth + noise: what a sparse reconstruction hands train.py --
    through create_from_pcd (distCUDA2 scales, opacity 0.1, mask 0.5)."""
    from scene.gaussian_model import GaussianModel
    from utils.graphics import BasicPointCloud
    from utils.sh import SH2RGB
    rng = np.random.default_rng(seed)
    mid = 0.5 * (gt_pts[:, 1:] + gt_pts[:, :-1]).reshape(-1, 3)
    pts = (mid + rng.normal(size=mid.shape) * jitter).astype(np.float32)
    with torch.no_grad():
        gt_rgb = SH2RGB(gt_model._features_dc.detach()[:, 0]).clamp(0, 1).cpu().numpy()
    cloud = GaussianModel(sh_degree=0, spatial_lr_scale=extent, device=device)
    cloud.create_from_pcd(BasicPointCloud(points=pts, colors=np.clip(gt_rgb + rng.normal(size=gt_rgb.shape) * colour_noise, 0, 1).astype(np.float32),
                                          normals=np.zeros_like(pts)))
    cloud.ref_strand_root = gt_pts[:, 0].astype(np.float64)
    return cloud


# The states the pipeline lives in, as bench.py workloads ("what's missing" 2): built the way
# tools/three_stage.py builds them.
#   stage1_1080p   the Stage-I cloud of c3_capture (200 k Gaussians from jittered midpoints) after `stage1_iters` iterations of the
#                  reference's Stage-I loop (densification from 500, every 100)
#   stage3_merged  the Stage-II product of the same capture: Stage I for `stage1_iters` iterations, to_hair_gaussian_model, merge
#                  rounds to the fixed point (merge.merge_rounds) -- the model Stage III starts from
PIPELINE_STATES = {"stage1_1080p": ("c3_capture", "cloud"), "stage3_merged": ("c3_capture", "merged"),
                   # BASELINE config 4's Stage I (10^6 Gaussians from jittered midpoints of curly strands): bench.py --workload stage1_c4
                   # --stage1-iters 0 times its first iterations; not a leg of the default bench line
                   "stage1_c4": ("c4_capture", "cloud")}


def build_pipeline_state(name, device="cuda", seed=0, stage1_iters=None, n_views=None, log=None):
    """(model, cams, extent, info) of a PIPELINE_STATES entry; the model comes with training_setup() done for the stage it is
    in.  stage1_iters: iterations of the Stage-I loop before the state is taken (default: 1000 for stage1_1080p -- five
    densification events in --, 5000 for stage3_merged, the length tools/three_stage.py runs)."""
    import time
    from arguments import OptimizationParams
    from train import training
    from utils.general import safe_state
    capture, kind = PIPELINE_STATES[name]
    spec = CAPTURES[capture]
    if n_views is not None:
        spec = spec[:2] + (n_views,) + spec[3:]
    safe_state(True)
    gt_pts, gt_model, cams, extent = build_capture(spec, device=device, seed=seed)
    cloud = stage1_cloud(gt_pts, gt_model, extent, device=device)
    n0 = int(cloud.get_xyz.shape[0])
    del gt_model
    n1 = int(stage1_iters if stage1_iters is not None else (1000 if kind == "cloud" else 5000))
    opt1 = OptimizationParams()
    opt1.iterations = max(n1, 1)
    opt1._finalise()
    cloud.training_setup(opt1)
    events = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if n1 > 0:
        training(cloud, cams, opt1, iterations=n1, extent=extent, start_iteration=0, event_log=events)
    torch.cuda.synchronize()
    t1 = time.perf_counter() - t0
    info = {"capture": capture, "ground_truth_segments": int(gt_pts.shape[0] * (gt_pts.shape[1] - 1)), "stage1_start_gaussians": n0,
            "stage1_iterations": n1, "stage1_seconds": t1, "stage1_whole_loop_iters_per_sec": (n1 / t1 if n1 else None),
            "stage1_end_gaussians": int(cloud.get_xyz.shape[0]), "stage1_events": len(events)}
    if log is not None:
        log("stage I", info)
    if kind == "cloud":
        return cloud, cams, extent, info
    from merge import merge_rounds
    t0 = time.perf_counter()
    hair = cloud.to_hair_gaussian_model()
    n_before = int(hair.strands_info.n_strands)
    rounds = merge_rounds(hair, 100, log=(lambda *a: None))
    torch.cuda.synchronize()
    info.update(stage2_seconds=time.perf_counter() - t0, merge_rounds=rounds, strands_before=n_before,
                strands_after=int(hair.strands_info.n_strands), segments=int(hair.get_xyz.shape[0]))
    del cloud
    opt3 = OptimizationParams()
    hair.training_setup(opt3)
    if log is not None:
        log("stage II", info)
    return hair, cams, extent, info
