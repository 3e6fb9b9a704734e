"""render(): the Python drop-in boundary of the hot path (reference gaussian_renderer/__init__.py:24-127).
Same signature, same returned dict; packs camera + model tensors into a GaussianRasterizer call."""
import math

import torch

from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

_ZEROS = {}      # (P, device) -> zeros [P, 3]: "viewspace_points" of forward-only calls
_EMPTY = {}      # device -> 0-element tensor (an absent optional input of the native module)


def _tanfov(cam):
    """(tan(FoVx / 2), tan(FoVy / 2)) of a camera, remembered on it for as long as its field of view is what it was."""
    c = getattr(cam, "_hgs_tanfov", None)
    if c is None or c[0] != cam.FoVx or c[1] != cam.FoVy:
        c = (cam.FoVx, cam.FoVy, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5))
        try:
            cam._hgs_tanfov = c
        except AttributeError:
            pass
    return c[2], c[3]


def _render_forward_only(viewpoint_camera, pc, bg_color, scaling_modifier):
    """render() with gradients off and the default options -- a frame of an evaluation / viewer loop (reference render.py:60-82,
    utils/visualization.py:43): the same native call with the same arguments, without the objects that only autograd needs (the
    zero tensor that would receive dL/dmean2D is one shared tensor of zeros, no autograd.Function, no nn.Module per frame).
    Host time per frame was 0.115 ms for 0.067 ms of kernels (tools/dev/render_host_profile.py)."""
    from diff_gaussian_rasterization import _C
    if getattr(pc, "fused_geometry", False) and hasattr(pc, "endpoint_pairs") and pc._endpoints.is_cuda \
            and pc.endpoint_pairs.shape[0] > 0 and scaling_modifier == 1.0:
        return _render_strands_forward_only(viewpoint_camera, pc, bg_color, _C)
    xyz, scales_d, rotations_d = _geometry(pc)
    dev = xyz.device
    key = (xyz.shape[0], dev)
    zeros = _ZEROS.get(key)
    if zeros is None:
        if len(_ZEROS) > 8:
            _ZEROS.clear()
        zeros = _ZEROS[key] = torch.zeros_like(xyz)
    empty = _EMPTY.get(dev)
    if empty is None:
        empty = _EMPTY[dev] = torch.empty(0, device=dev)
    tanfovx, tanfovy = _tanfov(viewpoint_camera)
    _, color, radii, _, _, _ = _C.rasterize_gaussians_culled(
        bg_color, xyz, empty, pc.get_opacity, scales_d(), rotations_d(), scaling_modifier, empty,
        viewpoint_camera.world_view_transform, viewpoint_camera.full_proj_transform, tanfovx, tanfovy,
        int(viewpoint_camera.image_height), int(viewpoint_camera.image_width), pc.get_features, pc.active_sh_degree,
        viewpoint_camera.camera_center, False, False)
    return {"render": color, "viewspace_points": zeros, "visibility_filter": radii > 0, "radii": radii}


def _render_strands_forward_only(viewpoint_camera, pc, bg_color, _C):
    """The forward-only frame of a strand model: the segments' Gaussians (geometry, sigmoid of the opacity) are derived by the
    rasterizer's own first launch from the strand parameters (diff_gaussian_rasterization._C.HairSource: the launch
    gaussian_renderer.frames and the training iteration use -- the same bits as derived_gaussians() + get_opacity,
    tests/test_gpu_frames.py) instead of by three launches in front of it."""
    dev = pc._endpoints.device
    P = pc.endpoint_pairs.shape[0]
    f32 = dict(dtype=torch.float32, device=dev)
    buf = torch.empty((11 * P + 8,), **f32)            # xyz | scale | quat | opacity, each 16-byte aligned
    a = (3 * P + 3) // 4 * 4
    xyz, scale = buf[0:3 * P].view(P, 3), buf[a:a + 3 * P].view(P, 3)
    quat, opacity = buf[2 * a:2 * a + 4 * P].view(P, 4), buf[2 * a + 4 * P:2 * a + 5 * P].view(P, 1)
    key = (P, dev)
    zeros = _ZEROS.get(key)
    if zeros is None:
        if len(_ZEROS) > 8:
            _ZEROS.clear()
        zeros = _ZEROS[key] = torch.zeros((P, 3), **f32)
    empty = _EMPTY.get(dev)
    if empty is None:
        empty = _EMPTY[dev] = torch.empty(0, device=dev)
    tanfovx, tanfovy = _tanfov(viewpoint_camera)
    hair = _C.HairSource(pc._endpoints, pc.endpoint_pairs, pc._width, pc.dist_to_scale_factor, pc._opacity, pc._mask)
    # (get_features is cat(dc, rest): with no higher-order coefficients the DC tensor itself is that array)
    shs = pc._features_dc if pc._features_rest.shape[1] == 0 else pc.get_features
    _, color, radii, _, _, _ = _C.rasterize_gaussians_prezeroed(
        bg_color, xyz, empty, opacity, scale, quat, 1.0, empty, viewpoint_camera.world_view_transform,
        viewpoint_camera.full_proj_transform, tanfovx, tanfovy, int(viewpoint_camera.image_height),
        int(viewpoint_camera.image_width), shs, pc.active_sh_degree, viewpoint_camera.camera_center, None, None, hair)
    return {"render": color, "viewspace_points": zeros, "visibility_filter": radii > 0, "radii": radii}


def render(viewpoint_camera, pc, bg_color, scaling_modifier=1.0, override_color=None, debug=False,
           compute_cov3D_python=False, convert_SHs_python=False):
    """Render the scene.  `bg_color` must live on the GPU."""
    if not (torch.is_grad_enabled() or debug or compute_cov3D_python or convert_SHs_python or override_color is not None):
        return _render_forward_only(viewpoint_camera, pc, bg_color, scaling_modifier)
    xyz, scales_d, rotations_d = _geometry(pc)
    # zero tensor whose gradient receives dL/d(screen-space mean) (reference :41-50), read by the densification stats
    if torch.is_grad_enabled():
        screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device=xyz.device) + 0
        try:
            screenspace_points.retain_grad()
        except Exception:
            pass
    else:       # forward-only call: the same zeros, nothing to retain a gradient for
        screenspace_points = torch.zeros_like(xyz)
    tanfovx, tanfovy = _tanfov(viewpoint_camera)
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=tanfovx, tanfovy=tanfovy, bg=bg_color, scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform, projmatrix=viewpoint_camera.full_proj_transform,
        sh_degree=pc.active_sh_degree, campos=viewpoint_camera.camera_center, prefiltered=False, debug=debug)
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)

    scales = rotations = cov3D_precomp = None
    if compute_cov3D_python:
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales, rotations = scales_d(), rotations_d()

    shs = colors_precomp = None
    if override_color is None:
        if convert_SHs_python:
            from utils.sh import eval_sh
            feats = pc.get_features
            shs_view = feats.transpose(1, 2).reshape(-1, 3, (pc.max_sh_degree + 1) ** 2)
            dir_pp = xyz - viewpoint_camera.camera_center.repeat(feats.shape[0], 1)
            dir_pp = dir_pp / dir_pp.norm(dim=1, keepdim=True)
            colors_precomp = torch.clamp_min(eval_sh(pc.active_sh_degree, shs_view, dir_pp) + 0.5, 0.0)
        else:
            shs = pc.get_features
    else:
        colors_precomp = override_color

    rendered_image, radii = rasterizer(means3D=xyz, means2D=screenspace_points, shs=shs, colors_precomp=colors_precomp,
                                       opacities=pc.get_opacity, scales=scales, rotations=rotations,
                                       cov3D_precomp=cov3D_precomp)
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "radii": radii}


def _geometry(pc):
    """(means, scaling getter, rotation getter) of one parameter state.  A strand model derives all of them from its
    endpoints with one kernel (HairGaussianModel.derived_gaussians); the model itself keeps no derived state, so the
    bundle lives exactly as long as this render call (the reference recomputes per getter call,
    scene/hair_gaussian_model.py:134-172)."""
    bundle = pc.derived_gaussians() if hasattr(pc, "derived_gaussians") else None
    if bundle is None:
        return pc.get_xyz, (lambda: pc.get_scaling), (lambda: pc.get_rotation)
    return bundle[0], (lambda: bundle[1]), (lambda: bundle[2])


def render_multi(viewpoint_camera, pc, bg_color, extra4, scaling_modifier=1.0, debug=False, splits=(4,),
                 black_background=False):
    """One traversal for RGB + 4 extra per-Gaussian channels.  Equivalent to render(...) plus
    render(..., override_color=extra) on a black background, which is how the reference's mask and orientation losses
    obtain their images (loss/losses.py:247,312).  Returns render()'s dict + "extra": one tensor per entry of `splits`
    ((4,) -> a single [4,H,W] tensor; (1,3) -> ([H,W], [3,H,W])).  `black_background=True`: the caller states that bg_color
    is all zero (selects the backward's black-background specialisation; bg_color is not inspected)."""
    xyz, scales_d, rotations_d = _geometry(pc)
    screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device=xyz.device) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass
    # RGB background + black for the extra channels (the reference's default bg of those passes); built per call, so a
    # background that changes between calls (random_background) is the one rendered
    bg7 = torch.nn.functional.pad(bg_color.to(torch.float32), (0, 4))
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5), bg=bg7,
        scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform, sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center, prefiltered=False, debug=debug)
    out = GaussianRasterizer(raster_settings=raster_settings).forward_multi(
        means3D=xyz, means2D=screenspace_points, opacities=pc.get_opacity, extra4=extra4, shs=pc.get_features,
        scales=scales_d(), rotations=rotations_d(), splits=splits, black_background=black_background)
    rgb, radii, extras = out[0], out[1], out[2:]
    return {"render": rgb, "extra": extras[0] if len(extras) == 1 else extras, "viewspace_points": screenspace_points,
            "visibility_filter": radii > 0, "radii": radii}
