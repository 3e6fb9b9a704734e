"""render(): the Python drop-in boundary of the hot path (reference gaussian_renderer/__init__.py:24-127).
Same signature, same returned dict; packs camera + model tensors into a GaussianRasterizer call."""
import math

import torch

from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer


def render(viewpoint_camera, pc, bg_color, scaling_modifier=1.0, override_color=None, debug=False,
           compute_cov3D_python=False, convert_SHs_python=False):
    """Render the scene.  `bg_color` must live on the GPU."""
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
    tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
    tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)
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
