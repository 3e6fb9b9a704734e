"""MI355X drop-in for the reference's `diff_gaussian_rasterization` package
(submodules/diff-gaussian-rasterization/diff_gaussian_rasterization/__init__.py:21-220): same public names
(GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians), same argument validation and the same
autograd contract (gradients for means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
cov3Ds_precomp; means2D's gradient carries dL/dmean2D * (0.5W, 0.5H) for the densification statistics)."""
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C


def cpu_deep_copy_tuple(input_tuple):
    return tuple(item.cpu().clone() if isinstance(item, torch.Tensor) else item for item in input_tuple)


_EMPTY = torch.Tensor([])   # stands for an absent optional input (never written to)


class GaussianRasterizationSettings(NamedTuple):  # reference :157-169
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
        rs = raster_settings
        args = (rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh, rs.sh_degree,
                rs.campos, rs.prefiltered, rs.debug)
        if rs.debug:  # reference :83-90: snapshot the inputs so a failing call can be replayed
            cpu_args = cpu_deep_copy_tuple(args)
            try:
                num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer = _C.rasterize_gaussians_culled(*args)   # (callers of the autograd op see image, radii, gradients: unchanged by culling)
            except Exception:
                torch.save(cpu_args, "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise
        else:
            num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer = _C.rasterize_gaussians_culled(*args)   # (callers of the autograd op see image, radii, gradients: unchanged by culling)
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer,
                              binningBuffer, imgBuffer)
        ctx.mark_non_differentiable(radii)
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _):
        rs = ctx.raster_settings
        colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer, imgBuffer = \
            ctx.saved_tensors
        args = (rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color, sh, rs.sh_degree, rs.campos,
                geomBuffer, ctx.num_rendered, binningBuffer, imgBuffer, rs.debug)
        if rs.debug:
            cpu_args = cpu_deep_copy_tuple(args)
            try:
                grads = _C.rasterize_gaussians_backward(*args)
            except Exception:
                torch.save(cpu_args, "snapshot_bw.dump")
                print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                raise
        else:
            grads = _C.rasterize_gaussians_backward(*args)
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations) = grads

        def _m(g, ref):  # absent inputs were 0-element tensors: hand back a matching empty gradient
            return g if ref.numel() != 0 else None
        return (grad_means3D, grad_means2D, _m(grad_sh, sh), _m(grad_colors_precomp, colors_precomp), grad_opacities,
                _m(grad_scales, scales), _m(grad_rotations, rotations), _m(grad_cov3Ds_precomp, cov3Ds_precomp), None)


class _RasterizeGaussiansMulti(torch.autograd.Function):
    """Single-pass mode: RGB (from SH or precomputed colours) + 4 extra unclamped channels in one forward/backward.
    `raster_settings.bg` must have 7 entries.  Returns (rgb[3,H,W], extra[4,H,W], radii): two views of one 7-channel
    buffer, so the losses' gradients arrive as separate tensors and are handed to the kernel plane by plane (no
    zero-filled 7-channel gradient is ever assembled).  means2D's gradient is the RGB channels' screen-space gradient.
    `black_background=True` is the CALLER's statement that raster_settings.bg is all zero: the backward then runs the
    black-background specialisation (include/hgs.h hgs_backward_multi, bg == NULL); nothing here inspects the tensor."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, extra4, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings, splits, black_background=False):
        rs = raster_settings
        ctx.black_background = bool(black_background)
        num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer = _C.rasterize_gaussians_multi(
            rs.bg, means3D, colors_precomp, extra4, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
            rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh, rs.sh_degree,
            rs.campos, rs.prefiltered, rs.debug)
        ctx.raster_settings, ctx.num_rendered = rs, num_rendered
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer,
                              binningBuffer, imgBuffer)
        ctx.mark_non_differentiable(radii)
        ctx.hw = (color.shape[1], color.shape[2])
        ctx.splits = tuple(splits)
        assert sum(ctx.splits) == 4
        outs, c0 = [], 3
        for n in ctx.splits:  # the extra channels as separate (contiguous) views: their gradients arrive separately
            outs.append(color[c0] if n == 1 else color[c0:c0 + n])
            c0 += n
        return (color[:3], radii) + tuple(outs)

    @staticmethod
    def backward(ctx, g_rgb, _, *g_splits):
        rs = ctx.raster_settings
        colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer, imgBuffer = \
            ctx.saved_tensors
        H, W = ctx.hw
        dev = means3D.device
        g_rgb = torch.zeros((3, H, W), device=dev) if g_rgb is None else g_rgb.contiguous()
        planes = [g_rgb[0], g_rgb[1], g_rgb[2]]
        for n, g in zip(ctx.splits, g_splits):
            g = torch.zeros((n, H, W), device=dev) if g is None else g.contiguous().reshape(n, H, W)
            planes += [g[k] for k in range(n)]
        (g_means2D, g_colors, g_ex, g_opac, g_means3D, g_cov, g_sh, g_scales, g_rot) = \
            _C.rasterize_gaussians_multi_backward(None if ctx.black_background else rs.bg, means3D, radii, colors_precomp, scales, rotations,
                                                  rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix, rs.projmatrix,
                                                  rs.tanfovx, rs.tanfovy, planes, sh, rs.sh_degree, rs.campos,
                                                  geomBuffer, ctx.num_rendered, binningBuffer, imgBuffer, rs.debug)

        def _m(g, ref):
            return g if ref.numel() != 0 else None
        return (g_means3D, g_means2D, _m(g_sh, sh), _m(g_colors, colors_precomp), g_ex, g_opac, _m(g_scales, scales),
                _m(g_rot, rotations), _m(g_cov, cov3Ds_precomp), None, None, None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                     raster_settings)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        with torch.no_grad():
            rs = self.raster_settings
            return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        empty = _EMPTY
        shs = empty if shs is None else shs
        colors_precomp = empty if colors_precomp is None else colors_precomp
        scales = empty if scales is None else scales
        rotations = empty if rotations is None else rotations
        cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, rs)

    def forward_multi(self, means3D, means2D, opacities, extra4, shs=None, colors_precomp=None, scales=None,
                      rotations=None, cov3D_precomp=None, splits=(4,), black_background=False):
        """Single-pass mode: returns (rgb[3,H,W], radii, *extras) where the blended `extra4` [P,4] channels are handed
        back in groups of `splits` channels ((4,) -> one [4,H,W] tensor; (1,3) -> [H,W] and [3,H,W])."""
        if (shs is None) == (colors_precomp is None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        empty = _EMPTY
        return _RasterizeGaussiansMulti.apply(
            means3D, means2D, empty if shs is None else shs, empty if colors_precomp is None else colors_precomp, extra4,
            opacities, empty if scales is None else scales, empty if rotations is None else rotations,
            empty if cov3D_precomp is None else cov3D_precomp, self.raster_settings, tuple(splits), black_background)
