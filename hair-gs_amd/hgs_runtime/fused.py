"""torch.autograd front-ends of the fused HIP kernels that replace PyTorch-side chains on the training-step path
(include/hgs.h: hgs_strand_geometry_*, hgs_ssim_l1_*, hgs_orientation_loss_*).  GPU tensors only."""
import ctypes as C
import math

import torch

import hgs_runtime as rt

_WINDOW = None


def gaussian_window11():
    """The reference's 1-D window: fp32 exp(-(x-5)^2 / (2*1.5^2)), normalised in fp32 (loss/losses.py:24-31)."""
    global _WINDOW
    if _WINDOW is None:
        g = torch.tensor([math.exp(-((x - 5) ** 2) / float(2 * 1.5 ** 2)) for x in range(11)])
        g = (g / g.sum()).to(torch.float32)
        _WINDOW = (C.c_float * 11)(*[float(v) for v in g])
    return _WINDOW


class _StrandGeometry(torch.autograd.Function):
    @staticmethod
    def forward(ctx, endpoints, width, pairs, factor, owner):
        endpoints = rt.require_gpu_tensor(endpoints, "endpoints", torch.float32)
        width = rt.require_gpu_tensor(width, "width", torch.float32)
        pairs = rt.require_gpu_tensor(pairs, "endpoint_pairs", torch.int64)
        P, dev = pairs.shape[0], endpoints.device
        xyz = torch.empty((P, 3), dtype=torch.float32, device=dev)
        scale = torch.empty((P, 3), dtype=torch.float32, device=dev)
        quat = torch.empty((P, 4), dtype=torch.float32, device=dev)
        direction = torch.empty((P, 3), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rt.check(rt.lib().hgs_strand_geometry_forward(rt.current_stream(), P, rt.ptr(endpoints), rt.ptr(pairs),
                                                          rt.ptr(width), float(factor), rt.ptr(xyz), rt.ptr(scale),
                                                          rt.ptr(quat), rt.ptr(direction)))
        ctx.save_for_backward(endpoints, width, pairs)
        ctx.factor, ctx.owner = float(factor), owner
        return xyz, scale, quat, direction

    @staticmethod
    def backward(ctx, g_xyz, g_scale, g_quat, g_dir):
        endpoints, width, pairs = ctx.saved_tensors
        if ctx.owner is not None:
            ctx.owner._derived = None  # the graph behind the cached tensors is gone after this call
        P, E, dev = pairs.shape[0], endpoints.shape[0], endpoints.device
        gs = [None if g is None else g.contiguous() for g in (g_xyz, g_scale, g_quat, g_dir)]
        d_ep = torch.empty((E, 3), dtype=torch.float32, device=dev)
        d_w = torch.empty((P, 1), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rt.check(rt.lib().hgs_strand_geometry_backward(rt.current_stream(), P, E, rt.ptr(endpoints), rt.ptr(pairs),
                                                           rt.ptr(width), ctx.factor, rt.ptr(gs[0]), rt.ptr(gs[1]),
                                                           rt.ptr(gs[2]), rt.ptr(gs[3]), rt.ptr(d_ep), rt.ptr(d_w)))
        return d_ep, d_w, None, None, None


def strand_geometry(endpoints, width, pairs, factor, owner=None):
    """(xyz[P,3], scale[P,3], quat[P,4], direction[P,3]) of every segment, differentiable w.r.t. endpoints/width."""
    return _StrandGeometry.apply(endpoints, width, pairs, factor, owner)


class _SsimL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, gt):
        img = rt.require_gpu_tensor(img, "image", torch.float32)
        gt = rt.require_gpu_tensor(gt, "gt_image", torch.float32)
        Cc, H, W = img.shape[-3], img.shape[-2], img.shape[-1]
        dev, L = img.device, rt.lib()
        nb = L.hgs_ssim_l1_num_blocks(Cc, H, W)
        dmaps = torch.empty((3, Cc, H, W), dtype=torch.float32, device=dev)
        partials = torch.empty((nb, 2), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rt.check(L.hgs_ssim_l1_forward(rt.current_stream(), Cc, H, W, gaussian_window11(), rt.ptr(img), rt.ptr(gt),
                                           rt.ptr(dmaps), rt.ptr(partials)))
        sums = partials.sum(dim=0) / float(Cc * H * W)
        ctx.save_for_backward(img, gt, dmaps)
        return sums[0], sums[1]  # mean SSIM, mean |img - gt|

    @staticmethod
    def backward(ctx, g_ssim, g_l1):
        img, gt, dmaps = ctx.saved_tensors
        Cc, H, W = img.shape[-3], img.shape[-2], img.shape[-1]
        g_ssim = g_ssim.contiguous().to(torch.float32)
        g_l1 = g_l1.contiguous().to(torch.float32)
        d_img = torch.empty_like(img)
        with torch.cuda.device(img.device):
            rt.check(rt.lib().hgs_ssim_l1_backward(rt.current_stream(), Cc, H, W, gaussian_window11(), rt.ptr(img),
                                                   rt.ptr(gt), rt.ptr(dmaps), rt.ptr(g_ssim), rt.ptr(g_l1), rt.ptr(d_img)))
        return d_img, None


def ssim_l1(img, gt):
    """(mean SSIM, mean L1) of [C,H,W] images in one fused pass (window 11, sigma 1.5, zero padding)."""
    return _SsimL1.apply(img, gt)


class _OrientationLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, omap, viewmatrix, bg, min_val, gt_theta, confidence, mask):
        omap = rt.require_gpu_tensor(omap, "orientation map", torch.float32)
        view = rt.require_gpu_tensor(viewmatrix, "world_view_transform", torch.float32)
        gt_theta = rt.require_gpu_tensor(gt_theta, "orientation_field", torch.float32)
        confidence = rt.require_gpu_tensor(confidence, "orientation_confidence", torch.float32)
        H, W, dev, L = omap.shape[1], omap.shape[2], omap.device, rt.lib()
        mask_u8 = None
        if mask is not None:
            mask = rt.require_gpu_tensor(mask, "mask")
            mask_u8 = mask.view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8)
        nb = L.hgs_orientation_loss_num_blocks(H, W)
        partials = torch.empty((nb, 2), dtype=torch.float32, device=dev)
        bg3 = (C.c_float * 3)(*bg)
        with torch.cuda.device(dev):
            rt.check(L.hgs_orientation_loss_forward(rt.current_stream(), H, W, rt.ptr(omap), rt.ptr(view), bg3, float(min_val),
                                                    rt.ptr(gt_theta), rt.ptr(confidence), rt.ptr(mask_u8), rt.ptr(partials)))
        sums = partials.sum(dim=0)
        ctx.save_for_backward(omap, view, gt_theta, confidence,
                              mask_u8 if mask_u8 is not None else torch.empty(0, device=dev), sums)
        ctx.consts = (bg3, float(min_val), mask_u8 is not None)
        return sums[0] / sums[1]

    @staticmethod
    def backward(ctx, g):
        omap, view, gt_theta, confidence, mask_u8, sums = ctx.saved_tensors
        bg3, min_val, has_mask = ctx.consts
        H, W = omap.shape[1], omap.shape[2]
        g = g.contiguous().to(torch.float32)
        count = sums[1:2].contiguous()
        d = torch.empty_like(omap)
        with torch.cuda.device(omap.device):
            rt.check(rt.lib().hgs_orientation_loss_backward(rt.current_stream(), H, W, rt.ptr(omap), rt.ptr(view), bg3,
                                                            min_val, rt.ptr(gt_theta), rt.ptr(confidence),
                                                            rt.ptr(mask_u8) if has_mask else None, rt.ptr(g), rt.ptr(count),
                                                            rt.ptr(d)))
        return d, None, None, None, None, None, None


def orientation_loss(omap, viewmatrix, bg3, min_val, gt_theta, confidence, mask):
    return _OrientationLoss.apply(omap, viewmatrix, bg3, min_val, gt_theta, confidence, mask)
