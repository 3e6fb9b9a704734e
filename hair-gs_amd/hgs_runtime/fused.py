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


def _strand_geometry_launch(endpoints, width, pairs, factor):
    endpoints = rt.require_gpu_tensor(endpoints, "endpoints", torch.float32)
    width = rt.require_gpu_tensor(width, "width", torch.float32)
    pairs = rt.require_gpu_tensor(pairs, "endpoint_pairs", torch.int64)
    P, dev = pairs.shape[0], endpoints.device
    # one allocation for the four outputs (13 floats per segment; every view 16-byte aligned)
    buf = torch.empty((13 * P + 12,), dtype=torch.float32, device=dev)
    o = [0, (3 * P + 3) // 4 * 4]
    o.append(o[1] + (3 * P + 3) // 4 * 4)
    o.append(o[2] + 4 * P)
    xyz, scale = buf[o[0]:o[0] + 3 * P].view(P, 3), buf[o[1]:o[1] + 3 * P].view(P, 3)
    quat, direction = buf[o[2]:o[2] + 4 * P].view(P, 4), buf[o[3]:o[3] + 3 * P].view(P, 3)
    with torch.cuda.device(dev):
        rt.check(rt.lib().hgs_strand_geometry_forward(rt.current_stream(), P, rt.ptr(endpoints), rt.ptr(pairs),
                                                      rt.ptr(width), float(factor), rt.ptr(xyz), rt.ptr(scale),
                                                      rt.ptr(quat), rt.ptr(direction)))
    return endpoints, width, pairs, (xyz, scale, quat, direction)


class _StrandGeometry(torch.autograd.Function):
    @staticmethod
    def forward(ctx, endpoints, width, pairs, factor):
        endpoints, width, pairs, out = _strand_geometry_launch(endpoints, width, pairs, factor)
        ctx.save_for_backward(endpoints, width, pairs)
        ctx.factor = float(factor)
        return out

    @staticmethod
    def backward(ctx, g_xyz, g_scale, g_quat, g_dir):
        endpoints, width, pairs = ctx.saved_tensors
        P, E, dev = pairs.shape[0], endpoints.shape[0], endpoints.device
        gs = [None if g is None else g.contiguous() for g in (g_xyz, g_scale, g_quat, g_dir)]
        d_ep = torch.empty((E, 3), dtype=torch.float32, device=dev)
        d_w = torch.empty((P, 1), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rt.check(rt.lib().hgs_strand_geometry_backward(rt.current_stream(), P, E, rt.ptr(endpoints), rt.ptr(pairs),
                                                           rt.ptr(width), ctx.factor, rt.ptr(gs[0]), rt.ptr(gs[1]),
                                                           rt.ptr(gs[2]), rt.ptr(gs[3]), rt.ptr(d_ep), rt.ptr(d_w)))
        return d_ep, d_w, None, None


def strand_geometry(endpoints, width, pairs, factor):
    """(xyz[P,3], scale[P,3], quat[P,4], direction[P,3]) of every segment, differentiable w.r.t. endpoints/width."""
    if not torch.is_grad_enabled():          # forward-only (render loops): the launch without an autograd node around it
        return _strand_geometry_launch(endpoints, width, pairs, factor)[3]
    return _StrandGeometry.apply(endpoints, width, pairs, factor)


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


class _SmoothnessLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, endpoints, index_pairs, cos_th, eps):
        endpoints = rt.require_gpu_tensor(endpoints, "endpoints", torch.float32)
        idx = rt.require_gpu_tensor(index_pairs, "index_pairs", torch.int64)
        N, dev, L = idx.shape[0], endpoints.device, rt.lib()
        partials = torch.empty((max(L.hgs_smoothness_num_blocks(N), 1), 2), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rt.check(L.hgs_smoothness_forward(rt.current_stream(), N, rt.ptr(endpoints), rt.ptr(idx), float(cos_th),
                                              float(eps), rt.ptr(partials)))
        sums = partials.sum(dim=0)
        ctx.save_for_backward(endpoints, idx, sums)
        ctx.consts = (float(cos_th), float(eps))
        return sums[0] / torch.clamp(sums[1], min=1.0)

    @staticmethod
    def backward(ctx, g):
        endpoints, idx, sums = ctx.saved_tensors
        cos_th, eps = ctx.consts
        g = g.contiguous().to(torch.float32)
        count = sums[1:2].contiguous()
        d = torch.empty_like(endpoints)
        with torch.cuda.device(endpoints.device):
            rt.check(rt.lib().hgs_smoothness_backward(rt.current_stream(), idx.shape[0], endpoints.shape[0],
                                                      rt.ptr(endpoints), rt.ptr(idx), cos_th, eps, rt.ptr(g),
                                                      rt.ptr(count), rt.ptr(d)))
        return d, None, None, None


def smoothness_loss(endpoints, index_pairs, cos_th, eps):
    """Mean squared bending angle over the consecutive-segment pairs bent beyond the threshold (0 if none)."""
    return _SmoothnessLoss.apply(endpoints, index_pairs, cos_th, eps)


class FusedAdam(torch.optim.Optimizer):
    """Adam with the reference's settings (eps 1e-15, betas (0.9, 0.999), no weight decay, one group per tensor, per-group
    lr) whose whole update is ONE kernel launch (hgs_adam_step).  State layout (`exp_avg`, `exp_avg_sq`, `step` per
    parameter) is torch.optim.Adam's, so the models' optimizer-state surgery works unchanged.  `lr` of a group may be a
    Python float or a device scalar tensor (the latter is what graph capture needs)."""

    def __init__(self, params, lr=0.0, betas=(0.9, 0.999), eps=1e-15):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._lr_dev = {}
        self._call = None
        self._tickets = None   # the kernel's per-tensor ticket words: owned by this optimizer (include/hgs.h hgs_adam_step)
        self._plan = None      # _InlinePlan: the in-lane form of the update (include/hgs.h HgsAdamSlot)
        self._inline_done = False

    def inline_plan(self):
        """The plan of the in-lane update for the CURRENT parameter tensors (rebuilt when a tensor, a moment, a step counter or
        a learning-rate tensor has been replaced): device-resident HgsAdamPrep + coefficient table, slots per parameter."""
        rows = []
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                if p.numel() == 0:
                    continue
                if not p.is_cuda:
                    raise rt.HgsError("FusedAdam runs on the GPU only")
                st = self._state_for(p)
                rows.append((p, st["exp_avg"], st["exp_avg_sq"], self._lr_tensor(gi, group, p.device), st["step"], group))
        key = tuple(t.data_ptr() for r in rows for t in r[:5])
        if self._plan is None or self._plan.key != key:
            if torch.cuda.is_current_stream_capturing():
                # a new plan allocates and uploads from pageable memory: not capturable -- and a graph captured with the OLD plan
                # would go on writing its (freed) tables.  Whoever captures builds the plan first (train.GraphedStep does).
                raise rt.HgsError("FusedAdam.inline_plan(): a parameter / moment / learning-rate tensor was replaced during a "
                                  "graph capture; build the plan (enable_inline_adam) before capturing")
            self._plan = _InlinePlan(self, rows, key)
        return self._plan

    def zero_grad(self, set_to_none=True):
        # an in-lane update that no step() consumed (a backward without an optimizer step) must not swallow a later step()
        self._inline_done = False
        return super().zero_grad(set_to_none=set_to_none)

    def _state_for(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    def _lr_tensor(self, gi, group, dev):
        lr = group["lr"]
        if torch.is_tensor(lr):
            return lr
        cached = self._lr_dev.get(gi)
        if cached is None or cached[0] != float(lr) or cached[1].device != dev:
            t = cached[1] if cached is not None and cached[1].device == dev else torch.empty((), dtype=torch.float32, device=dev)
            t.fill_(float(lr))
            cached = (float(lr), t)
            self._lr_dev[gi] = cached
        return cached[1]

    @torch.no_grad()
    def step(self, closure=None):
        if self._inline_done:
            # the backward that just ran applied the update in its own lanes (strand_step, _InlinePlan.applied): nothing to
            # launch; the parameters changed in place behind autograd's back, like after the launch below
            self._inline_done = False
            for group in self.param_groups:
                for p in group["params"]:
                    if p.numel():
                        torch.autograd.graph.increment_version(p)
            return None
        rows = []
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                if p.grad is None or p.numel() == 0:
                    continue
                if not p.is_cuda:
                    raise rt.HgsError("FusedAdam runs on the GPU only")
                st = self._state_for(p)
                rows.append((p, p.grad.contiguous(), st["exp_avg"], st["exp_avg_sq"], self._lr_tensor(gi, group, p.device),
                             st["step"], group))
        if not rows:
            return None
        beta1, beta2 = rows[0][6]["betas"]
        eps = rows[0][6]["eps"]
        key = tuple(t.data_ptr() for r in rows for t in r[:6]) + tuple(r[0].numel() for r in rows)
        if self._call is None or self._call[0] != key:
            n = len(rows)
            arrs = [(C.c_void_p * n)(*[r[j].data_ptr() for r in rows]) for j in range(6)]
            numel = (C.c_longlong * n)(*[r[0].numel() for r in rows])
            self._call = (key, n, arrs, numel)
        _, n, arrs, numel = self._call
        dev = rows[0][0].device
        if self._tickets is None or self._tickets.device != dev:
            self._tickets = torch.zeros(8, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            rt.check(rt.lib().hgs_adam_step(rt.current_stream(), n, arrs[0], arrs[1], arrs[2], arrs[3], arrs[4], arrs[5],
                                            numel, float(beta1), float(beta2), float(eps), self._tickets.data_ptr()))
        # the kernel wrote the parameters through raw pointers: tell autograd that they changed in place
        for r in rows:
            torch.autograd.graph.increment_version(r[0])
        return None


class _InlinePlan:
    """FusedAdam.inline_plan(): what the kernels of an iteration with the in-lane update need of the optimizer (include/hgs.h
    HgsAdamPrep / HgsAdamSlot).  `prep_ptr`: the device-resident HgsAdamPrep the iteration's prologue works on."""

    def __init__(self, optimizer, rows, key):
        if len(rows) > rt.ADAM_MAX_TENSORS:
            raise rt.HgsError(f"in-lane Adam: at most {rt.ADAM_MAX_TENSORS} parameter tensors")
        self.optimizer, self.key = optimizer, key
        dev = rows[0][0].device
        beta1, beta2 = rows[0][5]["betas"]
        self.betas, self.eps = (float(beta1), float(beta2)), float(rows[0][5]["eps"])
        self.coef = torch.zeros((len(rows), 2), dtype=torch.float32, device=dev)
        prep = rt.AdamPrep()
        prep.n = len(rows)
        for k, r in enumerate(rows):
            prep.lr[k], prep.step[k] = r[3].data_ptr(), r[4].data_ptr()
        prep.beta1, prep.beta2, prep.coef = self.betas[0], self.betas[1], self.coef.data_ptr()
        self._prep = torch.frombuffer(bytearray(bytes(prep)), dtype=torch.uint8).to(dev)
        self.prep_ptr = self._prep.data_ptr()
        self._keep = rows
        self._slot = {id(r[0]): (r[0].data_ptr(), r[1].data_ptr(), r[2].data_ptr(), self.coef[k].data_ptr()) for k, r in enumerate(rows)}

    def fill(self, adam, params):
        """adam: rt.AdamInline; slot j <- the state of params[j]."""
        for j, p in enumerate(params):
            sl = adam.slot[j]
            sl.p, sl.m, sl.v, sl.coef = self._slot[id(p)]
        adam.beta1, adam.beta2, adam.eps = self.betas[0], self.betas[1], self.eps

    def applied(self):
        """The backward's launches that carry the update are enqueued: the optimizer's next step() launches nothing."""
        self.optimizer._inline_done = True
