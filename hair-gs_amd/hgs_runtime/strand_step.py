"""The strand-Gaussian training iteration as ONE autograd node over the C ABI (include/hgs.h: hgs_select_view,
hgs_hair_params_*, hgs_forward_render_multi / hgs_backward_multi, hgs_loss_head_*, hgs_densify_stats).

What the reference does between `gaussians` and `loss.backward()` in train.py:135-171 for a HairGaussianModel --
getters (scene/hair_gaussian_model.py:134-201), three render() calls, loss_function (loss/losses.py:319-355) and the
densification statistics (hair_gaussian_model.py:1401-1408) -- is ~60 small PyTorch kernels per iteration around the
rasterizer.  Here the same arithmetic is ~20 launches: parameters -> Gaussians (1), rasterizer forward (6), loss head
(4), loss head backward (4), rasterizer backward (4), Gaussians -> parameters (1), statistics (1).  Nothing is
approximated; tests/test_gpu_train.py checks loss, gradients and statistics against the op-by-op path.

Views live in a device-resident table (ViewTable); the iteration reads the current view through a 184-byte slot, so a
captured HIP graph switches views with one tiny launch (hgs_select_view) and no image copies."""
import ctypes as C
import os
import math

import numpy as np
import torch

import hgs_runtime as rt
from hgs_runtime.fused import gaussian_window11

_SLOT_BYTES = C.sizeof(rt.ViewTargets)
_F_VIEW, _F_PROJ, _F_CAM = (rt.ViewTargets.viewmatrix.offset // 4, rt.ViewTargets.projmatrix.offset // 4,
                            rt.ViewTargets.campos.offset // 4)


class ViewTable:
    """HgsViewTargets rows of a list of cameras in device memory + the slot the kernels read.  The cameras' tensors are
    referenced, not copied (they must stay alive and unchanged)."""

    def __init__(self, cameras, device=None, targets=True):
        """targets=False: the rows carry the camera matrices only (forward-only rendering, gaussian_renderer.frames): the
        cameras need no ground-truth tensors and the loss head must not be run on such a table."""
        if len(cameras) == 0:
            raise rt.HgsError("ViewTable needs at least one camera")
        c0 = cameras[0]
        self.has_targets = bool(targets)
        if device is not None:
            self.device = torch.device(device)
        elif targets:
            self.device = c0.original_image.device
        else:
            self.device = c0.world_view_transform.device
        self.H, self.W = int(c0.image_height), int(c0.image_width)
        self.tanfovx, self.tanfovy = math.tan(c0.FoVx * 0.5), math.tan(c0.FoVy * 0.5)
        self.has_float_mask = targets and getattr(c0, "float_mask", None) is not None
        self.has_mask = targets and getattr(c0, "mask", None) is not None
        rows = (rt.ViewTargets * len(cameras))()
        self._keep = []
        for i, c in enumerate(cameras):
            if (int(c.image_height), int(c.image_width), c.FoVx, c.FoVy) != (self.H, self.W, c0.FoVx, c0.FoVy):
                raise rt.HgsError("all views of a ViewTable must share resolution and field of view")
            if targets and ((getattr(c, "float_mask", None) is not None) != self.has_float_mask or
                            (getattr(c, "mask", None) is not None) != self.has_mask):
                raise rt.HgsError("either every view has a mask or none")
            r = rows[i]
            keep = []
            if targets:
                img = rt.require_gpu_tensor(c.original_image, "original_image", torch.float32)
                ori = rt.require_gpu_tensor(c.orientation_field, "orientation_field", torch.float32)
                conf = rt.require_gpu_tensor(c.orientation_confidence, "orientation_confidence", torch.float32)
                keep = [img, ori, conf]
                r.image, r.orientation, r.confidence = img.data_ptr(), ori.data_ptr(), conf.data_ptr()
            if self.has_float_mask:
                fm = rt.require_gpu_tensor(c.float_mask, "float_mask", torch.float32)
                keep.append(fm)
                r.float_mask = fm.data_ptr()
            if self.has_mask:
                m = rt.require_gpu_tensor(c.mask, "mask")
                m = m.view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8)
                keep.append(m)
                r.mask = m.data_ptr()
                r.mask_count = float(m.sum().item())
            for k, v in enumerate(c.world_view_transform.detach().float().cpu().reshape(-1).tolist()):
                r.viewmatrix[k] = v
            for k, v in enumerate(c.full_proj_transform.detach().float().cpu().reshape(-1).tolist()):
                r.projmatrix[k] = v
            for k, v in enumerate(c.camera_center.detach().float().cpu().reshape(-1).tolist()):
                r.campos[k] = v
            self._keep.append(keep)
        host = torch.from_numpy(np.frombuffer(bytes(rows), dtype=np.uint8).copy())
        self.table = host.to(self.device)
        self.slot = torch.zeros(_SLOT_BYTES, dtype=torch.uint8, device=self.device)
        f = self.slot.view(torch.float32)
        self.viewmatrix, self.projmatrix, self.campos = f[_F_VIEW:_F_VIEW + 16], f[_F_PROJ:_F_PROJ + 16], f[_F_CAM:_F_CAM + 3]
        self._cameras = list(cameras)     # (keeps the ids below unique for the table's lifetime)
        self.index = {id(c): i for i, c in enumerate(cameras)}
        self.n = len(cameras)
        self.current = -1
        # several views per captured step (include/hgs.h hgs_set_view_queue / hgs_select_view_queued)
        self.queue = torch.zeros(rt.VIEW_QUEUE_MAX, dtype=torch.int32, device=self.device)
        self.queue_lr = torch.zeros((), dtype=torch.float32, device=self.device)
        self._image, self._zero, self._image_ready = None, None, False
        self._rider = None
        # True while the image buffer's per-tile instance counters are known to be zero: what a forward pass in capacity mode
        # leaves behind (its scan clears them), and what the fused parameters + preprocess launch needs at its start
        self.counts_clean = False
        self._fill_behind = False      # zero range of the last fill_prologue(): True = starts behind those counters
        # device pointer of an HgsAdamPrep (hgs_runtime.fused.FusedAdam.inline_plan) every prologue issued from here on carries, or
        # None: an iteration whose backward applies Adam in its own lanes has its step counters advanced and its coefficients
        # formed by the prologue (include/hgs.h HgsAdamSlot).  Set by whoever runs such iterations (train.GraphedStep around its
        # captures, FusedStrandStep.enable_inline_adam), None otherwise.
        self.adam_prep = None

    def select(self, view, lr=0.0, lr_dst=None):
        """slot <- table[view] (and *lr_dst <- lr): one launch on the current stream."""
        if not 0 <= int(view) < self.n:
            raise rt.HgsError(f"view {view} outside the table (0..{self.n - 1})")
        with torch.cuda.device(self.device):
            rt.check(rt.lib().hgs_select_view(rt.current_stream(), self.table.data_ptr(), int(view), self.slot.data_ptr(),
                                              float(lr), None if lr_dst is None else lr_dst.data_ptr()))
        self.current = int(view)
        self._rider = None

    # ---- iteration prologue: view select + clearing of the image buffer's counters in ONE launch (include/hgs.h) ----
    def _image_zero_range(self, behind_counts=False):
        """(pointer, bytes) of the counters a prologue clears; behind_counts: without the per-tile instance counters in front."""
        if self._image is None:
            L = rt.lib()
            self._image = torch.zeros((L.hgs_image_bytes(self.W, self.H),), dtype=torch.uint8, device=self.device)
            self.counts_clean = True
            off, nbytes = C.c_size_t(0), C.c_size_t(0)
            rt.check(L.hgs_image_zero_range(self.W, self.H, C.addressof(off), C.addressof(nbytes)))
            lay = rt.layout("image", self.W, self.H)
            assert lay["tile_count"] == off.value and off.value < lay["tile_cursor"] < off.value + nbytes.value
            skip = lay["tile_cursor"] - off.value
            self._zero = ((self._image.data_ptr() + off.value, nbytes.value),
                          (self._image.data_ptr() + off.value + skip, nbytes.value - skip))
        return self._zero[1 if behind_counts else 0]

    def fused_preprocess_possible(self):
        return ((self.W + 15) // 16) * ((self.H + 15) // 16) <= rt.FUSED_PREPROCESS_MAX_TILES

    def prologue(self, view, lr=0.0, lr_dst=None, ride=False):
        """select(view, lr, lr_dst) and, in the same launch, the clearing of this table's image buffer for the coming
        forward pass, which take_image() then hands to the rasterizer (HGS_IMAGE_PREZEROED).
        ride=True: no launch here -- the prologue rides in spare workgroups of the fused iteration's first launch
        (fill_prologue(); the caller promises that Fused*Step.loss() is the next thing it runs on this table)."""
        if not 0 <= int(view) < self.n:
            raise rt.HgsError(f"view {view} outside the table (0..{self.n - 1})")
        zp, zb = self._image_zero_range()
        if ride:
            self._rider = (int(view), float(lr), None if lr_dst is None else lr_dst.data_ptr())
        else:
            self._rider = None
            with torch.cuda.device(self.device):
                rt.check(rt.lib().hgs_iteration_prologue(rt.current_stream(), self.table.data_ptr(), int(view),
                                                         self.slot.data_ptr(), float(lr),
                                                         None if lr_dst is None else lr_dst.data_ptr(), zp, zb, self.adam_prep))
            self.counts_clean = True
        self.current = int(view)
        self._image_ready = True

    def fill_prologue(self, fu, behind_counts=False):
        """Hand a prologue(ride=True) to the StrandFusion of the parameter forward launch (once).  behind_counts: the launch
        is hgs_hair_forward_preprocess, whose rider must leave the per-tile instance counters alone (include/hgs.h)."""
        if self._rider is None:
            return
        view, lr, lr_dst = self._rider
        adam_prep = self.adam_prep            # (as of now: the iteration's forward has just refreshed the optimizer's plan)
        self._rider = None
        self._fill_behind = bool(behind_counts)
        zp, zb = self._image_zero_range(behind_counts)
        pro = fu.prologue
        pro.table, pro.view, pro.slot, pro.lr, pro.lr_dst = self.table.data_ptr(), view, self.slot.data_ptr(), lr, lr_dst
        pro.zero_ptr, pro.zero_bytes, pro.adam_prep = zp, zb, adam_prep

    def flush_prologue(self):
        """A prologue(ride=True) nobody carried: launch it now (callers whose first launch cannot take a rider)."""
        if self._rider is None:
            return
        view, lr, lr_dst = self._rider
        adam_prep = self.adam_prep
        self._rider = None
        self._fill_behind = False
        zp, zb = self._image_zero_range()
        with torch.cuda.device(self.device):
            rt.check(rt.lib().hgs_iteration_prologue(rt.current_stream(), self.table.data_ptr(), view, self.slot.data_ptr(), lr,
                                                     lr_dst, zp, zb, adam_prep))
        self.counts_clean = True

    def ensure_counts_clean(self):
        """Zero the image buffer's per-tile instance counters unless they are known to be zero (one small fill; needed in
        front of a replayed graph whose first launch counts into them, after a blocking-mode pass used the buffer)."""
        if self.counts_clean or self._image is None:
            return
        (p0, _), (p1, _) = self._image_zero_range(False), self._image_zero_range(True)
        off = p0 - self._image.data_ptr()
        self._image[off:off + (p1 - p0)].zero_()
        self.counts_clean = True

    def take_image(self):
        """The image buffer whose counters the last prologue() cleared (once), else None."""
        if not self._image_ready:
            return None
        self._image_ready = False
        return self._image

    def graph_bind(self, cuda_graph, count=1):
        """After the capture of a graph that holds `count` prologue() launches, captured with lr = 0, 1, ... count - 1 as
        tags: find those nodes (torch.cuda.CUDAGraph built with keep_graph=True and instantiated).  Returns the binding
        graph_set() takes to re-point them between replays without any launch."""
        nodes, tags, n = (C.c_void_p * count)(), (C.c_float * count)(), C.c_int(0)
        rt.check(rt.lib().hgs_graph_find_prologues(C.c_void_p(int(cuda_graph.raw_cuda_graph())), count, nodes, tags, C.addressof(n)))
        order = sorted(range(n.value), key=lambda i: tags[i])
        if n.value != count or (count > 1 and [int(tags[i]) for i in order] != list(range(count))):
            raise rt.HgsError(f"graph_bind: expected {count} tagged prologue nodes, found {n.value} with tags {[tags[i] for i in order]}")
        return (cuda_graph, [nodes[i] for i in order], self._fill_behind)

    def graph_set(self, binding, view, lr=0.0, lr_dst=None, k=0):
        """Re-point the k-th prologue node of a graph_bind() result (host work only; takes effect at the next replay)."""
        if not 0 <= int(view) < self.n:
            raise rt.HgsError(f"view {view} outside the table (0..{self.n - 1})")
        graph, nodes, behind = binding
        zp, zb = self._image_zero_range(behind)
        rt.check(rt.lib().hgs_graph_set_prologue(C.c_void_p(int(graph.raw_cuda_graph_exec())), C.c_void_p(nodes[k]),
                                                 self.table.data_ptr(), int(view), self.slot.data_ptr(), float(lr),
                                                 None if lr_dst is None else lr_dst.data_ptr(), zp, zb))
        self.current = int(view)

    def set_queue(self, views, lr=0.0):
        """queue[:len(views)] <- views, queue_lr <- lr: one launch, values travel as kernel arguments."""
        views = [int(v) for v in views]
        if not 1 <= len(views) <= rt.VIEW_QUEUE_MAX or any(not 0 <= v < self.n for v in views):
            raise rt.HgsError(f"view queue {views}: 1..{rt.VIEW_QUEUE_MAX} views inside the table (0..{self.n - 1})")
        arr = (C.c_int * len(views))(*views)
        with torch.cuda.device(self.device):
            rt.check(rt.lib().hgs_set_view_queue(rt.current_stream(), self.queue.data_ptr(), len(views), arr, float(lr),
                                                 self.queue_lr.data_ptr()))

    def select_queued(self, k, lr_dst=None):
        """slot <- table[queue[k]] (and *lr_dst <- queue_lr), everything read on the device: capturable once, replayable
        with a different queue."""
        with torch.cuda.device(self.device):
            rt.check(rt.lib().hgs_select_view_queued(rt.current_stream(), self.table.data_ptr(), self.n,
                                                     self.queue[k:].data_ptr(), self.slot.data_ptr(), self.queue_lr.data_ptr(),
                                                     None if lr_dst is None else lr_dst.data_ptr()))
        self.current = -1


def _adjacency(member_ids, roles, n_targets, max_degree):
    """member_ids: flat int64 [n_items * roles] of target ids (item i, role r at i * roles + r).  Returns the int32 table
    [n_targets, max_degree] of codes item * roles + role (-1 = empty), or None if some target has more than max_degree."""
    ids = member_ids.to(torch.int64)
    if ids.numel() == 0:
        return torch.full((n_targets, max_degree), -1, dtype=torch.int32, device=ids.device)
    order = torch.argsort(ids, stable=True)
    ids_s = ids[order]
    first = torch.searchsorted(ids_s, ids_s, right=False)
    slot = torch.arange(ids.numel(), device=ids.device) - first
    if int(slot.max()) >= max_degree:
        return None
    table = torch.full((n_targets, max_degree), -1, dtype=torch.int32, device=ids.device)
    table[ids_s, slot] = order.to(torch.int32)          # position in the flat array IS item * roles + role
    return table.contiguous()


def head_params(H, W, opt, n_smooth, n_endpoints, min_val, has_float_mask, threshold_deg=30.0, eps=1e-6):
    p = rt.HeadParams()
    p.H, p.W = int(H), int(W)
    p.lambda_dssim = float(opt.lambda_dssim)
    p.lambda_mask = float(opt.lambda_mask) if has_float_mask else 0.0
    p.lambda_orientation = float(opt.lambda_orientation)
    p.lambda_smooth = float(getattr(opt, "lambda_smooth", 0.0)) if n_smooth > 0 else 0.0
    p.bg[0] = p.bg[1] = p.bg[2] = 0.0          # the orientation / mask renders use a black background (losses.py:247,312)
    p.min_val = float(min_val)
    win = gaussian_window11()
    for k in range(11):
        p.window[k] = win[k]
    p.n_smooth = int(n_smooth)
    p.cos_threshold = float(np.cos(threshold_deg * np.pi / 180))
    p.eps = float(eps)
    p.n_endpoints = int(n_endpoints)
    return p


def _raster_head_forward(step, xyz, scale, quat, opacity, extra4, shs, endpoints, smooth_idx, smooth_partials, n_endpoints,
                         hair=None):
    """Single-pass rasterizer forward + loss head on the CURRENT slot view (shared by the strand and the cloud iteration)."""
    from diff_gaussian_rasterization import _C as raster
    g, vt, hp, L = step.gaussians, step.views, step.head, rt.lib()
    dev = xyz.device
    f32 = dict(dtype=torch.float32, device=dev)
    empty = step.empty
    own_image = vt.take_image()
    R, planes, radii, geom, binning, img = raster.rasterize_gaussians_multi(
        step.bg7, xyz, empty, extra4, opacity, scale, quat, 1.0, empty, vt.viewmatrix, vt.projmatrix, vt.tanfovx,
        vt.tanfovy, vt.H, vt.W, shs, g.active_sh_degree, vt.campos, False, False, image_buffer=own_image, hair=hair)
    if own_image is not None and xyz.shape[0] > 0:
        vt.counts_clean = raster._state["last_counts_clean"]
    hp.n_endpoints = n_endpoints
    # the blend backward reads dL/dimage only on tiles where a pixel blended an entry (the image buffer's per-tile
    # contributor count, written by the forward pass above): the SSIM backward leaves the other blocks alone
    if step.skip_unread_blocks:
        if step._tile_maxc_offset is None:
            step._tile_maxc_offset = rt.layout("image", vt.W, vt.H)["tile_maxc"]
        hp.tile_used = img.data_ptr() + step._tile_maxc_offset
        hp.tiles_x, hp.tiles_y = (vt.W + 15) // 16, (vt.H + 15) // 16
    else:
        hp.tile_used = None
    scratch = torch.empty((L.hgs_loss_head_scratch_floats(C.byref(hp)),), **f32)
    out = torch.empty((rt.HEAD_NOUT,), **f32)
    # gradient planes of the per-pixel terms for an upstream gradient of 1, written by the forward's own pass over
    # the pixels (possible when the mask count is a per-view constant, i.e. the views carry masks)
    d_extra = torch.empty((4, vt.H, vt.W), **f32) if (step.one_pass_pixels and vt.has_mask) else None
    if step.poison_unwritten and d_extra is not None:     # test aid, like dL/dimage in _head_raster_backward
        d_extra.fill_(float("nan"))
    # the tail of the head's reduction rides in the backward's parameter launch (include/hgs.h HgsHeadTail)
    hp.defer_tail = 1 if (step.defer_tail and xyz.shape[0] > 0) else 0
    with torch.cuda.device(dev):
        rt.check(L.hgs_loss_head_forward(rt.current_stream(), C.byref(hp), planes[0:3].data_ptr(), planes[3].data_ptr(),
                                         planes[4:7].data_ptr(), vt.slot.data_ptr(), rt.ptr(endpoints), rt.ptr(smooth_idx),
                                         rt.ptr(scratch), rt.ptr(out), rt.ptr(d_extra), rt.ptr(smooth_partials)))
    return R, planes, radii, geom, binning, img, scratch, out, d_extra


def _head_raster_backward(ctx, step, go, xyz, scale, quat, shs, planes, radii, geom, binning, img, scratch, out, endpoints,
                          d_ep, n_endpoints, params=None):
    """Loss-head backward + single-pass rasterizer backward; returns (grad_out tensor, rasterizer gradients)."""
    from diff_gaussian_rasterization import _C as raster
    g, vt, hp, L = step.gaussians, step.views, step.head, rt.lib()
    dev = xyz.device
    f32 = dict(dtype=torch.float32, device=dev)
    # Fused*Step.backward() hands in its own ones tensor: the upstream gradient is then known to be exactly 1 and
    # the planes the forward wrote are final (the tensor's address, not its value, is what can be checked without a sync)
    unit_go = go is not None and go.data_ptr() == step.one.data_ptr()
    unit = ctx.d_extra is not None and unit_go
    go = step.one if go is None else go.contiguous().to(torch.float32)
    hp.n_endpoints = n_endpoints
    hp.defer_tail = 1 if ctx.defer_tail else 0
    if unit:
        d_image, d_extra = torch.empty((3, vt.H, vt.W), **f32), ctx.d_extra
    else:
        dplanes = torch.empty_like(planes)
        d_image, d_extra = dplanes[0:3], dplanes[3:7]
    if step.poison_unwritten:     # test aid: what the loss head leaves unwritten (HgsHeadParams.tile_used) must never be read
        d_image.fill_(float("nan"))
    with torch.cuda.device(dev):
        rt.check(L.hgs_loss_head_backward(rt.current_stream(), C.byref(hp), planes[0:3].data_ptr(), planes[3].data_ptr(),
                                          planes[4:7].data_ptr(), vt.slot.data_ptr(), rt.ptr(endpoints),
                                          rt.ptr(step.smooth_pairs), rt.ptr(scratch), rt.ptr(out), rt.ptr(go),
                                          (rt.HEAD_SKIP_PIXELS if unit else 0) | (rt.HEAD_SKIP_SMOOTH if ctx.fused_smooth else 0),
                                          d_image.data_ptr(), d_extra[0].data_ptr(), d_extra[1:4].data_ptr(), rt.ptr(d_ep)))
    grad_planes = [d_image[k] for k in range(3)] + [d_extra[k] for k in range(4)]
    if params is not None:
        # the backward mirror of the one-launch forward (include/hgs.h hgs_backward_multi_params): the rasterizer's per-Gaussian
        # gradients never leave the lane that computed them
        g_means2D = torch.empty((xyz.shape[0], 3), **f32)
        params.dL_dmeans2D_rgb = rt.ptr(g_means2D)
        g_sh = raster.rasterize_gaussians_multi_backward_params(
            step.bg7_backward, xyz, radii, scale, quat, vt.viewmatrix, vt.projmatrix, vt.tanfovx, vt.tanfovy, grad_planes, shs,
            g.active_sh_degree, vt.campos, geom, ctx.R, binning, img, params)
        return go, (g_means2D, None, None, None, g_sh, None, None)
    empty = step.empty
    (g_means2D, _gc, g_ex, g_opac, g_means3D, _gcov, g_sh, g_scales, g_rot) = raster.rasterize_gaussians_multi_backward(
        step.bg7_backward, xyz, radii, empty, scale, quat, 1.0, empty, vt.viewmatrix, vt.projmatrix, vt.tanfovx, vt.tanfovy,
        grad_planes, shs, g.active_sh_degree, vt.campos, geom, ctx.R, binning, img, False)
    return go, (g_means2D, g_ex, g_opac, g_means3D, g_sh, g_scales, g_rot)


def _params_stats(step, pb):
    """Densification statistics (train.py:170-171) from the lanes of hgs_backward_multi_params."""
    g = step.gaussians
    step.last["stats_done"] = False
    if step.stats_in_backward:
        pb.max_radii2D, pb.grad_accum, pb.denom = g.max_radii2D.data_ptr(), g.xyz_gradient_accum.data_ptr(), g.denom.data_ptr()
        step.last["stats_done"] = True


def _tail_group(ctx, step, fu, scratch, out):
    """The loss head's deferred tail, run by a spare workgroup of the backward's parameter launch."""
    if ctx.defer_tail:
        rt.check(rt.lib().hgs_loss_head_tail(C.byref(step.head), rt.ptr(scratch), rt.ptr(out), C.byref(fu.head_tail)))


def _stats_group(step, fu, radii, g_means2D):
    """Densification statistics (train.py:170-171) folded into the backward's parameter launch."""
    g = step.gaussians
    step.last["stats_done"] = False
    if step.stats_in_backward:
        fu.radii, fu.dmean2D, fu.dmean2D_stride = radii.data_ptr(), g_means2D.data_ptr(), int(g_means2D.shape[1])
        fu.max_radii2D, fu.grad_accum, fu.denom = g.max_radii2D.data_ptr(), g.xyz_gradient_accum.data_ptr(), g.denom.data_ptr()
        step.last["stats_done"] = True


class _StrandIteration(torch.autograd.Function):
    @staticmethod
    def forward(ctx, endpoints, width, opacity_raw, mask_raw, f_dc, f_rest, step):
        from diff_gaussian_rasterization import _C as raster
        g, vt, L = step.gaussians, step.views, rt.lib()
        endpoints = rt.require_gpu_tensor(endpoints, "endpoints", torch.float32)
        width = rt.require_gpu_tensor(width, "width", torch.float32)
        opacity_raw = rt.require_gpu_tensor(opacity_raw, "opacity", torch.float32)
        mask_raw = rt.require_gpu_tensor(mask_raw, "mask", torch.float32)
        pairs = rt.require_gpu_tensor(g.endpoint_pairs, "endpoint_pairs", torch.int64)
        dev, P, E = endpoints.device, pairs.shape[0], endpoints.shape[0]
        f32 = dict(dtype=torch.float32, device=dev)
        xyz, scale, quat = torch.empty((P, 3), **f32), torch.empty((P, 3), **f32), torch.empty((P, 4), **f32)
        opacity, extra4 = torch.empty((P, 1), **f32), torch.empty((P, 4), **f32)
        factor = float(g.dist_to_scale_factor)
        stream = rt.current_stream()
        step.refresh_inline_plan()
        # ONE decision for the whole iteration: when the prologue carries the optimizer's plan (it advances the step counters on
        # the device), the backward MUST apply the update in its lanes -- or Adam's launch would advance them a second time
        ctx.use_inline = step.inline_plan() is not None
        if ctx.use_inline:
            step.inline_adam._inline_done = False
            if not (step.fuse_param_backward and P > 0 and step.ep_segments is not None and step.ep_segments.shape[0] == E):
                raise rt.HgsError("in-lane Adam is enabled but this iteration's backward cannot apply it (no segments, the fused "
                                  "parameter backward off, or an endpoint adjacency that predates a topology change: call "
                                  "refresh()); disable it with enable_inline_adam(False)")
        idx = step.smooth_pairs
        hp = step.head
        # the smoothness term rides in extra workgroups of the parameter kernels (HgsStrandFusion)
        fu = rt.StrandFusion()
        smooth_partials = pair_grads = None
        if idx is not None and hp.lambda_smooth > 0:
            smooth_partials = torch.empty((2 * ((idx.shape[0] + 255) // 256),), **f32)
            fu.smooth_pairs, fu.n_smooth = idx.data_ptr(), int(idx.shape[0])
            fu.cos_threshold, fu.eps = hp.cos_threshold, hp.eps
            fu.smooth_partials = smooth_partials.data_ptr()
            # the pairs' unit gradients for the backward's endpoint gather (they do not depend on the rasterizer: the forward's
            # spare workgroups compute them beside the preprocess workgroups, HgsStrandFusion.smooth_pair_grads)
            if step.fuse_param_backward and step.ep_pairs is not None:
                pair_grads = torch.empty((idx.shape[0], 2, 4), **f32)
                fu.smooth_pair_grads = pair_grads.data_ptr()
        shs = f_dc if f_rest.numel() == 0 else torch.cat((f_dc, f_rest), dim=1)
        if step.fuse_preprocess:
            # parameters -> Gaussians -> preprocess as ONE launch where the pass runs in capacity mode (HairSource); the riders
            # then run beside the preprocess workgroups, so the prologue must leave the tile counters alone -- which it may
            # only if they are known to be zero (ViewTable.counts_clean); otherwise it is launched on its own, clearing them
            def fill(fused):
                if fused and not vt.counts_clean:
                    vt.flush_prologue()
                vt.fill_prologue(fu, behind_counts=fused)
            hair = raster.HairSource(endpoints, pairs, width, factor, opacity_raw, mask_raw, fu, fill)
        else:
            hair = None
            vt.fill_prologue(fu)
            with torch.cuda.device(dev):
                rt.check(L.hgs_hair_params_forward(stream, P, rt.ptr(endpoints), rt.ptr(pairs), rt.ptr(width), factor,
                                                   rt.ptr(opacity_raw), rt.ptr(mask_raw), rt.ptr(xyz), rt.ptr(scale),
                                                   rt.ptr(quat), None, rt.ptr(opacity), rt.ptr(extra4), C.byref(fu)))
        R, planes, radii, geom, binning, img, scratch, out, d_extra = _raster_head_forward(
            step, xyz, scale, quat, opacity, extra4, shs, endpoints, idx, smooth_partials, E, hair=hair)
        ctx.d_extra = d_extra
        ctx.defer_tail = bool(step.head.defer_tail)
        ctx.fused_smooth = smooth_partials is not None
        ctx.step, ctx.R, ctx.f_rest_k = step, R, f_rest.shape[1]
        ctx.set_materialize_grads(False)   # no zero tensor for the (non-differentiable) terms output
        ctx.save_for_backward(endpoints, width, pairs, xyz, scale, quat, opacity, extra4, shs, planes, radii, geom, binning,
                              img, scratch, out)
        ctx.pair_grads = pair_grads
        step.last = {"planes": planes, "radii": radii, "terms": out}
        terms = out.detach()
        ctx.mark_non_differentiable(terms)
        return out[0], terms

    @staticmethod
    def backward(ctx, go, _):
        from diff_gaussian_rasterization import _C as raster
        step, L = ctx.step, rt.lib()
        g, vt, hp = step.gaussians, step.views, step.head
        (endpoints, width, pairs, xyz, scale, quat, opacity, extra4, shs, planes, radii, geom, binning, img, scratch,
         out) = ctx.saved_tensors
        dev, P, E = endpoints.device, pairs.shape[0], endpoints.shape[0]
        f32 = dict(dtype=torch.float32, device=dev)
        d_ep = torch.empty((E, 3), **f32)
        gather = step.ep_segments is not None and step.ep_segments.shape[0] == E   # endpoint adjacency known: no atomics
        d_w, d_o, d_m = torch.empty((P, 1), **f32), torch.empty((P, 1), **f32), torch.empty((P, 1), **f32)
        if gather and step.fuse_param_backward and P > 0:
            # ---- round 5: the segment geometry's backward in the rasterizer backward's own lanes, then the endpoint gather
            pb = rt.ParamBackward()
            seg_contrib = torch.empty((P, 2, 4), **f32)
            pb.kind, pb.endpoints, pb.endpoint_pairs = rt.PARAMS_HAIR, rt.ptr(endpoints), rt.ptr(pairs)
            pb.dist_to_scale_factor = float(g.dist_to_scale_factor)
            pb.seg_contrib, pb.d_width, pb.extra4 = rt.ptr(seg_contrib), rt.ptr(d_w), rt.ptr(extra4)
            pb.d_opacity_raw, pb.d_mask_raw = rt.ptr(d_o), rt.ptr(d_m)
            _params_stats(step, pb)
            plan = step.inline_plan() if ctx.use_inline else None
            ep_adam = None
            if plan is not None:      # Adam in the backward's own lanes (include/hgs.h HgsAdamSlot)
                plan.fill(pb.adam, (g._width, g._opacity, g._mask, g._features_dc))
                ep_adam = rt.AdamInline()
                plan.fill(ep_adam, (g._endpoints,))
            go, (g_means2D, _, _, _, g_sh, _, _) = _head_raster_backward(
                ctx, step, go, xyz, scale, quat, shs, planes, radii, geom, binning, img, scratch, out, endpoints, None, E,
                params=pb)
            fu = rt.StrandFusion()
            if ctx.fused_smooth:
                idx = step.smooth_pairs
                fu.smooth_pairs, fu.n_smooth = idx.data_ptr(), int(idx.shape[0])
                fu.cos_threshold, fu.eps = hp.cos_threshold, hp.eps
                fu.head_out, fu.grad_out = out.data_ptr(), go.data_ptr()
                if ctx.pair_grads is not None:
                    fu.smooth_pair_grads = ctx.pair_grads.data_ptr()
            _tail_group(ctx, step, fu, scratch, out)
            fu.ep_segments, fu.n_endpoints = step.ep_segments.data_ptr(), E
            fu.ep_pairs = None if step.ep_pairs is None else step.ep_pairs.data_ptr()
            with torch.cuda.device(dev):
                rt.check(L.hgs_hair_endpoint_gather(rt.current_stream(), E, rt.ptr(seg_contrib), rt.ptr(endpoints), rt.ptr(d_ep),
                                                    C.byref(fu), None if ep_adam is None else C.byref(ep_adam)))
            if plan is not None:
                plan.applied()
            step.last["dmean2D"] = g_means2D
            if ctx.f_rest_k == 0:
                d_dc, d_rest = g_sh, None
            else:
                d_dc, d_rest = g_sh[:, :1], g_sh[:, 1:]
            return d_ep, d_w, d_o, d_m, d_dc, d_rest, None
        if ctx.use_inline:
            raise rt.HgsError("the forward's prologue advanced Adam's step counters for an in-lane update this backward cannot apply")
        go, (g_means2D, g_ex, g_opac, g_means3D, g_sh, g_scales, g_rot) = _head_raster_backward(
            ctx, step, go, xyz, scale, quat, shs, planes, radii, geom, binning, img, scratch, out, endpoints,
            None if gather else d_ep, E)
        stream = rt.current_stream()
        fu = rt.StrandFusion()
        if ctx.fused_smooth:      # smoothness gradient: extra workgroups of the same launch, same d_ep
            idx = step.smooth_pairs
            fu.smooth_pairs, fu.n_smooth = idx.data_ptr(), int(idx.shape[0])
            fu.cos_threshold, fu.eps = hp.cos_threshold, hp.eps
            fu.head_out, fu.grad_out = out.data_ptr(), go.data_ptr()
        _stats_group(step, fu, radii, g_means2D)
        _tail_group(ctx, step, fu, scratch, out)
        if gather:
            fu.ep_segments, fu.n_endpoints = step.ep_segments.data_ptr(), E
            fu.ep_pairs = None if step.ep_pairs is None else step.ep_pairs.data_ptr()
        with torch.cuda.device(dev):
            rt.check(L.hgs_hair_params_backward(stream, P, E, rt.ptr(endpoints), rt.ptr(pairs), rt.ptr(width),
                                                float(g.dist_to_scale_factor), rt.ptr(opacity), rt.ptr(extra4),
                                                rt.ptr(g_means3D), rt.ptr(g_scales), rt.ptr(g_rot), None, rt.ptr(g_opac),
                                                rt.ptr(g_ex), 1, rt.ptr(d_ep), rt.ptr(d_w), rt.ptr(d_o), rt.ptr(d_m),
                                                C.byref(fu)))
        step.last["dmean2D"] = g_means2D      # RGB-only screen-space gradient: what the densification statistics see
        if ctx.f_rest_k == 0:
            d_dc, d_rest = g_sh, None
        else:
            d_dc, d_rest = g_sh[:, :1], g_sh[:, 1:]
        return d_ep, d_w, d_o, d_m, d_dc, d_rest, None


class FusedStrandStep:
    """Host side of the fused iteration for one HairGaussianModel and one set of views."""

    def __init__(self, gaussians, cameras, opt, bg):
        self.gaussians, self.opt = gaussians, opt
        self.views = cameras if isinstance(cameras, ViewTable) else ViewTable(cameras)
        if not self.views.has_targets:
            raise rt.HgsError("the training iteration needs a ViewTable built with the views' targets")
        dev = self.views.device
        self.bg7 = torch.cat([bg.to(dev, torch.float32), torch.zeros(4, device=dev)]).contiguous()
        # a black background (the training default, train.py:94) is handed to the backward as NULL: its terms are compiled
        # out of the blend backward (include/hgs.h hgs_backward_multi); one read-back here, at construction
        self.bg7_backward = None if (bool((self.bg7 == 0).all()) and os.environ.get("HGS_BLACK_BG", "1") != "0") else self.bg7
        self.empty = torch.empty(0, device=dev)
        self.one = torch.ones((), dtype=torch.float32, device=dev)   # d loss / d loss, passed to backward(): no fill launch
        self.one_pass_pixels = True    # per-pixel loss terms: value and gradient in one pass over the pixels
        self.stats_in_backward = True  # densification statistics updated by the backward's last launch
        # dL/dimage is produced only where the rasterizer backward reads it (include/hgs.h HgsHeadParams.tile_used)
        self.skip_unread_blocks = bool(getattr(opt, "skip_unread_blocks", True))
        self._tile_maxc_offset = None
        self.poison_unwritten = False   # tests: dL/dimage starts as NaN
        # strand parameters -> Gaussians -> preprocess as one launch (hgs_hair_forward_preprocess) where the pass allows it
        self.fuse_preprocess = bool(getattr(opt, "fuse_preprocess", True)) and os.environ.get("HGS_FUSE_PREPROCESS", "1") != "0"
        # the backward mirror: parameters' backward in the rasterizer backward's per-Gaussian lanes (hgs_backward_multi_params)
        self.fuse_param_backward = bool(getattr(opt, "fuse_param_backward", True)) and os.environ.get("HGS_FUSE_PARAM_BACKWARD", "1") != "0"
        # Adam in the backward's own lanes (include/hgs.h HgsAdamSlot; enable_inline_adam): the model's FusedAdam or None
        self.inline_adam = None
        # True: the loss terms (loss(), terms()) are complete only once backward() has run -- the head's last sums ride in
        # the backward's parameter launch instead of a launch of their own (GraphedStep, which always runs both, sets it)
        self.defer_tail = False
        self.last = {}
        self.refresh()

    def inline_adam_possible(self):
        """Can this iteration's backward apply the Adam update itself?  One rank, the parameters' backward fused into the
        rasterizer's (hgs_backward_multi_params; a strand model also needs the endpoint adjacency), a FusedAdam that holds
        exactly the in-lane tensors (higher SH coefficients would need a launch of their own) -- and not switched off."""
        from hgs_runtime.fused import FusedAdam
        g = self.gaussians
        return (self.fuse_param_backward and isinstance(getattr(g, "optimizer", None), FusedAdam)
                and g._features_rest.numel() == 0 and getattr(self, "ep_segments", True) is not None
                and bool(getattr(self.opt, "inline_adam", True)) and os.environ.get("HGS_INLINE_ADAM", "1") != "0"
                and not (torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1))

    def enable_inline_adam(self, on=True):
        """From the next prologue on (off: until further notice) the iteration applies Adam in its backward's lanes: the views'
        prologues carry the optimizer's step-counter / coefficient plan, backward() fills the kernels' Adam slots and tells the
        optimizer, whose step() then has nothing left to launch.  Returns whether it is on."""
        on = bool(on) and self.inline_adam_possible()
        self.inline_adam = self.gaussians.optimizer if on else None
        self.views.adam_prep = self.inline_plan().prep_ptr if on else None
        return on

    def inline_plan(self):
        return None if self.inline_adam is None else self.inline_adam.inline_plan()

    def refresh_inline_plan(self):
        """Start of an iteration's forward, in front of the launch that carries the prologue: the optimizer's plan as of now
        (learning rates given as Python numbers are refreshed on the device, replaced tensors picked up) into the views."""
        if self.inline_adam is not None:
            self.views.adam_prep = self.inline_plan().prep_ptr

    def refresh(self):
        """Call after anything that changes the strands' topology (the smoothness index table and sizes)."""
        g = self.gaussians
        idx = g.smoothness_index_pairs() if float(getattr(self.opt, "lambda_smooth", 0.0)) > 0 else None
        n = 0 if idx is None else int(idx.shape[0])
        self.smooth_pairs = rt.require_gpu_tensor(idx, "index_pairs", torch.int64) if n > 0 else None
        self.head = head_params(self.views.H, self.views.W, self.opt, n, g._endpoints.shape[0], g.min_val,
                                self.views.has_float_mask)
        # endpoint adjacency for the gather-mode backward (HgsStrandFusion.ep_segments / ep_pairs): an endpoint of a
        # chain touches <= 2 segments and <= 4 smoothness-pair roles; anything denser keeps the scatter (atomic) mode.
        # Remembered on the model for as long as both index tables are the same tensors (a topology event assigns new ones): the
        # eager iteration of training() and the GraphedStep captured after it each build a step object per event.
        E = g._endpoints.shape[0]
        cached = getattr(g, "_adjacency_cache", None)
        if cached is not None and cached[0] is g.endpoint_pairs and cached[1] is self.smooth_pairs and cached[2] == E:
            self.ep_segments, self.ep_pairs = cached[3], cached[4]
            return
        self.ep_segments = _adjacency(g.endpoint_pairs.reshape(-1), 2, E, 2)
        self.ep_pairs = None
        if self.ep_segments is not None and self.smooth_pairs is not None:
            self.ep_pairs = _adjacency(self.smooth_pairs.reshape(-1), 4, E, 4)
            if self.ep_pairs is None:
                self.ep_segments = None
        g._adjacency_cache = (g.endpoint_pairs, self.smooth_pairs, E, self.ep_segments, self.ep_pairs)

    def loss(self):
        """(total loss, terms tensor) of the CURRENT slot view; differentiable w.r.t. the model parameters."""
        g = self.gaussians
        return _StrandIteration.apply(g._endpoints, g._width, g._opacity, g._mask, g._features_dc, g._features_rest, self)

    def backward(self, loss):
        """loss.backward() without the ones_like() launch."""
        loss.backward(self.one)

    def terms(self):
        t = self.last["terms"]
        return {k: t[i] for i, k in enumerate(rt.HEAD_OUT[1:6], start=1)}

    def update_densification_stats(self):
        """add_densification_stats + max_radii2D of the iteration just back-propagated (one launch)."""
        g, last = self.gaussians, self.last
        if last.get("stats_done"):
            return                     # already folded into the backward (stats_in_backward)
        gm, radii = last["dmean2D"], last["radii"]
        with torch.cuda.device(gm.device):
            rt.check(rt.lib().hgs_densify_stats(rt.current_stream(), gm.shape[0], rt.ptr(radii), rt.ptr(gm), gm.shape[1],
                                                rt.ptr(g.max_radii2D), rt.ptr(g.xyz_gradient_accum), rt.ptr(g.denom)))


class _CloudIteration(torch.autograd.Function):
    """The same iteration for the Stage-I Gaussian cloud (scene/gaussian_model.py): raw (scaling, rotation, opacity, mask)
    -> rasterizer inputs by hgs_cloud_params_*; xyz and the SH features go to the rasterizer as they are."""

    @staticmethod
    def forward(ctx, xyz, scaling_raw, rotation_raw, opacity_raw, mask_raw, f_dc, f_rest, step):
        L = rt.lib()
        xyz = rt.require_gpu_tensor(xyz, "xyz", torch.float32)
        scaling_raw = rt.require_gpu_tensor(scaling_raw, "scaling", torch.float32)
        rotation_raw = rt.require_gpu_tensor(rotation_raw, "rotation", torch.float32)
        opacity_raw = rt.require_gpu_tensor(opacity_raw, "opacity", torch.float32)
        mask_raw = rt.require_gpu_tensor(mask_raw, "mask", torch.float32)
        dev, P = xyz.device, xyz.shape[0]
        f32 = dict(dtype=torch.float32, device=dev)
        scale, quat = torch.empty((P, 3), **f32), torch.empty((P, 4), **f32)
        opacity, extra4 = torch.empty((P, 1), **f32), torch.empty((P, 4), **f32)
        from diff_gaussian_rasterization import _C as raster
        fu = rt.StrandFusion()
        vt = step.views
        step.refresh_inline_plan()
        ctx.use_inline = step.inline_plan() is not None      # (one decision per iteration: see _StrandIteration.forward)
        if ctx.use_inline:
            step.inline_adam._inline_done = False
            if not (step.fuse_param_backward and P > 0):
                raise rt.HgsError("in-lane Adam is enabled but this iteration's backward cannot apply it (no Gaussians, or the "
                                  "fused parameter backward off); disable it with enable_inline_adam(False)")
        if step.fuse_preprocess:      # (as _StrandIteration: parameters -> Gaussians -> preprocess as one launch)
            def fill(fused):
                if fused and not vt.counts_clean:
                    vt.flush_prologue()
                vt.fill_prologue(fu, behind_counts=fused)
            src = raster.CloudSource(scaling_raw, rotation_raw, opacity_raw, mask_raw, fu, fill)
        else:
            src = None
            vt.fill_prologue(fu)
            with torch.cuda.device(dev):
                rt.check(L.hgs_cloud_params_forward(rt.current_stream(), P, rt.ptr(scaling_raw), rt.ptr(rotation_raw),
                                                    rt.ptr(opacity_raw), rt.ptr(mask_raw), rt.ptr(scale), rt.ptr(quat),
                                                    rt.ptr(opacity), rt.ptr(extra4), C.byref(fu)))
        shs = f_dc if f_rest.numel() == 0 else torch.cat((f_dc, f_rest), dim=1)
        R, planes, radii, geom, binning, img, scratch, out, d_extra = _raster_head_forward(
            step, xyz, scale, quat, opacity, extra4, shs, None, None, None, 0, hair=src)
        ctx.d_extra, ctx.fused_smooth = d_extra, True   # (no smoothness term for a cloud: nothing to launch)
        ctx.defer_tail = bool(step.head.defer_tail)
        ctx.step, ctx.R, ctx.f_rest_k = step, R, f_rest.shape[1]
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(xyz, scaling_raw, rotation_raw, scale, quat, opacity, extra4, shs, planes, radii, geom, binning,
                              img, scratch, out)
        step.last = {"planes": planes, "radii": radii, "terms": out}
        terms = out.detach()
        ctx.mark_non_differentiable(terms)
        return out[0], terms

    @staticmethod
    def backward(ctx, go, _):
        step, L = ctx.step, rt.lib()
        (xyz, scaling_raw, rotation_raw, scale, quat, opacity, extra4, shs, planes, radii, geom, binning, img, scratch,
         out) = ctx.saved_tensors
        dev, P = xyz.device, xyz.shape[0]
        f32 = dict(dtype=torch.float32, device=dev)
        d_s, d_r = torch.empty((P, 3), **f32), torch.empty((P, 4), **f32)
        d_o, d_m = torch.empty((P, 1), **f32), torch.empty((P, 1), **f32)
        if step.fuse_param_backward and P > 0:
            # ---- round 5: the raw parameters' gradients from the rasterizer backward's own lanes; no second launch
            pb = rt.ParamBackward()
            g_means3D = torch.empty((P, 3), **f32)
            pb.kind, pb.rotation_raw, pb.extra4 = rt.PARAMS_CLOUD, rt.ptr(rotation_raw), rt.ptr(extra4)
            pb.d_means3D, pb.d_scaling_raw, pb.d_rotation_raw = rt.ptr(g_means3D), rt.ptr(d_s), rt.ptr(d_r)
            pb.d_opacity_raw, pb.d_mask_raw = rt.ptr(d_o), rt.ptr(d_m)
            _params_stats(step, pb)
            plan = step.inline_plan() if ctx.use_inline else None
            if plan is not None:      # Adam in the backward's own lanes (include/hgs.h HgsAdamSlot)
                gm = step.gaussians
                plan.fill(pb.adam, (gm._xyz, gm._scaling, gm._rotation, gm._opacity, gm._mask, gm._features_dc))
            if ctx.defer_tail:
                rt.check(L.hgs_loss_head_tail(C.byref(step.head), rt.ptr(scratch), rt.ptr(out), C.byref(pb.head_tail)))
            go, (g_means2D, _, _, _, g_sh, _, _) = _head_raster_backward(
                ctx, step, go, xyz, scale, quat, shs, planes, radii, geom, binning, img, scratch, out, None, None, 0, params=pb)
            if plan is not None:
                plan.applied()
            step.last["dmean2D"] = g_means2D
            if ctx.f_rest_k == 0:
                d_dc, d_rest = g_sh, None
            else:
                d_dc, d_rest = g_sh[:, :1], g_sh[:, 1:]
            return g_means3D, d_s, d_r, d_o, d_m, d_dc, d_rest, None
        if ctx.use_inline:
            raise rt.HgsError("the forward's prologue advanced Adam's step counters for an in-lane update this backward cannot apply")
        go, (g_means2D, g_ex, g_opac, g_means3D, g_sh, g_scales, g_rot) = _head_raster_backward(
            ctx, step, go, xyz, scale, quat, shs, planes, radii, geom, binning, img, scratch, out, None, None, 0)
        fu = rt.StrandFusion()
        _stats_group(step, fu, radii, g_means2D)
        _tail_group(ctx, step, fu, scratch, out)
        with torch.cuda.device(dev):
            rt.check(L.hgs_cloud_params_backward(rt.current_stream(), P, rt.ptr(scaling_raw), rt.ptr(rotation_raw),
                                                 rt.ptr(opacity), rt.ptr(extra4), rt.ptr(g_scales), rt.ptr(g_rot),
                                                 rt.ptr(g_opac), rt.ptr(g_ex), rt.ptr(d_s), rt.ptr(d_r), rt.ptr(d_o),
                                                 rt.ptr(d_m), C.byref(fu)))
        step.last["dmean2D"] = g_means2D
        if ctx.f_rest_k == 0:
            d_dc, d_rest = g_sh, None
        else:
            d_dc, d_rest = g_sh[:, :1], g_sh[:, 1:]
        return g_means3D, d_s, d_r, d_o, d_m, d_dc, d_rest, None


class FusedCloudStep(FusedStrandStep):
    """Host side of the fused iteration for a Stage-I GaussianModel (same views, same loss head, no smoothness term)."""

    def refresh(self):
        g = self.gaussians
        self.smooth_pairs = None
        self.head = head_params(self.views.H, self.views.W, self.opt, 0, 0, getattr(g, "min_val", 1e-7),
                                self.views.has_float_mask)

    def loss(self):
        g = self.gaussians
        return _CloudIteration.apply(g._xyz, g._scaling, g._rotation, g._opacity, g._mask, g._features_dc, g._features_rest, self)


def fused_step_for(gaussians, views, opt, bg):
    """The fused iteration matching the model class (strand model -> FusedStrandStep, Gaussian cloud -> FusedCloudStep)."""
    from scene.hair_gaussian_model import HairGaussianModel
    cls = FusedStrandStep if isinstance(gaussians, HairGaussianModel) else FusedCloudStep
    return cls(gaussians, views, opt, bg)
