"""Training step + loop (counterpart of the reference's train.py:38-265), single GPU or view-parallel.

One iteration (`training_step`) reproduces train.py:133-204: lr update, SH-degree bump every 1000 it, one view,
render -> loss_function (which renders twice more) -> backward, densification statistics / densification /
opacity reset / merging at their intervals, Adam step, zero_grad.

View-parallel extension (not in the reference, which is single-process): every rank draws a DIFFERENT view of the
same global step, gradients of all parameter groups are summed with ONE RCCL all-reduce over a flat fp32 buffer and
averaged, densification statistics are reduced the same way (SUM / MAX), then every rank applies the identical
Adam update, so parameters stay replicated bit for bit.  Topology operators run on identical replicated state with
identical seeds.
"""
import gc
import os
import random

import torch
import torch.distributed as dist

from gaussian_renderer import render
from loss.losses import loss_function, loss_function_single_pass
from scene.hair_gaussian_model import HairGaussianModel

# The cyclic garbage collector's oldest generation holds ~2.8 10^5 objects once torch and this package are imported, and a full
# collection walks all of them: ~100 ms, during which the GPU idles.  The topology operators of a strand model late in training
# (thousands of merge candidates per event) allocate enough containers to trigger one every few events -- 510 ms of a 2.9 s run of
# 2000 Stage-III iterations (tools/dev/gc_pauses.py).  What is alive NOW is the imported libraries, not a model: it moves to the
# permanent generation, which collections skip (gc.freeze: what CPython documents for exactly this); everything created from here
# on -- models, graphs, the operators' temporaries -- is collected as before.  HGS_GC_FREEZE=0 leaves the collector alone.
if os.environ.get("HGS_GC_FREEZE", "1") != "0":
    gc.freeze()


class ViewParallel:
    """Gradient + statistics exchange for view-parallel training (one process per GPU, torch.distributed;
    backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for tests)."""

    def __init__(self, enabled=None):
        self.enabled = dist.is_available() and dist.is_initialized() if enabled is None else enabled
        self.world = dist.get_world_size() if self.enabled else 1
        self.rank = dist.get_rank() if self.enabled else 0
        self._flat = None
        self._flat_key = None
        self._views = []

    def params(self, gaussians):
        return [g["params"][0] for g in gaussians.optimizer.param_groups]

    def graph_collective_ok(self, device=None):
        """May the gradient all-reduce be captured INTO the iteration's HIP graph?  Only with the RCCL backend (a gloo
        collective synchronises with the host), unless switched off (HGS_GRAPH_COLLECTIVE=0), and only if a probe -- a
        two-node graph holding one all-reduce of a small buffer, captured and replayed twice on every rank -- gives the
        right sums here.  Decided once per process, identically on every rank (the probe's verdict is itself reduced)."""
        if self.world == 1:
            return False
        if getattr(self, "_graph_ok", None) is None:
            import os
            ok = dist.get_backend() == "nccl" and os.environ.get("HGS_GRAPH_COLLECTIVE", "1") != "0" and torch.cuda.is_available()
            if ok:
                dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
                # Three agreed stages, so that a rank on which a stage fails never leaves the others waiting inside a collective
                # it will not join: (1) an eager all-reduce on the capture stream (communicator set-up); (2) the capture alone --
                # nothing is enqueued by it -- then the ranks tell each other whether it succeeded (eager MIN on the default
                # stream); (3) only if it did everywhere, two replays whose sums are checked, and a last agreement.
                def agree(flag):
                    verdict = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
                    dist.all_reduce(verdict, op=dist.ReduceOp.MIN)
                    return bool(int(verdict.item()))

                def report(stage, e):
                    if self.rank == 0 or e is not None:
                        print(f"ViewParallel[rank {self.rank}]: all-reduce inside a captured graph is not available "
                              f"({stage}: {type(e).__name__ if e is not None else 'another rank failed'}: {e}); using the eager exchange")

                g = buf = s = None
                try:
                    buf = torch.full((1024,), float(self.rank + 1), device=dev)
                    s = torch.cuda.Stream(device=dev)
                    s.wait_stream(torch.cuda.current_stream(dev))
                    with torch.cuda.stream(s):
                        dist.all_reduce(buf)                  # communicator set-up outside the capture
                    torch.cuda.current_stream(dev).wait_stream(s)
                    torch.cuda.synchronize(dev)
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
                        dist.all_reduce(buf)
                    err = None
                except Exception as e:       # capture of the collective is not supported here: keep the eager exchange
                    err = e
                ok = agree(err is None)
                if not ok:
                    report("capture", err)
                else:
                    n = dist.get_world_size()
                    want = float(n * (n + 1) // 2)
                    good = True
                    try:
                        for _ in range(2):
                            buf.fill_(float(self.rank + 1))
                            g.replay()
                            torch.cuda.synchronize(dev)
                            good = good and bool((buf == want).all())
                        err = None
                    except Exception as e:
                        good, err = False, e
                    ok = agree(good)
                    if not ok:
                        report("replay", err)
            self._graph_ok = ok
        return self._graph_ok

    def pack_gradients(self, gaussians):
        """Gather every parameter's gradient into ONE flat fp32 buffer (a single multi-tensor copy launch) and make the
        parameters' .grad views of it, so the exchange below is one in-place collective and Adam reads the reduced
        values with no copy back.  Layout: [endpoints|f_dc|f_rest|opacity|mask|width] (Stage I: xyz|...|rotation)."""
        ps = [p for p in self.params(gaussians) if p.grad is not None and p.numel() > 0]
        # (an iteration whose topology operators re-created every parameter has no gradient left for this Adam step, as in the
        # reference, where the operators sit between backward and optimizer.step(): nothing to exchange -- on any rank, the
        # replicas being identical)
        self._packed = bool(ps)
        if not ps:
            return
        n = sum(p.numel() for p in ps)
        key = tuple((p.data_ptr(), p.numel()) for p in ps)
        if self._flat is None or self._flat_key != key or self._flat.device != ps[0].device:
            self._flat = torch.zeros(n, dtype=torch.float32, device=ps[0].device)
            self._flat_key = key
            self._views, off = [], 0
            for p in ps:
                self._views.append(self._flat[off:off + p.numel()].view(p.shape))
                off += p.numel()
        srcs = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in ps]
        if not all(s_.data_ptr() == v.data_ptr() for s_, v in zip(srcs, self._views)):
            torch._foreach_copy_(self._views, srcs)
        for p, v in zip(ps, self._views):
            p.grad = v

    def exchange(self, average=True):
        """Average (or, average=False, sum) the packed gradients over the ranks: one all-reduce over xGMI (RCCL) / gloo
        on CPU."""
        if self.world == 1 or self._flat is None or not getattr(self, "_packed", True):
            return
        if average and dist.get_backend() == "nccl":
            dist.all_reduce(self._flat, op=dist.ReduceOp.AVG)
        else:
            dist.all_reduce(self._flat, op=dist.ReduceOp.SUM)
            if average:
                self._flat.div_(self.world)

    def reduce_gradients(self, gaussians):
        """pack + exchange (the eager path; GraphedStep captures the pack and replays around the exchange)."""
        if self.world == 1:
            return
        self.pack_gradients(gaussians)
        self.exchange()

    def reduce_stats(self, gaussians):
        if self.world == 1:
            return
        both = torch.cat([gaussians.xyz_gradient_accum.reshape(-1), gaussians.denom.reshape(-1)])
        dist.all_reduce(both, op=dist.ReduceOp.SUM)
        n = gaussians.xyz_gradient_accum.numel()
        gaussians.xyz_gradient_accum.copy_(both[:n].view_as(gaussians.xyz_gradient_accum))
        gaussians.denom.copy_(both[n:].view_as(gaussians.denom))
        dist.all_reduce(gaussians.max_radii2D, op=dist.ReduceOp.MAX)


class ViewSampler:
    """Pops random cameras without replacement, refilling when empty (train.py:139-143).  With `world` ranks the
    same shuffled order is generated on every rank and rank r takes entries r, r+world, ..."""

    def __init__(self, cameras, seed=0, rank=0, world=1):
        self.cameras, self.rank, self.world = cameras, rank, world
        self.rng = random.Random(seed)
        self.stack = []

    def _pop(self):
        if not self.stack:
            self.stack = list(range(len(self.cameras)))
        return self.stack.pop(self.rng.randint(0, len(self.stack) - 1))

    def next(self):
        picks = [self._pop() for _ in range(self.world)]
        return self.cameras[picks[self.rank]]

    def next_batch(self, total):
        """A global batch of `total` views for one optimizer step (strong scaling): the same draw on every
        rank, rank r takes entries r, r + world, ...  (`total` must be a multiple of the world size)."""
        if total % self.world:
            raise ValueError(f"global batch of {total} views does not divide over {self.world} ranks")
        picks = [self._pop() for _ in range(total)]
        return [self.cameras[i] for i in picks[self.rank::self.world]]


def fused_step_applicable(gaussians, opt):
    """The fused iteration (hgs_runtime.strand_step) covers both models (Stage-I cloud, Stage-III strands) with the
    single-pass rasterizer on the GPU and the reference's default loss terms (no magnet loss)."""
    return (getattr(opt, "fused_step", True) and getattr(opt, "single_pass", True)
            and gaussians.get_xyz.is_cuda and float(getattr(opt, "lambda_magnet", 0.0)) == 0.0)


class _EventInfo:
    """What the operators take as `training_info` (the reference's utils/logging.py TrainingInfo: an object with a
    `densification_info` dict they fill with their counts)."""

    def __init__(self):
        self.densification_info = {}


def training_step(gaussians, viewpoint_cam, opt, bg, iteration, extent=1.0, vp=None, stats_local=None, fused=None,
                  black_background=False, event_log=None):
    """One optimizer step.  Returns (loss tensor (detached, on device), loss_dict, render_pkg).
    In the rasterizer's asynchronous mode (diff_gaussian_rasterization._C.set_async) the three passes never block;
    their instance counts are validated once here, before Adam, and the step is repeated if a pass overflowed.
    `fused`: a hgs_runtime.strand_step.FusedStrandStep whose view table holds `viewpoint_cam` -> the iteration runs as
    one autograd node over the fused kernels (same loss, gradients and statistics).  `black_background=True`: the caller
    states that `bg` is all zero (op-by-op single-pass path: the blend backward's black-background specialisation).
    `event_log`: a list that receives one dict per iteration on which a topology operator ran -- the operators' own counts
    (clone / split / merge_collapsed / prune_* / merge, the reference's TrainingInfo.densification_info) and the primitive
    count before and after; asking for it costs the operators one host synchronisation per count."""
    from diff_gaussian_rasterization import _C as raster
    # once per ITERATION, not per attempt: a repeated attempt (capacity overflow) must not bump the SH degree again
    gaussians.update_learning_rate(iteration)
    if iteration % 1000 == 0:
        gaussians.oneupSHdegree()
    for _attempt in range(4):
        try:
            return _training_step(gaussians, viewpoint_cam, opt, bg, iteration, extent, vp, raster, fused,
                                  black_background, event_log)
        except raster.HgsCapacityOverflow:
            gaussians.optimizer.zero_grad(set_to_none=True)
    raise RuntimeError("rasterizer capacity kept overflowing")


def _training_step(gaussians, viewpoint_cam, opt, bg, iteration, extent, vp, raster, fused=None, black_background=False,
                   event_log=None):
    # The reference ends every iteration with zero_grad(set_to_none=True) (train.py:203): an iteration starts without
    # gradients.  A captured graph leaves its static gradient tensors in `.grad` after a replay -- an eager iteration that
    # follows (topology iterations of training(), bench.py's kernel-timing pass) would ACCUMULATE onto them.
    gaussians.optimizer.zero_grad(set_to_none=True)
    if fused is not None:
        fused.views.prologue(fused.views.index[id(viewpoint_cam)], ride=True)   # (no launch: rides in the forward's first one)
        fused.stats_in_backward = iteration < opt.densify_until_iter
        loss, _ = fused.loss()
        loss_dict = fused.terms()
        render_pkg = {"render": fused.last["planes"][:3], "radii": fused.last["radii"], "viewspace_points": None,
                      "visibility_filter": None}
    elif getattr(opt, "single_pass", True) and gaussians.get_xyz.is_cuda:
        # RGB + mask + orientation in one rasterizer traversal (same loss, 3x fewer raster passes)
        loss, loss_dict, render_pkg = loss_function_single_pass(gaussians, viewpoint_cam, opt, bg, black_background)
    else:
        render_pkg = render(viewpoint_cam, gaussians, bg)
        loss, loss_dict = loss_function(gaussians, render_pkg["render"], viewpoint_cam, opt)
    if fused is not None:
        fused.backward(loss)
    else:
        loss.backward()
    raster.check_async()  # async mode: the step's single synchronisation (raises -> step repeated); no-op otherwise
    densified = False
    info = _EventInfo() if event_log is not None else None
    n_before = int(gaussians.get_xyz.shape[0]) if info is not None else 0
    with torch.no_grad():
        if iteration < opt.densify_until_iter:
            if fused is not None:
                fused.update_densification_stats()
            else:
                gaussians.update_densification_stats(render_pkg["viewspace_points"], render_pkg["radii"],
                                                     render_pkg["visibility_filter"])
            if iteration > opt.densify_from_iter and iteration % opt.densification_interval == 0 \
                    and hasattr(gaussians, "densification") and getattr(opt, "enable_topology", True):
                if vp is not None:
                    vp.reduce_stats(gaussians)
                size_threshold = opt.prune_max_radii_2d if iteration > opt.opacity_reset_interval else None
                gaussians.densification(extent, size_threshold, info)
                densified = True
            if iteration % opt.opacity_reset_interval == 0 and getattr(opt, "enable_topology", True):
                gaussians.reset_opacity()     # (part of the densification schedule, train.py:188-190: off with the operators)
                if info is not None:
                    info.densification_info["opacity_reset"] = 1
        if isinstance(gaussians, HairGaussianModel) and getattr(opt, "enable_topology", True):
            if iteration % opt.merge_interval == 0 and hasattr(gaussians, "merging"):
                gaussians.merging(training_info=info, strands_info_is_current=densified)
            if getattr(gaussians, "_storage_dirty", False):
                gaussians._maybe_sort_spatially()     # back to strand order, once per iteration (scene/hair_gaussian_model.py)
        if info is not None and info.densification_info:
            event_log.append(dict(iteration=int(iteration), primitives_before=n_before,
                                  primitives_after=int(gaussians.get_xyz.shape[0]), **info.densification_info))
        if vp is not None:
            vp.reduce_gradients(gaussians)
        gaussians.optimizer.step()
        gaussians.optimizer.zero_grad(set_to_none=True)
    return loss.detach(), loss_dict, render_pkg


import contextlib


@contextlib.contextmanager
def lean_graph_capture(graph, stream, pool=None, **mode):
    """torch.cuda.graph(...) without its torch.cuda.empty_cache() on entry (3.5 ms, twice per re-capture, i.e. per topology
    event; it only hands cached blocks back to the driver): synchronize, switch to the capture stream, capture_begin /
    capture_end.  (No gc.collect() either, like torch.cuda.graph itself unless torch.compiler.config.force_cudagraph_gc: a full
    collection costs 100-150 ms in a process that holds a training run -- measured: 3000 iterations 3.5 -> 8-10 s.)"""
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        if pool is not None:
            graph.capture_begin(pool, **mode)
        else:
            graph.capture_begin(**mode)
        try:
            yield
        finally:
            graph.capture_end()


_GRAPH_POOLS = {}


def graph_pool(device):
    """ONE allocator pool for every graph this process captures on `device` (torch.cuda.graph_pool_handle).  A graph
    captured into a pool of its own keeps that pool's blocks RESERVED after the graph is dropped -- a private pool's free blocks
    serve no other pool, and nothing returns them to the driver short of empty_cache() -- so a run that re-captures after every
    topology event (100 iterations) reserved another 1-2 GB per event: 24 GB after 1500 iterations at north_star with 1.9 GB
    allocated, and BASELINE config 4's three stages (1 M Gaussians, ~100 re-captures) ran out of the GPU's 288 GB (round 6,
    tools/dev/recapture_memory.py).  With one shared pool the next capture takes the blocks the dropped graph has freed.  Graphs
    that share a pool must not be alive and in use at once unless they share no temporaries in time: GraphedStep's single-step
    and K-step graphs already share one (they are replayed one after the other on one stream and exchange nothing through
    capture-time temporaries), and a re-capture happens only after the previous GraphedStep has been dropped."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _GRAPH_POOLS:
        # A torch.cuda.MemPool OBJECT, kept for the life of the process: the allocator counts it as a user of the pool, so the
        # pool outlives the graphs captured into it (a bare graph_pool_handle() id is retired with its last graph: the next
        # capture_begin on it fails the allocator's use_count assertion)
        with torch.cuda.device(key):
            _GRAPH_POOLS[key] = torch.cuda.MemPool()
    return _GRAPH_POOLS[key].id


def release_graph_pool(device):
    """Drop the shared pool of `device` and hand every cached block back to the driver.  The MemPool object's destructor frees the
    pool's cached blocks (once the last graph captured into it is gone); empty_cache() does the same for the ordinary pools.
    GraphedStep.capture() calls this when the binning capacity has moved to another bucket since the previous capture: the
    workspaces carved for the old capacity (the largest allocations of a pass) fit no request any more, and three classes of
    them linger -- eager topology iterations on the default stream, warm-up passes on the capture stream, the graph's private
    pool (tools/dev/recapture_memory.py: 3 x the sum of all retired bucket sizes, 10 GB after 1500 iterations at north_star)."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    _GRAPH_POOLS.pop(key, None)
    torch.cuda.empty_cache()


_LAST_CAPTURE_CAP = {}
_CAPTURE_STREAMS = {}


def capture_stream(device):
    """The side stream every GraphedStep of this process warms up and captures on (one per device)."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _CAPTURE_STREAMS:
        _CAPTURE_STREAMS[key] = torch.cuda.Stream(device=key)
    return _CAPTURE_STREAMS[key]


class GraphedStep:
    """The whole training iteration (3 raster fwd+bwd, losses, statistics, Adam) captured ONCE into a HIP graph and
    replayed per step: ~100 kernel launches cost one graph launch on the host, so the step runs at GPU speed instead
    of Python-dispatch speed.  Requirements, all met by the library: no host synchronisation inside the step (the
    rasterizer's async capacity mode), every per-step input in device memory (the camera is a static slot that
    `step()` refreshes with device-to-device copies; learning rates are device tensors), in-place statistics.
    Re-capture (`capture()`) after anything that changes shapes: densification, SH-degree bump, opacity reset.
    Multi-GPU: gradients are exchanged between two graphs (fwd/bwd | eager RCCL all-reduce | Adam)."""

    CAMERA_FIELDS = ("world_view_transform", "full_proj_transform", "camera_center", "original_image", "mask",
                     "float_mask", "orientation_field", "orientation_confidence")

    def __init__(self, gaussians, cameras, opt, bg, extent=1.0, vp=None, slack=2.0, views=None, views_per_step=1,
                 steps_per_graph=1):
        """views_per_step > 1: every rank renders that many views per optimizer step inside the one captured graph and
        sums their gradients; with W ranks the step's gradient is the MEAN over the views_per_step x W views of the
        global batch (strong-scaling protocol: the batch is fixed, the ranks share it)."""
        import copy
        from diff_gaussian_rasterization import _C as raster
        self.g, self.opt, self.bg, self.extent, self.raster = gaussians, opt, bg, extent, raster
        # the background is a constant of the captured step: looked at once, here, before anything is captured
        self.black_background = bool((bg == 0).all())
        self.vp = vp if vp is not None else ViewParallel()
        self.fused = None
        if fused_step_applicable(gaussians, opt):
            # device-resident view table: a view switch is one tiny launch, not eight tensor copies
            from hgs_runtime.strand_step import ViewTable, fused_step_for
            self.fused = fused_step_for(gaussians, views if views is not None else ViewTable(cameras), opt, bg)
            # forward and backward always run together here: no launch of its own for the loss head's last sums
            self.fused.defer_tail = bool(getattr(opt, "defer_head_tail", True))
        # the iteration prologue rides in the first launch of the fused iteration instead of being one (A/B switch)
        self._ride = self.fused is not None and bool(getattr(opt, "ride_prologue", True))
        c0 = cameras[0]
        for c in cameras:  # by-value kernel arguments are frozen into the graph
            assert (c.image_width, c.image_height, c.FoVx, c.FoVy) == (c0.image_width, c0.image_height, c0.FoVx, c0.FoVy)
        self.slot = copy.copy(c0)
        for f in self.CAMERA_FIELDS:
            v = getattr(c0, f, None)
            if torch.is_tensor(v):
                setattr(self.slot, f, v.clone())
        raster.set_async(True, slack=slack)
        self._graphs = None
        self._prologue_in_graph = False
        self._binding = self._many = None
        self._cap = None
        self.views_per_step = int(views_per_step)
        # steps_per_graph > 1: a second graph holds that many consecutive optimizer steps (step_many): one graph launch costs
        # ~8 us of idle GPU between two replays whatever the graph holds, so K steps per launch save (1 - 1/K) of that
        self.steps_per_graph = int(steps_per_graph)
        if self.views_per_step > 1 and self.fused is None:
            raise ValueError("several views per captured step need the fused iteration (hgs_runtime.strand_step)")
        # several ranks: capture the gradient all-reduce and Adam into the step's graph when the backend allows (RCCL)
        self.collective_in_graph = bool(getattr(opt, "collective_in_graph", True))
        self.collective_captured = False
        # Adam in the backward's own lanes (hgs_runtime.strand_step.FusedStrandStep.enable_inline_adam): one rank, one view per
        # step, the fused iteration -- the captured step then holds no optimizer launch
        self.inline_adam = False
        self._make_capturable()

    def _make_capturable(self):
        opt_ = self.g.optimizer
        dev = self.g.get_xyz.device
        for group in opt_.param_groups:
            if isinstance(opt_, torch.optim.Adam):
                group["capturable"] = True
            if not torch.is_tensor(group["lr"]):
                group["lr"] = torch.tensor(float(group["lr"]), dtype=torch.float32, device=dev)
            for p in group["params"]:
                st = opt_.state[p]
                if len(st) == 0:
                    # create the Adam state NOW (exactly what the first optimizer.step() would create): state created
                    # lazily inside the capture would be re-zeroed by every replay
                    st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                elif torch.is_tensor(st.get("step")) and not st["step"].is_cuda:
                    st["step"] = st["step"].to(dev)

    def _position_lr(self):
        for group in self.g.optimizer.param_groups:
            if group["name"] == self.g._POSITION_GROUP:
                return group["lr"]
        return None

    def _set_lr(self, iteration):
        g = self.g
        lr = g.xyz_scheduler_args(iteration)
        self._lr_now = float(lr)
        if self.fused is None:  # the fused path writes the position lr with the view-select launch
            self._position_lr().fill_(float(lr))
        if hasattr(g, "merge_dist_th_scheduler"):
            g.merge_dist_th = g.merge_dist_th_scheduler(iteration)
            g.merge_angle_th = g.merge_angle_th_scheduler(iteration)

    def load_camera(self, cam):
        if self.fused is not None:
            v = self.fused.views
            if self._prologue_in_graph and not torch.cuda.is_current_stream_capturing() and self._graphs is not None:
                v.graph_set(self._binding, v.index[id(cam)], lr=self._lr_now, lr_dst=self._position_lr())   # host work only: no launch
            else:
                v.prologue(v.index[id(cam)], lr=self._lr_now, lr_dst=self._position_lr(), ride=self._ride)
            return
        for f in self.CAMERA_FIELDS:
            dst, src = getattr(self.slot, f, None), getattr(cam, f, None)
            if torch.is_tensor(dst) and torch.is_tensor(src):
                dst.copy_(src, non_blocking=True)
        self.slot.uid = cam.uid

    def _forward_backward_queue(self):
        """views_per_step views, selected on the device from the view queue; gradients accumulate in .grad (autograd's
        AccumulateGrad), statistics per view.  Returns the mean loss of the local views."""
        total = None
        for k in range(self.views_per_step):
            self.fused.views.select_queued(k, lr_dst=self._position_lr() if k == 0 else None)
            loss, _ = self.fused.loss()
            self.fused.backward(loss)
            self.fused.update_densification_stats()
            total = loss.detach().clone() if total is None else total + loss.detach()
        return total / self.views_per_step

    def _scale_gradients(self):
        """.grad <- .grad / (views of the global batch): one multi-tensor launch (captured in front of Adam)."""
        grads = [p.grad for p in self.vp.params(self.g) if p.grad is not None]
        if grads:
            torch._foreach_mul_(grads, 1.0 / (self.views_per_step * self.vp.world))

    def _forward_backward(self):
        if self.fused is not None:
            loss, _ = self.fused.loss()
            self.fused.backward(loss)
            self.fused.update_densification_stats()
            return loss.detach()
        if getattr(self.opt, "single_pass", True):
            loss, _, pkg = loss_function_single_pass(self.g, self.slot, self.opt, self.bg, self.black_background)
        else:
            pkg = render(self.slot, self.g, self.bg)
            loss, _ = loss_function(self.g, pkg["render"], self.slot, self.opt)
        loss.backward()
        with torch.no_grad():
            self.g.update_densification_stats(pkg["viewspace_points"], pkg["radii"], pkg["visibility_filter"])
        return loss.detach()

    def capture(self, warmup_cams, iteration=1):
        """Warm up eagerly on a side stream (allocator, lazy module loads, capacity), then capture."""
        g, raster = self.g, self.raster
        self._make_capturable()
        self._graphs = None            # (load_camera launches the prologue until the new graph exists)
        saved_stats = (g.max_radii2D.clone(), g.xyz_gradient_accum.clone(), g.denom.clone())  # warm-up must not count
        # warm-up AND capture run on one side stream (AccumulateGrad nodes are per stream) -- the SAME one for every capture of the
        # process: the caching allocator keeps freed blocks per stream, so a new stream per re-capture left the previous
        # capture's warm-up buffers (binning, scratch: 1-2 GB at 250 k segments) cached where nothing would ever ask for them
        # again (tools/dev/recapture_memory.py: reserved memory grew by that much per topology event)
        dev_key = g.get_xyz.device.index
        s = self._stream = capture_stream(g.get_xyz.device)
        s.wait_stream(torch.cuda.current_stream())
        warm_R = 0
        with torch.cuda.stream(s):
            for cam in warmup_cams:  # every view once: the capacity must cover the busiest camera
                self._set_lr(iteration)
                self.load_camera(cam)
                self._forward_backward()
                if self.vp.world > 1:
                    self.vp.pack_gradients(g)   # creates the flat exchange buffer outside the capture
                g.optimizer.zero_grad(set_to_none=True)
                try:
                    raster.check_async()        # learns the capacity; an overflow here only raises it for the capture
                except raster.HgsCapacityOverflow:
                    pass
                warm_R = max(warm_R, int(raster._state.get("last_exact_R", 0)) & 0x7FFFFFFF)
        # how the single-pass backward sums its instance rows (include/hgs.h hgs_set_row_reduce) follows the MODEL: the instance
        # counts the warm-up has just measured, not the capacity -- for this capture, its replays and the eager iterations until
        # the next capture (every rank sees the same model and the same warm-up views)
        raster.set_row_reduce(warm_R >= 4 * int(g.get_xyz.shape[0]))
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g.optimizer.zero_grad(set_to_none=True)
        if _LAST_CAPTURE_CAP.get(dev_key) not in (None, raster._state["cap"]):
            # the capacity (as the warm-up has just settled it) is in another bucket than the previous capture's: nothing the
            # allocator has cached for the old one fits any more.  Handing it back costs the next passes their hipMallocs
            # (tens of ms per event: tools/soak.py 2.4 -> 3.1 s when done at every bucket change), so it waits until the
            # cache holds a quarter of the device's memory -- never on a 100 k - 500 k-segment run, every few events at 1 M
            free_b, total_b = torch.cuda.mem_get_info(g.get_xyz.device)
            if torch.cuda.memory_reserved(g.get_xyz.device) > float(getattr(self.opt, "release_cache_fraction", 0.25)) * total_b:
                release_graph_pool(g.get_xyz.device)
        for dst, src in zip((g.max_radii2D, g.xyz_gradient_accum, g.denom), saved_stats):
            dst.copy_(src)
        self.loss_buf = None
        multi = self.views_per_step > 1
        if multi:
            self.fused.views.set_queue([0] * self.views_per_step, lr=self._lr_now)   # (the capture itself launches nothing)
        fwd_bwd = self._forward_backward_queue if multi else self._forward_backward
        # one view per step on the fused path: the view select (+ image-buffer clearing) is the graph's first node, and
        # step() re-points it by updating that node's arguments (no launch between two replays)
        self._prologue_in_graph = self.fused is not None and not multi
        # With several ranks the gradient exchange is part of the step.  RCCL collectives can be captured into the graph
        # (ViewParallel.graph_collective_ok probes it): the whole step -- prologue, forward/backward, packing, all-reduce,
        # Adam -- is then ONE graph, and several optimizer steps per launch work across ranks exactly as on one.  Otherwise
        # (gloo on CPU-side tests, HGS_GRAPH_COLLECTIVE=0) the graph ends with the packing and step() issues the exchange
        # and Adam eagerly behind it.
        world = self.vp.world
        in_graph = world == 1 or (self.collective_in_graph and self.vp.graph_collective_ok())
        # a live RCCL communicator has a watchdog thread that polls events: with the default (global) capture error mode
        # its calls would invalidate the capture; only this thread's unsafe calls must be errors
        mode = dict(capture_error_mode="thread_local") if world > 1 else {}

        def exchange_and_step():
            if world > 1:
                self.vp.pack_gradients(g)       # .grad become views of the flat exchange buffer
                self.vp.exchange(average=not multi)
            if multi:
                self._scale_gradients()
            g.optimizer.step()

        # (the warm-up above ran with the optimizer untouched; the captured iterations update in the backward's lanes where possible)
        self.inline_adam = (self.fused is not None and not multi and world == 1 and self.fused.enable_inline_adam(True))
        # (the graphs bake the plan's device tables into their nodes: they stay alive as long as the graphs do, whatever the
        # optimizer does with its own reference)
        self._plan_keep = self.fused.inline_plan() if self.inline_adam else None
        ga = torch.cuda.CUDAGraph(keep_graph=True) if self._prologue_in_graph else torch.cuda.CUDAGraph()
        pool = graph_pool(g.get_xyz.device)
        with lean_graph_capture(ga, s, pool=pool, **mode):
            if self._prologue_in_graph:
                self.load_camera(warmup_cams[0])
            self.loss_buf = fwd_bwd()
            if in_graph:
                exchange_and_step()
            else:
                self.vp.pack_gradients(g)
        # "eager-tail": what follows the packing -- the all-reduce, (the gradient scale and) ONE Adam launch -- is issued by
        # step(): a graph of one or two kernels costs more GPU-idle time per launch (~8 us) than the launches it saves
        # (tools/probes/graph_gap.py: 2 kernels, 16.6 us per replay against 10.0 us eager).
        self._graphs = (ga, None if in_graph else "eager-tail")
        self.collective_captured = world > 1 and in_graph
        if self._prologue_in_graph:
            ga.instantiate()
            self._binding = self.fused.views.graph_bind(ga)
        self._many = None
        if self.steps_per_graph > 1:
            if not self._prologue_in_graph or not in_graph:
                raise ValueError("steps_per_graph > 1 needs the fused iteration, one view per step and a step that is one "
                                 "graph (one rank, or the all-reduce captured with RCCL)")
            K, v = self.steps_per_graph, self.fused.views
            gk = torch.cuda.CUDAGraph(keep_graph=True)
            losses = []
            with lean_graph_capture(gk, s, pool=pool, **mode):
                for j in range(K):
                    g.optimizer.zero_grad(set_to_none=True)   # (host side: this step's backward ASSIGNS its gradients)
                    v.prologue(j % v.n, lr=float(j), lr_dst=self._position_lr(), ride=self._ride)   # lr = j: the tag graph_bind sorts by
                    losses.append(fwd_bwd())
                    exchange_and_step()
            gk.instantiate()
            self._many = (gk, v.graph_bind(gk, K), losses)
        if self.fused is not None:
            self.fused.enable_inline_adam(False)      # (the graphs keep what they captured; eager users of these views do not inherit it)
        # every replay raises the library's sticky device-side maximum of num_rendered; check() compares it with the
        # capacity the captured passes were built for
        self._cap = raster._state["cap_used"]
        _LAST_CAPTURE_CAP[dev_key] = raster._state["cap"]
        for t in raster._state["max_R"].values():
            t.zero_()                       # the capture itself launched nothing
        raster._state["dirty"] = False
        raster._state["cap_used"] = None

    def step(self, cam, iteration):
        """One optimizer step on `cam` (views_per_step > 1: on the list of this rank's views of the global batch);
        returns the (device) loss of this step."""
        self._set_lr(iteration)
        if self.views_per_step > 1:
            cams = list(cam)
            if len(cams) != self.views_per_step:
                raise ValueError(f"captured for {self.views_per_step} views per step, got {len(cams)}")
            v = self.fused.views
            v.set_queue([v.index[id(c)] for c in cams], lr=self._lr_now)
        else:
            self.load_camera(cam)
        ga, gb = self._graphs
        if self.fused is not None:
            self.fused.views.ensure_counts_clean()     # (a no-op unless a blocking-mode pass used the views' image buffer)
        ga.replay()
        if gb is not None:
            # eager behind the graph: one in-place all-reduce (mean over the ranks; with several views per rank a sum,
            # scaled to the mean over the global batch in front of Adam), then Adam
            self.vp.exchange(average=self.views_per_step == 1)
            if self.views_per_step > 1:
                self._scale_gradients()
            self.g.optimizer.step()
        return self.loss_buf

    def step_many(self, cams, iteration):
        """steps_per_graph optimizer steps -- iterations `iteration`, `iteration` + 1, ... on the views `cams` -- with ONE
        graph launch; returns the list of their (device) losses.  Same arithmetic as that many step() calls."""
        cams = list(cams)
        if self._many is None or len(cams) != self.steps_per_graph:
            raise ValueError(f"captured for {self.steps_per_graph} steps per graph, got {len(cams)}")
        gk, binding, losses = self._many
        v = self.fused.views
        for j, cam in enumerate(cams):
            self._set_lr(iteration + j)
            v.graph_set(binding, v.index[id(cam)], lr=self._lr_now, lr_dst=self._position_lr(), k=j)
        v.ensure_counts_clean()
        gk.replay()
        return losses

    def headroom(self):
        """(largest num_rendered of the replays since the last check, captured capacity): one synchronisation, no
        exception -- lets a long run re-capture BEFORE a growing model overflows the captured binning capacity."""
        raster = self.raster
        worst = 0
        for t in raster._state["max_R"].values():
            worst = max(worst, int(t.item()) & 0xFFFFFFFF)   # (the word is unsigned)
            t.zero_()
        if worst == 0xFFFFFFFF:   # include/hgs.h HGS_WAIT_TIMED_OUT
            raise self.raster.rt.HgsError("a replayed raster pass gave up an inter-workgroup wait: its frame is invalid")
        return worst, self._cap

    def check(self):
        """Synchronise and validate the instance counts of ALL replays since the last check (raises on capacity
        overflow); returns [largest num_rendered seen]."""
        raster = self.raster
        worst, _ = self.headroom()
        if self._cap is not None and worst > self._cap:
            raise raster.HgsCapacityOverflow(
                f"captured step needed {worst} instances > capacity {self._cap}: re-capture with a larger slack")
        return [worst]


def topology_due(gaussians, opt, iteration):
    """Which shape-changing operators train.py:172-200 schedules at this iteration."""
    due = []
    if iteration < opt.densify_until_iter:
        if iteration > opt.densify_from_iter and iteration % opt.densification_interval == 0:
            due.append("densify")
        if iteration % opt.opacity_reset_interval == 0:
            due.append("reset_opacity")
    if isinstance(gaussians, HairGaussianModel) and iteration % opt.merge_interval == 0:
        due.append("merge")
    if iteration % 1000 == 0 and gaussians.active_sh_degree < gaussians.max_sh_degree:
        due.append("sh")
    return due


def training(gaussians, cameras, opt, iterations=None, extent=1.0, seed=0, log_every=0, vp=None, start_iteration=0,
             use_graph=True, sampler=None, steps_per_graph=8, event_log=None):
    """Training loop (reference train.py:91-254 without logger / viewer / dataset IO, which are outside the
    accelerated path).  With use_graph the iteration body replays a captured HIP graph (GraphedStep).  Iterations on which
    the reference schedules a shape-changing operator (densification, merging, opacity reset, SH-degree bump:
    train.py:136-200) run EAGERLY through `training_step`, which has the reference's order -- operators between backward
    and optimizer.step(), so re-created tensors skip that Adam step exactly as in the reference --, and the graph is
    captured again afterwards.  Runs of steps_per_graph plain iterations go out as ONE graph launch (GraphedStep.step_many;
    single rank, fused iteration).  `sampler`: a ViewSampler to continue (main() trains in chunks between saves).
    `event_log`: see training_step (per-event operator counts)."""
    vp = ViewParallel() if vp is None else vp
    dev = gaussians.get_xyz.device
    bg = torch.zeros(3, dtype=torch.float32, device=dev)
    sampler = ViewSampler(cameras, seed=seed, rank=vp.rank, world=vp.world) if sampler is None else sampler
    ema = None
    n = opt.iterations if iterations is None else iterations
    gs = None
    use_graph = use_graph and dev.type == "cuda"
    views = fused = None
    if dev.type == "cuda" and fused_step_applicable(gaussians, opt):
        import hgs_runtime as rt
        from hgs_runtime.strand_step import ViewTable, fused_step_for
        try:
            views = ViewTable(cameras)             # built once; survives topology changes
        except (rt.HgsError, AttributeError) as e:
            # a capture the view table cannot hold (views of different sizes, a view without orientation maps, masks on some
            # views only): the op-by-op iteration takes every camera as it is, eagerly -- slower, same arithmetic
            if vp.rank == 0:
                print(f"training(): the fused iteration needs uniform views ({e}); running the op-by-op iteration eagerly")
            views, use_graph = None, False
    if views is not None:
        fused = fused_step_for(gaussians, views, opt, bg)   # eager launches of the same iteration (topology iterations)
        fused.defer_tail = bool(getattr(opt, "defer_head_tail", True))   # (training_step runs forward and backward together)
    topology = getattr(opt, "enable_topology", True)
    rollbacks = 0
    ckpt = None             # _Checkpoint of the last state known to be exact (captured-graph mode only)

    def overflowed(gs_):
        """(worst instance count since the last check over all ranks, captured capacity): one synchronisation."""
        worst, cap = gs_.headroom()
        if vp.world > 1:
            w = torch.tensor([worst], dtype=torch.int64, device=dev)
            dist.all_reduce(w, op=dist.ReduceOp.MAX)
            worst = int(w.item())
        return worst, cap

    def roll_back(worst, cap, it_now):
        """A replayed pass needed more instances than the graph was captured for: its gradients were zero (include/hgs.h),
        so the steps since the last check are not what the reference computes (train.py:146-204: every step has its
        gradient).  Return to the checkpoint, raise the capacity and run those iterations again."""
        from diff_gaussian_rasterization import _C as raster
        if vp.rank == 0:
            print(f"[it {it_now}] binning capacity {cap} exceeded ({worst} instances): iterations {ckpt.it + 1}..{it_now} "
                  "are run again from the last checkpoint with a larger capacity")
        raster._state["cap"] = max(raster._state["cap"], raster.bucket_capacity(int(worst * 2.0) + 4096))
        return ckpt.restore(gaussians, sampler)

    try:
        it, last = start_iteration + 1, start_iteration + n
        while it <= last:
            due = topology_due(gaussians, opt, it) if topology else []
            if use_graph and not due:
                if gs is None:
                    many = steps_per_graph if (fused is not None and (vp.world == 1 or (
                        getattr(opt, "collective_in_graph", True) and vp.graph_collective_ok()))) else 1
                    # A K-step graph saves (1 - 1/K) of the ~8 us a graph launch idles the GPU (~7 us per iteration at K = 8) and
                    # costs K more captured iterations (~1.3 ms each) per re-capture: it pays for runs of more than ~1500 plain
                    # iterations.  Between two topology events of the default schedule (100 iterations) it does not: 10 ms per event.
                    if many > 1:
                        need, run = int(getattr(opt, "multi_step_graph_min_run", 1500)), 0
                        while it + run <= last and run < need and not (topology and topology_due(gaussians, opt, it + run)):
                            run += 1
                        if run < need:
                            many = 1
                    gs = GraphedStep(gaussians, cameras, opt, bg, extent=extent, vp=vp, views=views, steps_per_graph=many,
                                     slack=float(getattr(opt, "capacity_slack", 2.0)))
                    # warm-up on a spread of the views (allocator, lazy loads, a capacity estimate): a view that needs more than
                    # slack x their largest instance count is caught by the headroom check and rolled back exactly, so the
                    # capture need not render every camera first (32 eager iterations per re-capture at north_star)
                    nw = min(len(cameras), int(getattr(opt, "capture_warmup_views", 2)))
                    gs.capture([cameras[(k * len(cameras)) // nw] for k in range(nw)], iteration=it)
                    ckpt = _Checkpoint(gaussians, sampler, ema, it - 1)    # (eager steps are exact: they repeat on overflow)
                K = gs.steps_per_graph
                if K > 1 and it + K - 1 <= last and not (topology and any(topology_due(gaussians, opt, j) for j in range(it + 1, it + K))):
                    losses = gs.step_many([sampler.next() for _ in range(K)], it)   # K optimizer steps, one graph launch
                else:
                    losses = [gs.step(sampler.next(), it)]
            else:
                if gs is not None:
                    # leaving the captured graph: every replay since the last check must have been exact
                    worst, cap = overflowed(gs)
                    gs = None                  # shapes change below: capture again at the next iteration
                    if cap is not None and worst > cap:
                        it, ema = roll_back(worst, cap, it - 1)
                        rollbacks += 1
                        continue
                # only the (detached) loss is kept: the terms and the render package of an op-by-op iteration hold its autograd
                # graph -- and with it the parameters' AccumulateGrad nodes, created on THIS stream -- alive, which breaks the
                # next capture on the side stream (stream mismatch; observed as a crash in capture_end)
                losses = [training_step(gaussians, sampler.next(), opt, bg, it, extent=extent, vp=vp, fused=fused,
                                        black_background=True, event_log=event_log)[0]]
                if fused is not None and due:
                    fused.refresh()
            first, it = it, it + len(losses)
            for loss in losses:
                ema = loss.clone() if ema is None else 0.4 * loss + 0.6 * ema  # on the device: no per-iteration host sync
            if gs is not None and (it > last or any(j % 64 == 0 for j in range(first, it))):
                # The model grows while it trains.  Every 64 iterations (and at the end of the run): if a replay overflowed
                # the captured capacity, roll back to the last checkpoint and run the iterations again; otherwise this
                # state is exact -- checkpoint it -- and the graph is re-captured with a larger capacity once 80 % of the
                # captured one is in use.  Decisions are taken on the maximum over the ranks (replicas act together).
                worst, cap = overflowed(gs)
                if cap is not None and worst > cap:
                    gs = None
                    it, ema = roll_back(worst, cap, it - 1)
                    rollbacks += 1
                    continue
                ckpt.take(gaussians, sampler, ema, it - 1)
                if cap is not None and worst > 0.8 * cap:
                    from diff_gaussian_rasterization import _C as raster
                    raster._state["cap"] = max(raster._state["cap"], raster.bucket_capacity(int(worst * 2.0) + 4096))
                    gs = None
            if log_every and vp.rank == 0 and any(j % log_every == 0 for j in range(first, it)):
                print(f"[it {it - 1}] loss(ema) {float(ema):.6f}  segments {gaussians.get_xyz.shape[0]}")
    finally:
        if use_graph:
            from diff_gaussian_rasterization import _C as raster
            raster.set_async(False)
            raster.set_row_reduce(None)       # (what the captures of this run chose: back to the per-call default)
    training.last_void_steps = 0          # (kept for callers of earlier rounds: overflowed steps are now run again)
    training.last_rollbacks = rollbacks
    return ema


class _Checkpoint:
    """Parameters, Adam state (moments and step counters), densification statistics, the view sampler and the loss average
    at an iteration whose state is known to be exact: what training() returns to when replays of the captured graph ran
    past its binning capacity.  One multi-tensor copy per take / restore (~120 B per Gaussian, every 64 iterations)."""

    def __init__(self, gaussians, sampler, ema, it):
        self._src = self._tensors(gaussians)
        self._buf = [torch.empty_like(t) for t in self._src]
        self.take(gaussians, sampler, ema, it)

    @staticmethod
    def _tensors(g):
        ts = []
        for group in g.optimizer.param_groups:
            for p in group["params"]:
                ts.append(p.data)
                ts += [v for v in g.optimizer.state.get(p, {}).values() if torch.is_tensor(v)]
        return ts + [g.max_radii2D, g.xyz_gradient_accum, g.denom]

    def take(self, gaussians, sampler, ema, it):
        with torch.no_grad():
            torch._foreach_copy_(self._buf, self._src)
        self.it = it
        self.sampler = (sampler.rng.getstate(), list(sampler.stack))
        self.ema = None if ema is None else ema.clone()

    def restore(self, gaussians, sampler):
        """Puts the state back; returns (next iteration, loss average)."""
        with torch.no_grad():
            torch._foreach_copy_(self._src, self._buf)
        sampler.rng.setstate(self.sampler[0])
        sampler.stack = list(self.sampler[1])
        return self.it + 1, (None if self.ema is None else self.ema.clone())


def main(argv=None):
    """Command-line driver (reference train.py:38-131, 256-265 without logger / viewer / evaluation):
      python train.py -s <colmap scene> -m <model dir> [--iterations N] [--save_frequency K]
    Resumes from the newest <model dir>/point_cloud/iteration_*/point_cloud.ply (Gaussian cloud = Stage I, strand model =
    Stage III) or starts Stage I from the sparse COLMAP points; saves every save_frequency iterations and at the end."""
    import os
    import sys
    from argparse import ArgumentParser
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from arguments import GeneralParams, ModelParams, OptimizationParams
    from scene import Scene
    from utils.general import safe_state
    parser = ArgumentParser(description="Training script parameters")
    mp, op, gp = ModelParams(parser), OptimizationParams(parser), GeneralParams(parser)
    args = parser.parse_args(argv)
    # view-parallel run (python -m torch.distributed.run ... train.py): one process per GPU, initialised before anything
    # touches the GPU; every rank seeds identically (replicated topology operators draw the same random numbers), rank 0
    # alone writes to the model directory
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        local = int(os.environ.get("LOCAL_RANK", "0"))
        # (HGS_DIST_BACKEND=gloo: ranks may share a GPU -- RCCL needs one per rank --, which is how the tests run this path
        # on a one-GPU box)
        backend = os.environ.get("HGS_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            local = local % torch.cuda.device_count() if backend == "gloo" else local
            torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend)
    safe_state(args.quiet or rank != 0)
    if rank == 0:
        os.makedirs(args.model_path, exist_ok=True)
        with open(os.path.join(args.model_path, "cfg_args"), "w") as fh:   # what render.py's get_combined_args reads back
            fh.write(str(args))
    # Every rank loads the capture at the same time: the one shared file a first run creates -- sparse/0/points3D.ply -- is
    # written under a private name and renamed into place (data/dataset_readers.py: whoever converts, every reader sees a
    # complete file), and the model directory's input.ply / cameras.json are rank 0's alone (scene/scene.py).
    scene = Scene(mp.extract(args))
    if world > 1:
        dist.barrier()
    opt = op.extract(args)
    g = scene.gaussians
    g.training_setup(opt)
    cams = scene.getCameras()
    vp = ViewParallel()
    sampler = ViewSampler(cams, seed=0, rank=vp.rank, world=vp.world)   # ONE view order for the whole run
    # As in the reference (train.py:91-92, 251-253), `iteration` counts from 1 in EVERY invocation -- the learning-rate
    # schedule, the SH-degree bumps and the operators' intervals restart with each stage of the workflow -- and only the
    # name of a saved iteration adds what the model had when it was loaded.
    base, total, every = scene.loaded_iter, int(opt.iterations), max(1, int(args.save_frequency))
    eval_every = max(1, int(getattr(args, "eval_frequency", 30000)))

    def evaluate(iteration):
        """The reference's evaluation hook (train.py:229-246): strand metrics against the capture's ground truth, where it
        has one (hair_eval_data.npz), every eval_frequency iterations and at the end.  CPU code (cKDTree), rank 0 only."""
        import numpy as np
        from loss.metrics import compute_eval_data_from_gs, compute_eval_data_from_hair_gs, compute_metrics
        from scene.hair_gaussian_model import HairGaussianModel
        pred = compute_eval_data_from_hair_gs(g) if isinstance(g, HairGaussianModel) else compute_eval_data_from_gs(g)
        scene.eval_metrics, scene.eval_thresholds = compute_metrics(pred=pred, gt=scene.gt, bidirectional=bool(opt.bidirectional_eval))
        if not args.quiet:
            print(f"[it {iteration}] " + "  ".join(f"{k} {np.round(np.asarray(v), 3).tolist()}" for k, v in scene.eval_metrics.items())
                  + f"  at {scene.eval_thresholds}")

    it = 0
    while it < total:
        n = min(every - it % every, total - it)
        if scene.gt is not None:
            n = min(n, eval_every - it % eval_every)
        ema = training(g, cams, opt, iterations=n, extent=scene.cameras_extent, start_iteration=it, vp=vp, sampler=sampler,
                       log_every=0 if (args.quiet or rank != 0) else max(1, n // 4))
        it += n
        if rank == 0 and scene.gt is not None and (it % eval_every == 0 or it == total):
            evaluate(base + it)
        if rank == 0 and (it % every == 0 or it == total):
            scene.save(it)                  # (Scene.save adds the loaded iteration, like the reference's)
            if not args.quiet:
                print(f"[it {base + it}] saved; loss(ema) {float(ema):.6f}")
        if world > 1:
            dist.barrier()
    if world > 1:
        dist.destroy_process_group()
    return scene


if __name__ == "__main__":
    main()
