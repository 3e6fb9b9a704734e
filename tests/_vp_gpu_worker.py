"""Worker of tests/test_gpu_view_parallel.py: N ranks run the view-parallel GraphedStep; every rank must end with
bit-identical parameters, and rank 0 checks them against a single-process run that averages the same per-view gradients by
hand.  HGS_VP_MODE=strong: a fixed global batch of HGS_VP_GLOBAL_VIEWS views per optimizer step shared by the ranks
(several views per rank inside one captured graph, bench.py --scaling strong).
HGS_VP_BACKEND=gloo (default): the ranks SHARE one GPU (what a one-GPU box can run: everything but the transport).
HGS_VP_BACKEND=nccl: ONE RANK PER DEVICE over RCCL / xGMI (needs as many devices as ranks) -- additionally checks that the
all-reduce was captured INTO the step's graph (HGS_VP_STEPS_PER_GRAPH optimizer steps per launch) and that one averaged
gradient exchange equals rank 0's own mean of every rank's gradient to 1e-5 relative."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import torch
import torch.distributed as dist


def main():
    rccl = os.environ.get("HGS_VP_BACKEND", "gloo") == "nccl"
    dev_index = int(os.environ.get("LOCAL_RANK", "0")) if rccl else 0
    torch.cuda.set_device(dev_index)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        if rccl:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend="gloo")
    distributed = dist.is_initialized()
    rank, world = (dist.get_rank(), dist.get_world_size()) if distributed else (0, 1)
    from arguments import OptimizationParams
    from diff_gaussian_rasterization import _C as raster
    from synthetic import build_workload
    from train import GraphedStep, ViewParallel, ViewSampler
    from utils.general import safe_state
    safe_state(True)
    model, cams, extent = build_workload("tiny", device="cuda", with_targets=True)
    opt = OptimizationParams()
    opt.enable_topology = False
    model.training_setup(opt)
    bg = torch.zeros(3, device="cuda")
    vp = ViewParallel()
    assert vp.world == world
    sampler = ViewSampler(cams, seed=0, rank=vp.rank, world=vp.world)
    strong = os.environ.get("HGS_VP_MODE", "weak") == "strong"
    V = int(os.environ.get("HGS_VP_GLOBAL_VIEWS", str(world)))
    per_rank = V // world if strong else 1
    spg = int(os.environ.get("HGS_VP_STEPS_PER_GRAPH", "1"))
    gs = GraphedStep(model, cams, opt, bg, extent=extent, vp=vp, views_per_step=per_rank, steps_per_graph=spg)
    gs.capture(cams)
    if rccl and world > 1:
        assert gs.collective_captured is True, "the RCCL all-reduce was not captured into the step's graph"
        assert spg == 1 or strong or gs.steps_per_graph == spg, (gs.steps_per_graph, spg)
    picks = []
    if spg > 1 and not strong and gs.steps_per_graph == spg == 4:
        mine4 = [sampler.next() for _ in range(4)]
        picks = [[cams.index(c)] for c in mine4]
        loss = gs.step_many(mine4, 1)
    else:
        for it in range(1, 5):
            mine = sampler.next_batch(V) if strong else [sampler.next()]
            assert len(mine) == per_rank
            picks.append([cams.index(c) for c in mine])
            loss = gs.step(mine if per_rank > 1 else mine[0], it)
    gs.check()
    raster.set_async(False)
    assert torch.isfinite(loss)
    flat = torch.cat([p.detach().reshape(-1) for p in vp.params(model)])
    if not rccl:
        flat = flat.cpu()
    gathered = [torch.empty_like(flat) for _ in range(world)]
    allpicks = [None] * world
    if distributed:
        dist.all_gather(gathered, flat)
        dist.all_gather_object(allpicks, picks)
    else:
        gathered, allpicks = [flat], [picks]
    gathered = [g.cpu() for g in gathered]
    for r in range(1, world):
        assert torch.equal(gathered[0], gathered[r]), f"rank {r} diverged from rank 0"
    if rank == 0:
        # single-process reference: same views, gradients averaged by hand, same Adam
        from hgs_runtime.strand_step import FusedStrandStep
        safe_state(True)
        ref, rcams, _ = build_workload("tiny", device="cuda", with_targets=True)
        ref.training_setup(opt)
        fused = FusedStrandStep(ref, rcams, opt, bg)
        params = vp.params(ref)
        for it in range(1, 5):
            ref.update_learning_rate(it)
            acc = [torch.zeros_like(p) for p in params]
            for r in range(world):
                for view in allpicks[r][it - 1]:
                    fused.views.select(view)
                    l, _ = fused.loss()
                    fused.backward(l)
                    for a, p in zip(acc, params):
                        if p.grad is not None:
                            a += p.grad
                        p.grad = None
            for a, p in zip(acc, params):
                p.grad = a / (world * per_rank)
            ref.optimizer.step()
            ref.optimizer.zero_grad(set_to_none=True)
        rflat = torch.cat([p.detach().reshape(-1) for p in params]).cpu()
        d = (rflat - gathered[0]).abs()
        assert float(d.max()) <= 2e-4 * float(rflat.abs().max()), float(d.max())
        print("VP_GPU_OK", float(d.max()))
    if rccl and world > 1:
        # the transport itself: every rank's gradient of ITS view of one fixed draw, averaged by ONE RCCL all-reduce over the
        # flat buffer, against rank 0's own mean of all those views' gradients (<= 1e-5 relative)
        from hgs_runtime.strand_step import FusedStrandStep
        fused = FusedStrandStep(model, cams, opt, bg)
        params = vp.params(model)

        def grad_of(view):
            for p in params:
                p.grad = None
            fused.views.select(view)
            l, _ = fused.loss()
            fused.backward(l)
            return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params]).clone()

        for p in params:
            p.grad = None
        fused.views.select(cams[rank % len(cams)])
        l, _ = fused.loss()
        fused.backward(l)
        vp.reduce_gradients(model)                       # pack + all_reduce(AVG) over RCCL
        got = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params]).clone()
        if rank == 0:
            want = sum(grad_of(cams[r % len(cams)]).double() for r in range(world)) / world
            err = float((got.double() - want).abs().max()) / max(float(want.abs().max()), 1e-30)
            assert err <= 1e-5, err
            print("VP_RCCL_GRAD_OK", err)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
