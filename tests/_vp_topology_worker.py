"""Worker of tests/test_gpu_view_parallel.py::test_two_ranks_through_the_topology_operators: N ranks sharing one GPU (gloo) run
train.training() -- graph replays, densification + merging + opacity reset at their intervals, re-captures -- with one view per
rank and step.  SURVEY.md 8e: the operators must leave identical state on every rank: the statistics they decide on are
reduced over the ranks, their random draws come from identically seeded generators."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import torch
import torch.distributed as dist


def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    from arguments import OptimizationParams
    from diff_gaussian_rasterization import _C as raster
    from synthetic import build_workload
    from train import ViewParallel, training
    from utils.general import safe_state
    safe_state(True)
    model, cams, extent = build_workload("tiny", device="cuda", with_targets=True)
    opt = OptimizationParams()
    opt.densify_from_iter, opt.densification_interval, opt.merge_interval, opt.opacity_reset_interval = 3, 6, 8, 12
    model.training_setup(opt)
    P0 = model.get_xyz.shape[0]
    vp = ViewParallel()
    assert vp.world == world
    ema = training(model, cams, opt, iterations=22, extent=extent, use_graph=True, vp=vp)
    raster.set_async(False)
    assert torch.isfinite(ema)
    P1 = model.get_xyz.shape[0]
    sizes = [None] * world
    dist.all_gather_object(sizes, (P1, int(model._endpoints.shape[0]), int(model.strands_info.n_strands)))
    assert all(s == sizes[0] for s in sizes), f"ranks disagree on the topology: {sizes}"
    assert P1 != P0, "the operators did not change the topology"
    flat = torch.cat([p.detach().reshape(-1) for p in vp.params(model)] + [model.endpoint_pairs.reshape(-1).float()]).cpu()
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    for r in range(1, world):
        assert torch.equal(gathered[0], gathered[r]), f"rank {r} diverged from rank 0"
    moments = torch.cat([model.optimizer.state[g["params"][0]]["exp_avg"].reshape(-1) for g in model.optimizer.param_groups
                         if g["params"][0].numel()]).cpu()
    gm = [torch.empty_like(moments) for _ in range(world)]
    dist.all_gather(gm, moments)
    for r in range(1, world):
        assert torch.equal(gm[0], gm[r]), f"Adam moments of rank {r} diverged"
    if rank == 0:
        print(f"VP_TOPOLOGY_OK segments {P0} -> {P1}, {world} ranks")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
