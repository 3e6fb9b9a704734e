"""Worker of tests/test_gpu_view_parallel.py::test_two_ranks_roll_a_capacity_overflow_back_together: train.training() on two ranks
(gloo, one GPU) with a binning capacity that is too small -- the ranks must decide on the maximum over both, return to the same
checkpoint and end like a run that never overflowed, bit for bit, on every rank."""
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
    from synthetic import attach_targets, cameras_extent, make_cameras, make_strand_model
    import train as T
    from utils.general import safe_state
    cams = make_cameras(4, 400, 240, device="cuda")
    extent = cameras_extent(cams)

    def run(slack, iterations):
        safe_state(True)
        raster._state["cap"] = 0
        model = make_strand_model(300, 40, device="cuda", spatial_lr_scale=extent)
        model.compute_strands_info(only_foreground=True)
        attach_targets(cams, model)
        opt = OptimizationParams()
        opt.enable_topology = False
        opt.capacity_slack = slack
        model.training_setup(opt)
        vp = T.ViewParallel()
        ema = T.training(model, cams, opt, iterations=iterations, extent=extent, vp=vp)
        state = [g["params"][0].detach().clone() for g in model.optimizer.param_groups]
        for g in model.optimizer.param_groups:
            st = model.optimizer.state[g["params"][0]]
            state += [st["exp_avg"].clone(), st["exp_avg_sq"].clone()]
        return state + [ema.clone()], T.training.last_rollbacks

    small, rb_small = run(0.3, 100)
    roomy, rb_roomy = run(4.0, 100)
    raster.set_async(False)
    assert rb_small >= 1 and rb_roomy == 0, (rb_small, rb_roomy)
    for a, b in zip(small, roomy):
        assert torch.equal(a, b), "the rolled-back run differs from the run that never overflowed"
    flat = torch.cat([t.reshape(-1).float() for t in small[:-1]]).cpu()     # (the loss average is each rank's own views')
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    for r in range(1, world):
        assert torch.equal(gathered[0], gathered[r]), f"rank {r} diverged from rank 0"
    rbs = [None] * world
    dist.all_gather_object(rbs, rb_small)
    assert all(x == rbs[0] for x in rbs), rbs
    if rank == 0:
        print(f"VP_ROLLBACK_OK rollbacks {rb_small}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
