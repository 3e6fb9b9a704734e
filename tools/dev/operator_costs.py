"""True cost of the topology operators on a trained strand model (synchronised timers around each call)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import torch
from arguments import OptimizationParams
from synthetic import build_workload
from train import training
from utils.general import safe_state
safe_state(True)
model, cams, extent = build_workload("north_star", device="cuda", seed=0, n_views=16)
opt = OptimizationParams()
model.training_setup(opt)
training(model, cams, opt, iterations=int(sys.argv[1]) if len(sys.argv) > 1 else 1450, extent=extent)
torch.cuda.synchronize()
print("segments", model.get_xyz.shape[0], "strands", model.strands_info.n_strands)


def timed(name, fn, n=3):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{name:38s} " + " ".join(f"{t:7.1f}" for t in ts) + " ms")


timed("compute_strands_info", lambda: model.compute_strands_info())
timed("storage_order", lambda: model.storage_order())
timed("smoothness_index_pairs (rebuild)", lambda: (setattr(model, "_smooth_pairs", None), model.smoothness_index_pairs()))
timed("compute_endpoint_pair_to_merge", lambda: model.compute_endpoint_pair_to_merge())
timed("foreground mask + unique", lambda: torch.unique(model.endpoint_pairs, return_counts=True))
# one full event, stage by stage (each changes the model: once)
grads = model.xyz_gradient_accum / model.denom
grads[grads.isnan()] = 0.0
timed("clone_strategy", lambda: model.clone_strategy(grads, extent, {}), n=1)
timed("split_strategy", lambda: model.split_strategy(grads, extent, {}), n=1)
timed("merge_collapsed_segments", lambda: model.merge_collapsed_segments({}), n=1)
timed("prune_strategy", lambda: model.prune_strategy(extent, None, {}, avoid_connected=True), n=1)
timed("compute_strands_info (after)", lambda: model.compute_strands_info(), n=1)
timed("sort_spatially", lambda: model.sort_spatially(), n=1)
timed("merging", lambda: model.merging(), n=1)

# ---- re-capture cost on this model (no topology change in between: the allocator holds blocks of the right sizes)
from train import GraphedStep
from hgs_runtime.strand_step import ViewTable
from diff_gaussian_rasterization import _C as raster
bg = torch.zeros(3, device="cuda")
views = ViewTable(cams)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    gs = GraphedStep(model, cams, opt, bg, extent=extent, views=views, steps_per_graph=1)
    t1 = time.perf_counter()
    gs.capture([cams[0], cams[4], cams[8], cams[12]], iteration=2000)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"GraphedStep() {1e3 * (t1 - t0):6.1f} ms   capture (4 warm-up views + 1 captured iteration) {1e3 * (t2 - t1):6.1f} ms")
    gs = None
raster.set_async(False)
