import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import torch
import hgs_runtime as rt
from arguments import OptimizationParams
from synthetic import build_workload
import train as T
from utils.general import safe_state
rt.lib(); safe_state(True)
model, cams, extent = build_workload("north_star", device=torch.device("cuda"), seed=0, n_views=16)
opt = OptimizationParams(); model.training_setup(opt)
T.training(model, cams, opt, iterations=599, extent=extent, start_iteration=0)
torch.cuda.synchronize()
# time the pieces of one full cycle: 100 iterations ending in a densification
orig = T.apply_topology
acc = {}
def timed(g, o, it, ext, due, vp=None):
    torch.cuda.synchronize(); t = time.perf_counter()
    orig(g, o, it, ext, due, vp)
    torch.cuda.synchronize(); acc.setdefault(tuple(due), []).append(time.perf_counter() - t)
T.apply_topology = timed
pr = cProfile.Profile()
torch.cuda.synchronize(); t0 = time.perf_counter()
pr.enable()
T.training(model, cams, opt, iterations=401, extent=extent, start_iteration=599)
torch.cuda.synchronize()
pr.disable()
print("401 iterations:", time.perf_counter() - t0, "s; topology events:", {k: [round(x, 3) for x in v] for k, v in acc.items()}, "segments", model.get_xyz.shape[0])
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
