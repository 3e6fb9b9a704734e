"""One densification + merging event of the training loop under torch.profiler: host time per aten operator and runtime call
(hipMalloc / hipFree / synchronisations show here), device time per kernel -- where an event's ~25 ms go on a trained model."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import torch
from torch.profiler import profile, ProfilerActivity
from arguments import OptimizationParams
from synthetic import build_workload
from train import training
from utils.general import safe_state
wl = sys.argv[1] if len(sys.argv) > 1 else "north_star"
start = int(sys.argv[2]) if len(sys.argv) > 2 else 2450
safe_state(True)
model, cams, extent = build_workload(wl, device="cuda", seed=0, n_views=16)
opt = OptimizationParams()
model.training_setup(opt)
training(model, cams, opt, iterations=start, extent=extent)
torch.cuda.synchronize()
t0 = time.perf_counter()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    training(model, cams, opt, iterations=100, extent=extent, start_iteration=start)     # holds exactly one event
    torch.cuda.synchronize()
print(f"100 iterations with one event: {1e3 * (time.perf_counter() - t0):.1f} ms; segments {model.get_xyz.shape[0]}")
ka = prof.key_averages()
print(ka.table(sort_by="self_cpu_time_total", row_limit=40, max_name_column_width=60))
print(ka.table(sort_by="self_cuda_time_total", row_limit=25, max_name_column_width=60))
