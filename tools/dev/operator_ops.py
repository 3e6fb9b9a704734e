"""torch.profiler over ONE densification + merging event on a trained strand model: which torch operators (and how many launches)
the topology operators spend their host and device time in."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import torch
from torch.profiler import profile, ProfilerActivity
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
print("segments", model.get_xyz.shape[0])
for name, fn in (("densification", lambda: model.densification(extent, None, None)), ("merging", lambda: model.merging())):
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    ka = prof.key_averages()
    tot_cpu = sum(e.self_cpu_time_total for e in ka) / 1e3
    tot_dev = sum(getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0)) for e in ka) / 1e3
    print(f"== {name}: self CPU {tot_cpu:.1f} ms, device {tot_dev:.1f} ms, {sum(e.count for e in ka)} calls")
    print(ka.table(sort_by="self_cpu_time_total", row_limit=22, max_name_column_width=48))
