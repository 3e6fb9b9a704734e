"""N eager (ungraphed) fused training iterations of a workload -- a target for rocprofv3 --pmc passes (tools/pmc_generic.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import torch
from arguments import OptimizationParams
from synthetic import build_workload
from diff_gaussian_rasterization import _C as raster
from train import training_step
from hgs_runtime.strand_step import ViewTable, fused_step_for
wl = sys.argv[1] if len(sys.argv) > 1 else "north_star"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
model, cams, extent = build_workload(wl, device="cuda", seed=0, n_views=4)
opt = OptimizationParams(); opt.enable_topology = False
model.training_setup(opt)
bg = torch.zeros(3, device="cuda")
fused = fused_step_for(model, ViewTable(cams), opt, bg)
fused.defer_tail = True   # (as the captured iteration: no launch for the loss head's tail)
fused.enable_inline_adam(True)   # (as the captured single-rank iteration: Adam in the backward's lanes, no optimizer launch)
raster.set_async(True)    # capacity mode, as the captured iteration runs: the first launch is hair_preprocess_fwd_kernel
for it in range(1, n + 1):
    training_step(model, cams[it % len(cams)], opt, bg, it, extent=extent, fused=fused)
torch.cuda.synchronize()
raster.check_async()
print("done")
