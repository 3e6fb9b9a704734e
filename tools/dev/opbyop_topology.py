"""Repro driver: op-by-op iteration + graph replay + topology operators (re-captures)."""
import faulthandler, os, sys
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import torch
from arguments import OptimizationParams
from synthetic import build_workload
from train import training
from utils.general import safe_state
safe_state(True)
wl = sys.argv[1] if len(sys.argv) > 1 else "tiny"
model, cams, extent = build_workload(wl, device=torch.device("cuda"), seed=0, n_views=4, consistent=True)
opt = OptimizationParams()
opt.fused_step = False
if wl == "tiny":
    opt.densify_from_iter, opt.densification_interval, opt.merge_interval, opt.opacity_reset_interval = 3, 6, 8, 12
model.training_setup(opt)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for chunk in range(3):
    ema = training(model, cams, opt, iterations=n, extent=extent, start_iteration=chunk * n)
    torch.cuda.synchronize()
    print("chunk", chunk, float(ema), model.get_xyz.shape[0], flush=True)
