"""Where the wall clock of training() goes between the graph replays (cProfile over a soak run with the operators)."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import torch
from arguments import OptimizationParams
from synthetic import build_workload
from train import training
from utils.general import safe_state
wl = sys.argv[1] if len(sys.argv) > 1 else "north_star"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
safe_state(True)
model, cams, extent = build_workload(wl, device="cuda", seed=0, n_views=16)
opt = OptimizationParams()
model.training_setup(opt)
training(model, cams, opt, iterations=100, extent=extent)          # warm-up (lazy loads)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
training(model, cams, opt, iterations=iters, extent=extent, start_iteration=100)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.strip_dirs(); st.sort_stats(os.environ.get("SORT", "cumulative")).print_stats(int(os.environ.get("TOP", "45")))
for fn in os.environ.get("CALLEES", "").split(","):
    if fn:
        st.print_callees(fn)
