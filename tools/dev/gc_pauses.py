"""How long the Python garbage collector stops the training loop: gc.callbacks time every collection (by generation) during a soak
run (argv: workload | pipeline state, iterations); prints the number of tracked objects too."""
import gc, os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import torch
import train
from arguments import OptimizationParams
from synthetic import build_workload, build_pipeline_state, PIPELINE_STATES
from utils.general import safe_state
wl = sys.argv[1] if len(sys.argv) > 1 else "north_star"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
pause, count, t_start = collections.Counter(), collections.Counter(), {}


def cb(phase, info):
    if phase == "start":
        t_start[0] = time.perf_counter()
    else:
        pause[info["generation"]] += time.perf_counter() - t_start[0]; count[info["generation"]] += 1


gc.callbacks.append(cb)
for rep in range(2):
    safe_state(True)
    opt = OptimizationParams()
    if wl in PIPELINE_STATES:
        model, cams, extent, _ = build_pipeline_state(wl, device="cuda", seed=0)
        opt.iterations = 5000; opt._finalise()
    else:
        model, cams, extent = build_workload(wl, device="cuda", seed=0, n_views=16)
    model.training_setup(opt)
    pause.clear(); count.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    train.training(model, cams, opt, iterations=iters, extent=extent)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"run {rep}: {iters} iterations in {dt:.3f} s; collector pauses by generation: "
          + ", ".join(f"gen{g}: {count[g]} x, {1e3 * pause[g]:.1f} ms" for g in sorted(count)) + f"; tracked objects {len(gc.get_objects())}", flush=True)
    del model, cams
