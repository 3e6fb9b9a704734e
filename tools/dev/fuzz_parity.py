"""Randomised runs of the parity tests (development aid, on the GPU box): the raster tests on random scene parameters (sizes that
are no multiples of anything, 0 / 1 / few Gaussians, every SH degree, backgrounds, opacity and scale ranges) and the training-step
equality tests on random tiny strand workloads.  Prints the parameters of every failure.   python tools/dev/fuzz_parity.py [n] [seed]"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import numpy as np
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
from tests import test_gpu_raster as R
from tests import test_gpu_train as T
import synthetic

fails = 0


def attempt(label, fn):
    global fails
    try:
        fn()
        return True
    except Exception:
        fails += 1
        print("FAIL", label)
        traceback.print_exc(limit=6)
        return False


for k in range(n):
    P = int(rng.choice([0, 1, 2, 7, 63, 64, 65, 300, 1500, 4000]))
    W, H = int(rng.integers(1, 300)), int(rng.integers(1, 200))
    lo = float(rng.choice([0.002, 0.01, 0.03]))
    v = dict(P=P, W=W, H=H, seed=int(rng.integers(0, 1 << 30)), sh_degree=int(rng.integers(0, 4)),
             bg=tuple(float(x) for x in rng.choice([0.0, 0.0, 0.3, 1.0], 3)), scale_lo=lo, scale_hi=lo * float(rng.choice([2, 5, 20])),
             opacity_lo=float(rng.choice([0.003, 0.05, 0.9])), opacity_hi=0.99, behind_frac=float(rng.choice([0.0, 0.05, 0.5])),
             fovx_deg=float(rng.choice([30.0, 60.0, 100.0])), depth_levels=int(rng.choice([0, 0, 4])),
             use_colors_precomp=bool(rng.random() < 0.2), use_cov3D_precomp=bool(rng.random() < 0.2),
             scale_modifier=float(rng.choice([1.0, 1.0, 0.5])))
    if v["use_colors_precomp"]:
        v["sh_degree"] = 0
    R.VARIANTS["fuzz"] = v
    print(f"[raster {k}] {v}", flush=True)
    if P == 0:
        continue
    ok = attempt(f"forward {v}", lambda: R.test_forward_matches_oracle("fuzz"))
    if ok and P > 0:
        attempt(f"backward {v}", lambda: R.test_backward_matches_oracle("fuzz"))
        attempt(f"tile cull {v}", lambda: R.test_tile_cull_changes_no_output("fuzz"))
        attempt(f"capacity mode {v}", lambda: R.test_capacity_mode_binning_equals_blocking_mode("fuzz"))

tiny0 = synthetic.WORKLOADS["tiny"]
for k in range(max(4, n // 4)):
    kw = dict(n_strands=int(rng.choice([1, 2, 9, 40, 130])), n_seg=int(rng.choice([1, 2, 5, 30, 64])))
    W, H = int(rng.choice([16, 33, 256, 321])), int(rng.choice([16, 47, 144, 201]))
    synthetic.WORKLOADS["tiny"] = ("strands", kw, 4, W, H)
    print(f"[train {k}] {kw} {W}x{H}", flush=True)
    for fn in (T.test_fused_parameter_backward_equals_two_launches, T.test_one_launch_parameters_and_preprocess_equals_two_launches,
               lambda: T.test_adam_in_the_backward_lanes_equals_the_adam_launch("strands"),
               T.test_graphed_step_matches_eager_steps, T.test_deferred_head_tail_gives_the_same_terms_and_gradients,
               T.test_iteration_prologue_equals_select_then_forward):
        # (not here: the tests whose assertions assume the size of the stock tiny scene -- more than 100 visible segments, a DSSIM
        # term far enough from 0 for a relative tolerance)
        attempt(f"{getattr(fn, '__name__', 'adam/op-by-op')} {kw} {W}x{H}", fn)
synthetic.WORKLOADS["tiny"] = tiny0
print("failures:", fails)
