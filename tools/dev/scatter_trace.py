"""Phase times of the scatter kernel's workgroups on the state N iterations of the full loop leave (or the initial state, N = 0).
Needs a library built with -DHGS_SCATTER_TRACE=1 (hgs_preprocess.hip), selected through HGS_LIB:
  HGS_LIB=$PWD/hair-gs_amd/libhgs_sctrace.so python tools/dev/scatter_trace.py [iterations=1000] [workload=north_star]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import numpy as np, torch
import hgs_runtime as rt
from arguments import OptimizationParams
from synthetic import PIPELINE_STATES, build_pipeline_state, build_workload
from train import training, training_step, ViewSampler
from diff_gaussian_rasterization import _C
from utils.general import safe_state
safe_state(True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
wl = sys.argv[2] if len(sys.argv) > 2 else "north_star"
if wl in PIPELINE_STATES:      # (a state of the three-stage workflow: `iterations` is ignored, the state's own Stage-I loop runs)
    model, cams, extent, _info = build_pipeline_state(wl, device="cuda", n_views=8)
    n = 0
    opt = OptimizationParams()
else:
    model, cams, extent = build_workload(wl, device=torch.device("cuda"), seed=0, n_views=8)
    opt = OptimizationParams()
    model.training_setup(opt)
bg = torch.zeros(3, device="cuda")
if n:
    training(model, cams, opt, iterations=n, extent=extent, seed=1)
opt.enable_topology = False
from hgs_runtime.strand_step import ViewTable, fused_step_for
fused = fused_step_for(model, ViewTable(cams), opt, bg)
fused.defer_tail = True
sampler = ViewSampler(cams, seed=0)
_C.set_async(True)
L = rt.lib()
L.hgs_debug_scatter_trace.argtypes = [C.c_void_p, C.c_int]
P = model.get_xyz.shape[0]
nwg = (P + 255) // 256 + 4
acc, ev = [], []
for i in range(8):
    rt.prof_collect(); rt.prof_enable(True)
    training_step(model, sampler.next(), opt, bg, n + 1 + i, extent=extent, fused=fused)
    torch.cuda.synchronize()
    k = rt.prof_collect(); rt.prof_enable(False)
    buf = np.zeros((min(nwg, 8192), 8), dtype=np.uint64)
    assert L.hgs_debug_scatter_trace(buf.ctypes.data, buf.shape[0]) == 0
    if i >= 3:
        acc.append(buf.astype(np.int64))
        ev.append({a: round(v[0] / v[1] * 1e3, 1) for a, v in k.items() if v[1] and a in ("scatter_kernel", "preprocess_fwd_kernel", "sort_tiles_kernel")})
_C.set_async(False)
names = ["loads + block prefix", "count tiles (LDS)", "reserve (global atomics)", "wait for the scan", "place keys"]
print(f"P {P} workgroups {nwg} (first 4: scan)")
for t, e in zip(acc[-3:], ev[-3:]):
    t0 = t[:, 0].min()
    print("  HIP events around the launches of this step (us):", e, " trace span:", round((t[:, 5].max() - t0) * 0.01, 1))
    g = t[4:]
    us = lambda a: a * 0.01
    print("  scan workgroups: start", us(t[:4, 0] - t0).round(1).tolist(), "end", us(t[:4, 5] - t0).round(1).tolist())
    print("  gaussian workgroups: start p50/p90/max", np.percentile(us(g[:, 0] - t0), [50, 90, 100]).round(1).tolist(),
          "end p50/p90/max", np.percentile(us(g[:, 5] - t0), [50, 90, 100]).round(1).tolist())
    for k, nm in enumerate(names):
        d = us(g[:, k + 1] - g[:, k])
        print(f"    {nm:28s} mean {d.mean():6.2f} p90 {np.percentile(d, 90):6.2f} max {d.max():6.2f} us")
