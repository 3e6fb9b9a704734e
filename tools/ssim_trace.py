"""Phase times inside the SSIM kernels of one training iteration (development aid; needs a library built with
-DHGS_SSIM_TRACE=1: tools/build_variant.sh ssimtrace hgs_losses -DHGS_SSIM_TRACE=1; HGS_LIB=... python tools/ssim_trace.py)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import numpy as np, torch
import hgs_runtime as rt
from arguments import OptimizationParams
from synthetic import build_workload
from train import training_step
from hgs_runtime.strand_step import ViewTable, fused_step_for
wl = sys.argv[1] if len(sys.argv) > 1 else "north_star"
model, cams, extent = build_workload(wl, device="cuda", seed=0, n_views=4)
opt = OptimizationParams(); opt.enable_topology = False
model.training_setup(opt)
bg = torch.zeros(3, device="cuda")
views = ViewTable(cams)
fused = fused_step_for(model, views, opt, bg)
for it in range(1, 6):
    training_step(model, cams[it % len(cams)], opt, bg, it, extent=extent, fused=fused)
torch.cuda.synchronize()
NWG = 2048
tf = torch.zeros(16 * NWG, dtype=torch.int64, device="cuda"); tb = torch.zeros(16 * NWG, dtype=torch.int64, device="cuda")
fn = rt.lib().hgs_debug_set_ssim_trace
fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p]; fn.restype = ctypes.c_int
assert fn(tf.data_ptr(), tb.data_ptr()) == 0
training_step(model, cams[2], opt, bg, 6, extent=extent, fused=fused)
torch.cuda.synchronize()
assert fn(None, None) == 0
names = ["tile->LDS+barrier", "row pass+barrier", "col pass", "epilogue", "block sums+barrier", "-", "-", "prefetch issue/loop"]
for name, t in (("ssim_l1_fwd", tf), ("ssim_l1_bwd", tb)):
    a = t.cpu().numpy().reshape(NWG, 16)
    a = a[a[:, 11] > 0]
    span = (a[:, 11].max() - a[:, 10].min()) * 0.01
    life = (a[:, 11] - a[:, 10]) * 0.01
    print(f"{name}: {len(a)} workgroups, span {span:.1f} us, WG lifetime mean {life.mean():.1f} max {life.max():.1f} us, blocks {a[:, 0].sum()} (filtered {a[:, 1].sum()}), "
          f"blocks per WG mean {a[:, 0].mean():.1f} max {a[:, 0].max()}")
    ph = a[:, 2:10] * 0.01
    for k in range(8):
        if ph[:, k].sum() > 0:
            print(f"   {names[k]:22s} mean per WG {ph[:, k].mean():6.2f} us   per filtered block {ph[:, k].sum() / max(1, a[:, 1].sum()):5.2f} us")
    print("   start offsets (us) p50/p99:", np.round(np.percentile((a[:, 10] - a[:, 10].min()) * 0.01, [50, 99]), 1).tolist(),
          " end offsets p10/p50/p99:", np.round(np.percentile((a[:, 11] - a[:, 10].min()) * 0.01, [10, 50, 99]), 1).tolist())
