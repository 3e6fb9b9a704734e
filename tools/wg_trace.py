"""Workgroup timeline of the blend kernels on a workload (development aid, hgs_debug_set_wg_trace): residency over time,
per-tile duration against list length."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import numpy as np, torch
import hgs_runtime as rt
from gaussian_renderer import render_multi
from synthetic import build_workload
wl = sys.argv[1] if len(sys.argv) > 1 else "north_star"
model, cams, _ = build_workload(wl, device="cuda", with_targets=False, n_views=4)
bg = torch.zeros(3, device="cuda")
cam = cams[0]
H, W = cam.image_height, cam.image_width
T = ((W + 15) // 16) * ((H + 15) // 16)
w3 = torch.randn(3, H, W, device="cuda"); w1 = torch.randn(H, W, device="cuda"); wo = torch.randn(3, H, W, device="cuda")
tf = torch.zeros(2 * T, dtype=torch.int64, device="cuda"); tb = torch.zeros(2 * T, dtype=torch.int64, device="cuda")
def run():
    extra = torch.cat((model.get_mask, model.get_orientation), dim=1)
    pkg = render_multi(cam, model, bg, extra, splits=(1, 3))
    loss = (pkg["render"] * w3).sum() + (pkg["extra"][0] * w1).sum() + (pkg["extra"][1] * wo).sum()
    loss.backward()
    model._derived = None
for _ in range(3): run()
torch.cuda.synchronize()
rt.check(rt.lib().hgs_debug_set_wg_trace(tf.data_ptr(), tb.data_ptr()))
run(); torch.cuda.synchronize()
rt.check(rt.lib().hgs_debug_set_wg_trace(None, None))
from diff_gaussian_rasterization import _C
for name, t in (("fwd", tf), ("bwd", tb)):
    a = t.cpu().numpy().reshape(T, 2).astype(np.float64) * 0.01   # us (100 MHz)
    ok = a[:, 1] > 0
    t0, t1 = a[ok, 0].min(), a[ok, 1].max()
    dur = a[:, 1] - a[:, 0]
    print(f"{name}: kernel span {t1 - t0:.1f} us, tiles {ok.sum()}, mean WG dur {dur[ok].mean():.2f} us, max {dur[ok].max():.2f}, sum {dur[ok].sum():.0f} us")
    # residency over time
    edges = np.linspace(t0, t1, 24)
    occ = [(((a[ok, 0] <= e) & (a[ok, 1] > e)).sum()) for e in edges]
    print("   resident WGs over time:", occ)
    order = np.argsort(a[ok, 0]); first = a[ok, 0][order]
    print("   WG start times (us from kernel start), every 800th:", np.round(first[::800] - t0, 1).tolist())
    long = np.argsort(-dur)[:8]
    print("   longest tiles:", [(int(i), round(float(dur[i]), 1), round(float(a[i, 0] - t0), 1)) for i in long])
    print("   duration percentiles 10/50/90/99:", np.round(np.percentile(dur[ok], [10, 50, 90, 99]), 2).tolist())
