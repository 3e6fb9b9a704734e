"""Rows per Gaussian that preprocess_bwd_kernel gathers (instances after the tile cull), on the state N iterations of the full loop
leave: distribution, and what a wavefront waits for -- its largest lane -- under the shipped loop (2 rows per trip) and under a
split at K rows with the rest summed by the whole wavefront (4 rows per load, U loads in flight)."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import numpy as np, torch
import hgs_runtime as rt
from arguments import OptimizationParams
from synthetic import build_workload
from train import training
from utils.general import safe_state
from diff_gaussian_rasterization import _C

wl = sys.argv[1] if len(sys.argv) > 1 else "north_star"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
safe_state(True)
model, cams, extent = build_workload(wl, device="cuda", seed=0, with_targets=True)
opt = OptimizationParams()
model.training_setup(opt)
if iters:
    training(model, cams, opt, iterations=iters, extent=extent, start_iteration=10, seed=1, steps_per_graph=8)
torch.cuda.synchronize()
_C.set_async(False)
P = model.get_xyz.shape[0]
lay = rt.layout("geom", P)
for cam in cams[:3]:
    with torch.no_grad():
        R, color, radii, geom, binning, img = _C.rasterize_gaussians_culled(
            torch.zeros(3, device="cuda"), model.get_xyz, torch.empty(0, device="cuda"), model.get_opacity, model.get_scaling,
            model.get_rotation, 1.0, torch.empty(0, device="cuda"), cam.world_view_transform, cam.full_proj_transform,
            math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), cam.image_height, cam.image_width, model.get_features,
            model.active_sh_degree, cam.camera_center, False, False)
    tt = geom.cpu().numpy()[lay["tiles_touched"]:lay["tiles_touched"] + 4 * P].view(np.uint32).astype(np.int64)
    waves = [tt[i:i + 64] for i in range(0, P, 64)]
    wmax = np.array([w.max() for w in waves])
    cur = np.ceil(wmax / 2).sum()
    print(f"{wl} after {iters} iterations: P {P} rows {tt.sum()} mean {tt.mean():.2f} p50/90/99/99.9/max "
          f"{np.percentile(tt, [50, 90, 99, 99.9]).tolist()} {tt.max()}  per-wave max: mean {wmax.mean():.1f} p90 {np.percentile(wmax, 90):.0f}")
    print(f"   dependent trips per wave, shipped (2 rows per trip): {cur / len(waves):.2f}")
    for K in (4, 8, 16):
        for U in (2, 4):
            t = 0.0
            for w in waves:
                t += math.ceil(min(w.max(), K) / 2)
                heavy = w[w > K]
                t += sum(math.ceil((h - K) / (4 * U)) for h in heavy)
            print(f"   split at K = {K}, {U} loads in flight: {t / len(waves):.2f} trips per wave, heavy lanes per wave {sum((w > K).sum() for w in waves) / len(waves):.2f}")
