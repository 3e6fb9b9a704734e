"""Tiles per Gaussian (after culling) on the state bench.py --trained-iters leaves: how many Gaussians take the scatter
kernel's slow path (more than TH_MAX_AREA = 16 tiles), and how long the longest lane's loop is."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import numpy as np, torch
import hgs_runtime as rt
from arguments import OptimizationParams
from synthetic import build_workload
from train import training
from diff_gaussian_rasterization import _C
from utils.general import safe_state
safe_state(True)
model, cams, extent = build_workload("north_star", device=torch.device("cuda"), seed=0, n_views=8)
opt = OptimizationParams()
model.training_setup(opt)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
for phase in ("initial", "trained"):
    if phase == "trained":
        training(model, cams, opt, iterations=n, extent=extent, seed=1)
    bg = torch.zeros(3, device="cuda")
    c = cams[1]
    was = _C.set_tile_cull(True)
    with torch.no_grad():
        out = _C.rasterize_gaussians(bg, model.get_xyz, torch.empty(0, device="cuda"), model.get_opacity, model.get_scaling,
                                     model.get_rotation, 1.0, torch.empty(0, device="cuda"), c.world_view_transform,
                                     c.full_proj_transform, math.tan(c.FoVx * 0.5), math.tan(c.FoVy * 0.5), c.image_height,
                                     c.image_width, model.get_features, model.active_sh_degree, c.camera_center, False, False)
    _C.set_tile_cull(was)
    P = model.get_xyz.shape[0]
    lay = rt.layout("geom", P)
    geom = out[3]
    tt = geom[lay["tiles_touched"]:lay["tiles_touched"] + 4 * P].view(torch.int32).cpu().numpy()
    vis = tt[tt > 0]
    print(phase, "P", P, "R", out[0], "visible", vis.size, "mean tiles", vis.mean().round(2), "p50/p90/p99/max", np.percentile(vis, [50, 90, 99]).tolist(), vis.max(),
          "| > 16 tiles:", int((tt > 16).sum()), "instances on the slow path:", int(tt[tt > 16].sum()), f"({tt[tt > 16].sum() / max(1, tt.sum()):.1%})")
    # per 256-Gaussian block: the longest lane's tile count, distinct tiles are not known from here
    blk = np.pad(tt, (0, (-P) % 256)).reshape(-1, 256)
    print("   per workgroup: max lane tiles p50/p90/max", np.percentile(blk.max(1), [50, 90]).tolist(), blk.max(), " sum per WG p50/p90/max", np.percentile(blk.sum(1), [50, 90]).tolist(), blk.sum(1).max())
