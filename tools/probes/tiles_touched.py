import os, sys
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch, math
import hgs_runtime as rt
from diff_gaussian_rasterization import _C
from synthetic import build_workload
for wl in ("c2", "north_star"):
    model, cams, _ = build_workload(wl, device="cuda", with_targets=False, n_views=4)
    cam = cams[0]; H, W = cam.image_height, cam.image_width
    with torch.no_grad():
        tfx, tfy = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
        R, color, radii, geom, binning, img = _C.rasterize_gaussians_culled(
            torch.zeros(3, device="cuda"), model.get_xyz, torch.empty(0, device="cuda"), model.get_opacity, model.get_scaling,
            model.get_rotation, 1.0, torch.empty(0, device="cuda"), cam.world_view_transform, cam.full_proj_transform, tfx, tfy,
            H, W, model.get_features, model.active_sh_degree, cam.camera_center, False, False)
    P = model.get_xyz.shape[0]
    lay = rt.layout("geom", P)
    tt = geom.cpu().numpy()[lay["tiles_touched"]:lay["tiles_touched"] + 4 * P].view(np.uint32)
    print(wl, "P", P, "R", R, "tiles touched: mean %.2f max %d" % (tt.mean(), tt.max()), "percentiles 50/90/99/99.9:", np.percentile(tt, [50, 90, 99, 99.9]).tolist(),
          "per-wave max mean:", float(np.mean([tt[i:i+64].max() for i in range(0, P, 64)])), "sum of per-wave max*64 / sum:", float(sum(tt[i:i+64].max()*64 for i in range(0,P,64)) / max(1,tt.sum())))
