"""Status words of one forward pass of a workload (segment policy diagnostics): tools/seg_status.py [workload]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd"), os.path.join(ROOT, "tests")]
import math, numpy as np, torch
import hgs_runtime as rt
from diff_gaussian_rasterization import _C
from synthetic import build_workload
wl = sys.argv[1] if len(sys.argv) > 1 else "north_star"
model, cams, _ = build_workload(wl, device="cuda", with_targets=False, n_views=2)
cam = cams[0]
H, W = cam.image_height, cam.image_width
if os.environ.get("HGS_SEG_POLICY"):
    pol = [int(x) for x in os.environ["HGS_SEG_POLICY"].split(",")]
    rt.check(rt.lib().hgs_set_segment_policy(*pol[:3]))
with torch.no_grad():
    out = _C.rasterize_gaussians(torch.zeros(3, device="cuda"), model.get_xyz, torch.empty(0, device="cuda"), model.get_opacity,
                                 model.get_scaling, model.get_rotation, 1.0, torch.empty(0, device="cuda"), cam.world_view_transform,
                                 cam.full_proj_transform, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), H, W,
                                 model.get_features, model.active_sh_degree, cam.camera_center, False, False)
torch.cuda.synchronize()
lay = rt.layout("image", W, H)
st = out[5][lay["status"]:lay["status"] + 64].cpu().numpy().view(np.uint32)
names = ["R", "overflow", "scan_lo", "scan_hi", "sort_items", "split_items", "timeout", "seg_len", "unsplit"]
print(wl, {n: int(v) for n, v in zip(names, st)})
