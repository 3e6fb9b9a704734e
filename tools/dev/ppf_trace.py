"""Phase times of the preprocess kernel's workgroups on a workload / pipeline state (dev aid).
Needs a library built with -DHGS_PPF_TRACE=1 (hgs_preprocess.hip), selected through HGS_LIB:
  tools/build_variant.sh ppftrace hgs_preprocess -DHGS_PPF_TRACE=1
  HGS_LIB=$PWD/hair-gs_amd/libhgs_ppftrace.so python tools/dev/ppf_trace.py [workload=stage1_1080p]"""
import ctypes as C, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import numpy as np, torch
import hgs_runtime as rt
from synthetic import PIPELINE_STATES, build_pipeline_state, build_workload
from diff_gaussian_rasterization import _C
wl = sys.argv[1] if len(sys.argv) > 1 else "stage1_1080p"
if wl in PIPELINE_STATES:
    model, cams, extent, _ = build_pipeline_state(wl, device="cuda", n_views=4)
else:
    model, cams, extent = build_workload(wl, device="cuda", with_targets=False, n_views=4)
L = rt.lib()
L.hgs_debug_ppf_trace.argtypes = [C.c_void_p, C.c_int]
bg = torch.zeros(3, device="cuda")
_C.set_async(False)
P = model.get_xyz.shape[0]
nwg = (P + 255) // 256
for i in range(4):
    c = cams[i % len(cams)]
    was = _C.set_tile_cull(True)
    with torch.no_grad():
        _C.rasterize_gaussians(bg, model.get_xyz, torch.empty(0, device="cuda"), model.get_opacity, model.get_scaling, model.get_rotation,
                               1.0, torch.empty(0, device="cuda"), c.world_view_transform, c.full_proj_transform, math.tan(c.FoVx * 0.5),
                               math.tan(c.FoVy * 0.5), c.image_height, c.image_width, model.get_features, model.active_sh_degree,
                               c.camera_center, False, False)
    _C.set_tile_cull(was)
    torch.cuda.synchronize()
    buf = np.zeros((min(nwg, 8192), 8), dtype=np.uint64)
    assert L.hgs_debug_ppf_trace(buf.ctypes.data, buf.shape[0]) == 0
t = buf.astype(np.int64)
t0 = t[:, 0].min()
us = lambda a: a * 0.01
print(f"P {P} workgroups {nwg}; kernel span {us(t[:, 4].max() - t0):.1f} us")
print("  start p50/p90/max", np.percentile(us(t[:, 0] - t0), [50, 90, 100]).round(1).tolist(), " end p50/p90/max", np.percentile(us(t[:, 4] - t0), [50, 90, 100]).round(1).tolist())
for k, nm in enumerate(["loads + rectangles + own counting", "block sum", "dealt counting of large rectangles", "table flush"]):
    d = us(t[:, k + 1] - t[:, k])
    print(f"    {nm:36s} mean {d.mean():6.2f} p90 {np.percentile(d, 90):6.2f} max {d.max():6.2f} us")
bs = t[:, 5]
order = np.argsort(-(t[:, 4] - t[:, 0]))[:6]
print("  slowest workgroups (instances, us per phase):", [(int(bs[i]), us(t[i, 1:5] - t[i, 0:4]).round(1).tolist()) for i in order])
print("  correlation of a workgroup's duration with its instance count:", round(float(np.corrcoef(bs, t[:, 4] - t[:, 0])[0, 1]), 3))
