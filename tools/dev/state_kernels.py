"""Per-kernel times (HIP events around every launch, eager fused iterations) on the state N iterations of the full loop leave,
as it is and after an explicit sort_spatially(); and the block-level locality of that state's tile rectangles.
  python tools/dev/state_kernels.py [iterations=3000] [workload=north_star]"""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import numpy as np, torch
import hgs_runtime as rt
from arguments import OptimizationParams
from synthetic import build_workload
from train import training, training_step, ViewSampler, fused_step_applicable
from diff_gaussian_rasterization import _C
from utils.general import safe_state
safe_state(True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
wl = sys.argv[2] if len(sys.argv) > 2 else "north_star"
model, cams, extent = build_workload(wl, device=torch.device("cuda"), seed=0, n_views=8)
opt = OptimizationParams()
model.training_setup(opt)
bg = torch.zeros(3, device="cuda")
training(model, cams, opt, iterations=n, extent=extent, seed=1)
opt.enable_topology = False


def kernels(tag):
    from hgs_runtime.strand_step import ViewTable, fused_step_for
    views = ViewTable(cams)
    fused = fused_step_for(model, views, opt, bg)
    fused.defer_tail = True
    sampler = ViewSampler(cams, seed=0)
    _C.set_async(True)
    for i in range(3):
        training_step(model, sampler.next(), opt, bg, n + 1 + i, extent=extent, fused=fused)
    torch.cuda.synchronize()
    rt.prof_collect(); rt.prof_enable(True)
    for i in range(20):
        training_step(model, sampler.next(), opt, bg, n + 4 + i, extent=extent, fused=fused)
    torch.cuda.synchronize()
    k = rt.prof_collect(); rt.prof_enable(False)
    _C.set_async(False)
    print(tag, "P", model.get_xyz.shape[0], {a: round(v[0] / v[1] * 1e3, 1) for a, v in k.items() if v[1]})


def locality(tag):
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
    rect = geom[lay["rect"]:lay["rect"] + 16 * P].view(torch.int16).reshape(P, 8).cpu().numpy().astype(np.int64)
    x0, y0, x1, y1 = rect[:, 0], rect[:, 1], rect[:, 2], rect[:, 3]
    gx = (c.image_width + 15) // 16
    distinct, inst = [], []
    for b in range(0, P, 256):
        s = set()
        for i in range(b, min(P, b + 256)):
            if tt[i]:
                for ty in range(y0[i], y1[i]):
                    for tx in range(x0[i], x1[i]):
                        s.add(ty * gx + tx)
        distinct.append(len(s)); inst.append(int(tt[b:b + 256].sum()))
    distinct, inst = np.array(distinct), np.array(inst)
    print(tag, "R", out[0], "visible", int((tt > 0).sum()), "blocks", len(distinct), "distinct tiles per block p50/p90/max",
          np.percentile(distinct, [50, 90]).tolist(), distinct.max(), "instances per block p50/p90/max", np.percentile(inst, [50, 90]).tolist(), inst.max())


locality("as trained")
kernels("as trained")
model.sort_spatially()
locality("after sort_spatially()")
kernels("after sort_spatially()")
