"""Does tools/raster_only.py's train=N state match bench.py's trained_state leg?  Prints segments and instances per view of both recipes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import torch
from arguments import OptimizationParams
from synthetic import build_workload
from train import training
from utils.general import safe_state
from diff_gaussian_rasterization import _C as raster
import math

def stats(model, cams):
    bg = torch.zeros(3, device="cuda")
    raster.set_async(False)
    Rs = []
    for c in cams[:8]:
        out = raster.rasterize_gaussians_culled(bg, model.get_xyz, torch.empty(0, device="cuda"), model.get_opacity, model.get_scaling,
                                                model.get_rotation, 1.0, torch.empty(0, device="cuda"), c.world_view_transform,
                                                c.full_proj_transform, math.tan(c.FoVx * 0.5), math.tan(c.FoVy * 0.5), c.image_height,
                                                c.image_width, model.get_features, model.active_sh_degree, c.camera_center, False, False)
        Rs.append(out[0])
    return model.get_xyz.shape[0], sum(Rs) / len(Rs)

for recipe in ("raster_only", "bench"):
    safe_state(True)
    model, cams, extent = build_workload("north_star", device="cuda", seed=0, with_targets=True)
    opt = OptimizationParams()
    if recipe == "bench":
        opt.single_pass, opt.fused_step, opt.enable_topology = True, True, False
    model.training_setup(opt)
    if recipe == "bench":
        training(model, cams, opt, iterations=60, extent=extent, seed=0)     # (stands for the measurement protocol's steps)
        opt.enable_topology = True
    training(model, cams, opt, iterations=1000, extent=extent, start_iteration=10, seed=1, steps_per_graph=8)
    torch.cuda.synchronize()
    print(recipe, stats(model, cams))
