import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import torch
from arguments import OptimizationParams
from diff_gaussian_rasterization import _C as raster
from synthetic import build_workload
from train import GraphedStep, ViewSampler
wl, nv = sys.argv[1], int(sys.argv[2])
model, cams, extent = build_workload(wl, device="cuda", n_views=nv)
opt = OptimizationParams(); opt.enable_topology = False
model.training_setup(opt)
bg = torch.zeros(3, device="cuda")
gs = GraphedStep(model, cams, opt, bg, extent=extent)
gs.capture(cams); print("captured; cap", raster._state["cap"], flush=True)
sm = ViewSampler(cams, seed=0)
for it in range(1, 13):
    l = gs.step(sm.next(), it); torch.cuda.synchronize(); print(it, float(l), gs.check(), flush=True)
