import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import torch
from arguments import OptimizationParams
from diff_gaussian_rasterization import _C as raster
from synthetic import build_workload
from train import GraphedStep, training_step
from utils.general import safe_state
order = [1, 3, 0, 2, 1, 0]
res = {}
for mode in ("eager", "eager_capt", "graph"):
    safe_state(True)
    model, cams, extent = build_workload("tiny", device="cuda", with_targets=True)
    opt = OptimizationParams(); opt.enable_topology = False
    model.training_setup(opt)
    bg = torch.zeros(3, device="cuda")
    out = []
    if mode.startswith("eager"):
        if mode == "eager_capt":
            gs = GraphedStep(model, cams, opt, bg, extent=extent)   # capturable Adam + async, but eager execution
        for it, ci in enumerate(order, 1):
            if mode == "eager_capt":
                gs._set_lr(it); gs.load_camera(cams[ci]); l = gs._forward_backward(); model.optimizer.step(); model.optimizer.zero_grad(set_to_none=True); model._derived=None; raster.check_async()
            else:
                l, _, _ = training_step(model, cams[ci], opt, bg, it, extent=extent)
            out.append((float(l), float(model._endpoints.abs().sum()), float(model._opacity.abs().sum()), float(model._features_dc.abs().sum())))
    else:
        gs = GraphedStep(model, cams, opt, bg, extent=extent); gs.capture(cams)
        for it, ci in enumerate(order, 1):
            l = gs.step(cams[ci], it)
            out.append((float(l), float(model._endpoints.abs().sum()), float(model._opacity.abs().sum()), float(model._features_dc.abs().sum())))
    raster.set_async(False)
    res[mode] = out
for i in range(len(order)):
    print(i + 1, "cam", order[i])
    for m in res: print("   %-10s" % m, ["%.6f" % v for v in res[m][i]])
