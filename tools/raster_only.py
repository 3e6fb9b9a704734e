"""Runs only the single-pass rasterizer forward+backward on a workload (for rocprofv3 --pmc passes):
  python tools/raster_only.py [workload] [passes] [culled | train=N]
train=N: on the state N iterations of the FULL loop leave (densification, merging, opacity reset: bench.py's `trained_state`)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import torch
from gaussian_renderer import render_multi
from synthetic import build_workload
wl = sys.argv[1] if len(sys.argv) > 1 else "north_star"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n_train = int(sys.argv[3].split("=")[1]) if len(sys.argv) > 3 and sys.argv[3].startswith("train=") else 0
# (train=N: the workload's own views and the seeds of bench.py's trained_state leg, so that the state is the one that leg times
# as closely as a run-to-run different trajectory allows)
model, cams, extent = build_workload(wl, device="cuda", with_targets=n_train > 0, n_views=None if n_train else 4)
if n_train:
    from arguments import OptimizationParams
    from train import training
    from utils.general import safe_state
    safe_state(True)
    opt = OptimizationParams()
    model.training_setup(opt)
    training(model, cams, opt, iterations=n_train, extent=extent, start_iteration=10, seed=1, steps_per_graph=8)
    torch.cuda.synchronize()
    print("trained", n_train, "iterations:", model.get_xyz.shape[0], "segments")
if len(sys.argv) > 3 and sys.argv[3] == "culled":   # everything behind the camera: launch + output-write floor of the kernels
    with torch.no_grad():
        model._endpoints.data += 1.0e4
bg = torch.zeros(3, device="cuda")
H, W = cams[0].image_height, cams[0].image_width
w3 = torch.randn(3, H, W, device="cuda"); w1 = torch.randn(H, W, device="cuda"); wo = torch.randn(3, H, W, device="cuda")
for i in range(n):
    cam = cams[i % len(cams)]
    extra = torch.cat((model.get_mask, model.get_orientation), dim=1)
    pkg = render_multi(cam, model, bg, extra, splits=(1, 3), black_background=True)   # bg is the zeros made above
    loss = (pkg["render"] * w3).sum() + (pkg["extra"][0] * w1).sum() + (pkg["extra"][1] * wo).sum()
    loss.backward()
    for p in (model._endpoints, model._features_dc, model._opacity, model._mask, model._width):
        p.grad = None
torch.cuda.synchronize()
print("done")
