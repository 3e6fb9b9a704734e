"""Runs only the single-pass rasterizer forward+backward on a workload (for rocprofv3 --pmc passes):
  python tools/raster_only.py [workload] [passes] [culled | train=N]
train=N: on the state N iterations of the FULL loop leave (densification, merging, opacity reset: bench.py's `trained_state`).
workload stage1_1080p / stage3_merged (synthetic.PIPELINE_STATES): the states bench.py's `pipeline_states` legs time (the Stage-I
loop that leads to them runs in this process first: average the LAST dispatches only, tools/pmc_aggregate.py <out> <passes>)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import torch
from gaussian_renderer import render_multi
from synthetic import PIPELINE_STATES, build_pipeline_state, build_workload
wl = sys.argv[1] if len(sys.argv) > 1 else "north_star"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n_train = int(sys.argv[3].split("=")[1]) if len(sys.argv) > 3 and sys.argv[3].startswith("train=") else 0
# (train=N: the workload's own views and the seeds of bench.py's trained_state leg, so that the state is the one that leg times
# as closely as a run-to-run different trajectory allows)
if wl in PIPELINE_STATES:
    model, cams, extent, info = build_pipeline_state(wl, device="cuda")
    print(info)
else:
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
    for g_ in model.optimizer.param_groups if model.optimizer is not None else []:
        for p in g_["params"]:
            p.grad = None
    for name in ("_endpoints", "_features_dc", "_opacity", "_mask", "_width", "_xyz", "_scaling", "_rotation"):
        if hasattr(model, name):
            getattr(model, name).grad = None
torch.cuda.synchronize()
print("done")
