"""Bisect which part of the step breaks HIP-graph capture.  usage: graph_probe.py <stage>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import torch
from arguments import OptimizationParams
from diff_gaussian_rasterization import _C as raster
from gaussian_renderer import render
from loss import losses as Ls
from synthetic import build_workload
stage = sys.argv[1]
model, cams, extent = build_workload("tiny", device="cuda", with_targets=True)
opt = OptimizationParams(); opt.enable_topology = False
model.training_setup(opt)
bg = torch.zeros(3, device="cuda")
cam = cams[1]
raster.set_async(stage != "fwd_blocking_warm")
def body():
    if stage == "memset":
        x = torch.empty(1000, device="cuda"); x.zero_(); return x
    if stage == "fwd":
        with torch.no_grad(): return render(cam, model, bg)["render"]
    if stage == "fwd_nopin":
        raster._state["async"] = True
        with torch.no_grad(): return render(cam, model, bg)["render"]
    if stage == "ssim":
        from hgs_runtime.fused import ssim_l1
        return ssim_l1(cam.original_image.clone(), cam.original_image)
    if stage == "fwdbwd":
        pkg = render(cam, model, bg); l = pkg["render"].sum(); l.backward(); return l
    if stage == "loss":
        pkg = render(cam, model, bg); l, _ = Ls.loss_function(model, pkg["render"], cam, opt); l.backward(); return l
    if stage == "adam":
        (model._opacity.sum() + model._endpoints.sum() + model._width.sum() + model._mask.sum() + model._features_dc.sum() + model._features_rest.sum()).backward()
        model.optimizer.step(); return None
if stage == "adam":
    for g in model.optimizer.param_groups:
        g["capturable"] = True
        g["lr"] = torch.tensor(float(g["lr"]), device="cuda")
if stage == "fwd_nopin":
    import diff_gaussian_rasterization._C as m
    # disable the pinned status copy
    orig = torch.Tensor.copy_
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        body(); model.optimizer.zero_grad(set_to_none=True); model._derived = None
        raster.check_async() if raster._state["pending"] else None
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
raster._state["pending"].clear()
g = torch.cuda.CUDAGraph()
print("capturing", stage, flush=True)
with torch.cuda.graph(g, stream=s):
    out = body()
print("captured", flush=True)
g.replay(); torch.cuda.synchronize()
print("replayed OK", stage, flush=True)
