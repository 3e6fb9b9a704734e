"""Wall-clock breakdown of one training iteration (synchronising after each phase; diagnostic only)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import torch
from arguments import OptimizationParams
from gaussian_renderer import render
from loss import losses as Ls
from synthetic import build_workload
wl = sys.argv[1] if len(sys.argv) > 1 else "north_star"
model, cams, extent = build_workload(wl, device="cuda", n_views=4)
opt = OptimizationParams(); opt.enable_topology = False
model.training_setup(opt)
bg = torch.zeros(3, device="cuda")
def T(fn, n=5):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3, r
cam = cams[1]
for _ in range(3):
    pkg = render(cam, model, bg); l, _ = Ls.loss_function(model, pkg["render"], cam, opt); l.backward(); model.optimizer.step(); model.optimizer.zero_grad(set_to_none=True)
print("getters: xyz %.2f scaling %.2f rotation %.2f orientation %.2f features %.2f ms" % tuple(T(lambda: getattr(model, a))[0] for a in ("get_xyz","get_scaling","get_rotation","get_orientation","get_features")))
t, pkg = T(lambda: render(cam, model, bg)); print("render fwd (with getters, autograd graph) %.2f ms" % t)
with torch.no_grad():
    t, _ = T(lambda: render(cam, model, bg)); print("render fwd no_grad %.2f ms" % t)
img = pkg["render"]; gt = cam.original_image
print("l1 %.2f  ssim %.2f  mask_loss %.2f  orient_loss %.2f  smooth %.2f ms (forward only)" % (
    T(lambda: Ls.l1_loss(img, gt))[0], T(lambda: Ls.ssim(img, gt))[0], T(lambda: Ls.mask_loss_rast(model, cam, opt))[0],
    T(lambda: Ls.orientation_loss_rast(model, cam, opt))[0], T(lambda: Ls.angle_smoothness_loss(model))[0]))
def full():
    pkg = render(cam, model, bg); l, _ = Ls.loss_function(model, pkg["render"], cam, opt); return l
t, l = T(full); print("forward total (render + all losses) %.2f ms" % t)
def fb():
    l = full(); l.backward(); return l
t, _ = T(fb); print("forward+backward %.2f ms" % t)
def ssim_fb():
    x = img.detach().clone().requires_grad_(True); (1 - Ls.ssim(x, gt)).backward()
t, _ = T(ssim_fb); print("ssim fwd+bwd alone %.2f ms" % t)
t, _ = T(lambda: model.optimizer.step()); print("adam step %.2f ms" % t)
