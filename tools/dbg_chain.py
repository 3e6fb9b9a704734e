import sys, os, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import numpy as np, torch
from gaussian_renderer import render
from oracle import hgs_oracle as O
from synthetic import build_workload
model, cams, _ = build_workload("tiny", device="cuda", with_targets=False)
cam = cams[1]
bg = torch.tensor([0.1, 0.2, 0.3], device="cuda")
H, W = cam.image_height, cam.image_width
w = torch.randn(3, H, W, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0))
xyz, sc, rot, op, ft = model.get_xyz, model.get_scaling, model.get_rotation, model.get_opacity, model.get_features
for t in (xyz, sc, rot, op, ft): t.retain_grad()
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
rs = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx*0.5), math.tan(cam.FoVy*0.5), bg, 1.0, cam.world_view_transform, cam.full_proj_transform, 0, cam.camera_center, False, False)
sp = torch.zeros_like(xyz, requires_grad=True)
img, radii = GaussianRasterizer(rs)(means3D=xyz, means2D=sp, shs=ft, colors_precomp=None, opacities=op, scales=sc, rotations=rot, cov3D_precomp=None)
(img * w).sum().backward()
s = dict(means3D=xyz.detach().cpu().numpy(), opacities=op.detach().cpu().numpy().reshape(-1), scales=sc.detach().cpu().numpy(), rotations=rot.detach().cpu().numpy(),
         shs=ft.detach().cpu().numpy(), colors_precomp=None, cov3D_precomp=None, viewmatrix=cam.world_view_transform.cpu().numpy(), projmatrix=cam.full_proj_transform.cpu().numpy(),
         campos=cam.camera_center.cpu().numpy(), bg=bg.cpu().numpy(), tanfovx=math.tan(cam.FoVx*0.5), tanfovy=math.tan(cam.FoVy*0.5), W=W, H=H, sh_degree=0, scale_modifier=1.0)
f = O.forward(s); g = O.backward(s, f, w.cpu().numpy())
for name, t, k in (("xyz", xyz, "dL_dmeans3D"), ("scales", sc, "dL_dscales"), ("rot", rot, "dL_drotations"), ("op", op, "dL_dopacity"), ("sh", ft, "dL_dsh")):
    a = t.grad.cpu().numpy().reshape(g[k].shape); b = g[k]
    print(name, np.abs(a-b).max(), np.abs(b).max())
ep_grad = model._endpoints.grad.clone()
model._endpoints.grad = None
chain = (model.get_xyz * xyz.grad).sum() + (model.get_scaling * sc.grad).sum() + (model.get_rotation * rot.grad).sum()
chain.backward()
print("endpoint grad autograd vs re-chain:", (ep_grad - model._endpoints.grad).abs().max().item(), ep_grad.abs().max().item())
print("min seg len", (model._endpoints[model.endpoint_pairs][:,1]-model._endpoints[model.endpoint_pairs][:,0]).norm(dim=1).min().item())
from tests import gpu_util as G
fw = G.run_forward(s); got = G.intermediates(s, fw)
gg = G.run_backward(s, fw, w.cpu().numpy())
for k in ("dL_dmeans2D","dL_dconic","dL_dopacity","dL_dcolors","dL_dcov3D","dL_dmeans3D","dL_dscales","dL_drotations"):
    a = gg[k].reshape(g[k].shape); b = g[k]
    e = np.abs(a-b); i = np.unravel_index(e.argmax(), e.shape)
    print(k, "maxerr", e.max(), "scale", np.abs(b).max(), "at", i, a[i], b[i])
i = np.abs(gg["dL_dscales"]-g["dL_dscales"]).max(1).argmax()
print("worst gaussian", i, "radii", f["radii"][i], "tiles", f["tiles_touched"][i], "scales", s["scales"][i], "rot", s["rotations"][i])
print(" gpu dcov", gg["dL_dcov3D"][i], "\n ref dcov", g["dL_dcov3D"][i])
print(" gpu dscale", gg["dL_dscales"][i], "ref", g["dL_dscales"][i], " gpu drot", gg["dL_drotations"][i], "ref", g["dL_drotations"][i])
print(" cov3D gpu", got["cov3D"][i], "ref", f["cov3D"][i])
