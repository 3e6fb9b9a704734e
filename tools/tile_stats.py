"""Tile-list statistics of a synthetic workload: list-length distribution, quadrant-cull rate, critical path.
Runs the 3-channel forward through the C-ABI and reads the binning / image scratch back (diagnostics only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import hgs_runtime as rt
from diff_gaussian_rasterization import _C
from synthetic import build_workload
from utils.sh import RGB2SH  # noqa: F401

wl = sys.argv[1] if len(sys.argv) > 1 else "north_star"
model, cams, _ = build_workload(wl, device="cuda", with_targets=False, n_views=4)
cam = cams[0]
H, W = cam.image_height, cam.image_width
with torch.no_grad():
    import math
    tfx, tfy = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
    R, color, radii, geom, binning, img = _C.rasterize_gaussians(
        torch.zeros(3, device="cuda"), model.get_xyz, torch.empty(0, device="cuda"), model.get_opacity, model.get_scaling,
        model.get_rotation, 1.0, torch.empty(0, device="cuda"), cam.world_view_transform, cam.full_proj_transform, tfx, tfy,
        H, W, model.get_features, model.active_sh_degree, cam.camera_center, False, False)
torch.cuda.synchronize()
im = rt.layout("image", W, H)
ib = img.cpu().numpy()
T = ((W + 15) // 16) * ((H + 15) // 16)
ranges = ib[im["ranges"]:im["ranges"] + 8 * T].view(np.uint32).reshape(T, 2)
L = (ranges[:, 1] - ranges[:, 0]).astype(np.int64)
print("R", R, "tiles", T, "nonempty", int((L > 0).sum()), "mean L (nonempty)", L[L > 0].mean(), "max", L.max())
print("percentiles 50/90/99/99.9:", np.percentile(L[L > 0], [50, 90, 99, 99.9]))
hist = np.bincount(np.minimum(L // 64, 40))
print("L//64 histogram:", hist.tolist())
n_contrib = ib[im["n_contrib"]:im["n_contrib"] + 4 * W * H].view(np.uint32).reshape(H, W)
print("mean last contributor", n_contrib.mean(), "max", n_contrib.max())
b = rt.layout("binning", R)
bb = binning.cpu().numpy()
rec = bb[b["packed"]:b["packed"] + 48 * R].view(np.uint32).reshape(R, 12)
qm = rec[:, 10]
pop = np.array([bin(int(x)).count("1") for x in range(16)])[qm & 15]
print("quadrant mask popcount histogram:", np.bincount(pop, minlength=5).tolist(), "kept fraction", pop.sum() / (4.0 * R))
# entries a wave actually visits before its pixels are all done: per tile max n_contrib
ty, tx = (H + 15) // 16, (W + 15) // 16
pad = np.zeros((ty * 16, tx * 16), np.uint32); pad[:H, :W] = n_contrib
last = pad.reshape(ty, 16, tx, 16).max(axis=(1, 3)).reshape(-1)
print("sum L", L.sum(), "sum tile-max last contributor", last.sum())
# exact per-(entry, quadrant) visibility: does any of the quadrant's 64 pixels reach alpha >= 1/255?
recf = torch.from_numpy(rec.view(np.float32).copy()).cuda()
tile_of = torch.from_numpy(np.repeat(np.arange(T), L).astype(np.int64)).cuda()
tx0 = (tile_of % tx) * 16; ty0 = (tile_of // tx) * 16
ii = torch.arange(16, device="cuda", dtype=torch.float32)
dx = recf[:, 0, None] - (tx0[:, None] + ii[None, :])          # [R,16]
dy = recf[:, 1, None] - (ty0[:, None] + ii[None, :])
A, B, Cc, op = recf[:, 2], recf[:, 3], recf[:, 4], recf[:, 5]
power = -0.5 * (A[:, None, None] * dx[:, None, :] ** 2 + Cc[:, None, None] * dy[:, :, None] ** 2) - B[:, None, None] * dx[:, None, :] * dy[:, :, None]
vis = (power <= 0) & (op[:, None, None] * torch.exp(power) >= 1.0 / 255.0)   # [R,16(y),16(x)]
q = vis.reshape(-1, 2, 8, 2, 8).any(dim=4).any(dim=2)                          # [R, qy, qx]
print("instances with at least one passing pixel (exact tile-level keep):", vis.reshape(vis.shape[0], -1).any(dim=1).float().mean().item())
print("exact kept fraction", q.float().mean().item(), " pixels passing per kept (wave, entry):", vis.float().sum().item() / max(1.0, q.float().sum().item()))
# ---- what a 4x4-block (DPP row) work decomposition would do: per (tile, wave=quadrant, batch of 64 entries) the step
# count is the max over the quadrant's four 4x4 blocks of the entries whose footprint box reaches the block
tau = torch.log(255.0 * op)
det = A * Cc - B * B
hx = torch.sqrt(2 * tau.clamp(min=0) * (Cc / det)) * 1.0005 + 0.01
hy = torch.sqrt(2 * tau.clamp(min=0) * (A / det)) * 1.0005 + 0.01
valid = (tau > 0)
bi = torch.arange(4, device="cuda", dtype=torch.float32)
bx0 = tx0[:, None].float() + 4 * bi[None, :]            # [R,4] block x origins
by0 = ty0[:, None].float() + 4 * bi[None, :]
ox = (recf[:, 0, None] + hx[:, None] >= bx0) & (recf[:, 0, None] - hx[:, None] <= bx0 + 3)   # [R,4]
oy = (recf[:, 1, None] + hy[:, None] >= by0) & (recf[:, 1, None] - hy[:, None] <= by0 + 3)
blk = (oy[:, :, None] & ox[:, None, :]) & valid[:, None, None]                                  # [R, by, bx]
print("4x4 block kept fraction (box test)", blk.float().mean().item())
visb = vis.reshape(-1, 4, 4, 4, 4).any(dim=4).any(dim=2)                                        # exact, [R, by, bx]
print("4x4 block kept fraction (exact)", visb.float().mean().item(), "active pixels per kept block-pair",
      vis.float().sum().item() / max(1.0, visb.float().sum().item()))
# steps: group by (tile, batch of 64 positions), per quadrant max over its 4 blocks
pos_in_tile = torch.arange(R, device="cuda") - torch.from_numpy(np.repeat(ranges[:, 0].astype(np.int64), L)).cuda()
grp = tile_of * 64 + pos_in_tile // 64                     # (tile, batch) id
ng = int(grp.max().item()) + 1
cnt = torch.zeros((ng, 4, 4), device="cuda")
cnt.index_add_(0, grp, blk.float())
q = cnt.reshape(ng, 2, 2, 2, 2)                            # [g, qy, by, qx, bx]
steps_new = q.amax(dim=(2, 4)).sum().item()
pairs_quad = (cnt.reshape(ng, 2, 2, 2, 2).sum(dim=(2, 4)) > 0)
quad_kept = torch.zeros((ng, 2, 2), device="cuda")
qm_t = torch.from_numpy(((qm[:, None] >> np.arange(4)[None, :]) & 1).astype(np.float32)).cuda().reshape(-1, 2, 2)
quad_kept.index_add_(0, grp, qm_t)
print("quadrant pairs (current steps)", quad_kept.sum().item(), " 4x4-row steps", steps_new, " block pairs", blk.float().sum().item(),
      " ratio old*87 / new*150 =", quad_kept.sum().item() * 87 / max(1.0, steps_new * 150))
