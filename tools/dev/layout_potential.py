"""How many (wavefront, entry) pairs would the blend kernels evaluate under other pixel-to-wavefront layouts of a 16 x 16 tile?
CPU estimate from the oracle's lists (preprocess + binning, no GPU): for every (tile, entry) the pixels that pass the alpha test
(alpha >= 1/255, power <= 0; transmittance ignored), then the number of 64-pixel regions with at least one such pixel for
  quadrants (four 8 x 8: what the kernels use), rows (four 16 x 4 strips), columns (four 4 x 16 strips),
  and the best of the three chosen PER TILE (one layout for all entries of a tile).
  python tools/dev/layout_potential.py [workload=north_star]"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import numpy as np, torch
from oracle import hgs_oracle as O
from synthetic import build_workload
wl = sys.argv[1] if len(sys.argv) > 1 else "north_star"
model, cams, _ = build_workload(wl, device="cpu", seed=0, with_targets=False, n_views=2)
cam = cams[0]
with torch.no_grad():
    s = dict(means3D=model.get_xyz.numpy(), opacities=model.get_opacity.numpy().reshape(-1), scales=model.get_scaling.numpy(),
             rotations=model.get_rotation.numpy(), cov3D_precomp=None, viewmatrix=cam.world_view_transform.numpy(),
             projmatrix=cam.full_proj_transform.numpy(), campos=cam.camera_center.numpy(), bg=np.zeros(3, np.float32),
             tanfovx=float(math.tan(cam.FoVx * 0.5)), tanfovy=float(math.tan(cam.FoVy * 0.5)), W=cam.image_width, H=cam.image_height,
             sh_degree=model.active_sh_degree, scale_modifier=1.0, shs=model.get_features.numpy(), colors_precomp=None)
O.set_threads(8)
f = O.forward(s, render=False)
W, H = s["W"], s["H"]
gx = (W + 15) // 16
ranges, pl = f["ranges"], f["point_list"]
xy, co = f["means2D"], f["conic_opacity"]
n_tiles = ranges.shape[0]
tile_of = np.repeat(np.arange(n_tiles), (ranges[:, 1] - ranges[:, 0]).astype(np.int64))
R = len(pl)
print(wl, "entries", R)
py, px = np.mgrid[0:16, 0:16]
px, py = px.reshape(-1).astype(np.float32), py.reshape(-1).astype(np.float32)
lay = {"quadrants": ((py >= 8).astype(int) * 2 + (px >= 8).astype(int)), "rows 16x4": (py // 4).astype(int), "cols 4x16": (px // 4).astype(int),
       "rows 32x2 pairs": (py // 4).astype(int)}
del lay["rows 32x2 pairs"]
counts = {k: np.zeros(R, np.int8) for k in lay}
active = np.zeros(R, np.int16)
CH = 20000
for a in range(0, R, CH):
    b = min(R, a + CH)
    g = pl[a:b]
    t = tile_of[a:b]
    ox, oy = (t % gx) * 16, (t // gx) * 16
    dx = xy[g, 0][:, None] - (ox[:, None] + px[None])
    dy = xy[g, 1][:, None] - (oy[:, None] + py[None])
    c = co[g]
    power = -0.5 * (c[:, 0:1] * dx * dx + c[:, 2:3] * dy * dy) - c[:, 1:2] * dx * dy
    alpha = np.minimum(0.99, c[:, 3:4] * np.exp(power))
    inside = ((ox[:, None] + px[None]) < W) & ((oy[:, None] + py[None]) < H)
    act = (power <= 0) & (alpha >= 1.0 / 255.0) & inside
    active[a:b] = act.sum(1)
    for k, region in lay.items():
        cnt = np.zeros(b - a, np.int8)
        for r in range(4):
            cnt += act[:, region == r].any(1)
        counts[k][a:b] = cnt
tot = {k: int(v.sum()) for k, v in counts.items()}
print("blending pixels per entry (no transmittance):", round(float(active.mean()), 1), " entries without any:", int((active == 0).sum()))
for k, v in tot.items():
    print(f"  {k:12s} pairs {v:9d} = {v / R:.3f} per entry, blending lanes per pair {active.sum() / max(v, 1):.1f}")
# best single layout per tile
per_tile = {k: np.bincount(tile_of, weights=v.astype(np.float64), minlength=n_tiles) for k, v in counts.items()}
best = np.minimum.reduce([per_tile[k] for k in per_tile])
print(f"  best per tile pairs {int(best.sum()):9d} = {best.sum() / R:.3f} per entry ({100 * (1 - best.sum() / tot['quadrants']):.1f} % fewer than quadrants);"
      f" tiles choosing quadrants / rows / cols:", [int((per_tile[k] == best).sum()) for k in per_tile])
