#!/usr/bin/env python3
"""Where the instances of a state go (dev aid for the per-Gaussian kernels on many-tile states):
  python tools/dev/instance_stats.py [workload | stage1_1080p | stage3_merged] [views=4]
Per view (tile culling on, the training step's lists): instances per Gaussian (mean, percentiles, max), per wavefront of 64
consecutive Gaussians the sum and the largest lane (what a lane-per-Gaussian row loop waits for), the share of instances whose
quadrant mask is empty (the ellipse enters none of the tile's four 8x8 quadrants: nobody blends them), and the share that lies
behind its tile's last contributor (zero rows of the backward)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import hgs_runtime as rt  # noqa: E402
from synthetic import PIPELINE_STATES, build_pipeline_state, build_workload  # noqa: E402
from tests import gpu_util as G  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "north_star"
nv = int(sys.argv[2]) if len(sys.argv) > 2 else 4
_out, sys.stdout = sys.stdout, sys.stderr
if wl in PIPELINE_STATES:
    model, cams, extent, info = build_pipeline_state(wl, device="cuda")
else:
    model, cams, extent = build_workload(wl, device="cuda", with_targets=False, n_views=nv)
    info = None
from diff_gaussian_rasterization import _C  # noqa: E402
import math  # noqa: E402
_C.set_async(False)
res = []
bg = torch.zeros(3, device="cuda")
for c in cams[:nv]:
    was = _C.set_tile_cull(True)
    with torch.no_grad():
        out = _C.rasterize_gaussians(bg, model.get_xyz, torch.empty(0, device="cuda"), model.get_opacity, model.get_scaling,
                                     model.get_rotation, 1.0, torch.empty(0, device="cuda"), c.world_view_transform,
                                     c.full_proj_transform, math.tan(c.FoVx * 0.5), math.tan(c.FoVy * 0.5), c.image_height,
                                     c.image_width, model.get_features, model.active_sh_degree, c.camera_center, False, False)
    _C.set_tile_cull(was)
    torch.cuda.synchronize()
    R, _, radii, geom, binning, img = out
    P = model.get_xyz.shape[0]
    W, H = c.image_width, c.image_height
    T = ((W + 15) // 16) * ((H + 15) // 16)
    gl, il, bl = rt.layout("geom", P), rt.layout("image", W, H), rt.layout("binning", R)
    n = G._view(geom, gl["tiles_touched"], P, np.uint32).astype(np.int64)
    ks = G._view(binning, bl["keys_sorted"], R, np.uint64)
    ranges = G._view(img, il["ranges"], 2 * T, np.uint32).reshape(T, 2).astype(np.int64)
    maxc = G._view(img, il["tile_maxc"], T, np.uint32).astype(np.int64)
    lens = ranges[:, 1] - ranges[:, 0]
    empty_mask = int(((ks & np.uint64(15)) == 0).sum())
    pad = (-P) % 64
    nw = np.concatenate([n, np.zeros(pad, np.int64)]).reshape(-1, 64)
    pad2 = (-P) % 256
    nb = np.concatenate([n, np.zeros(pad2, np.int64)]).reshape(-1, 256)
    vis = n > 0
    q = lambda a: [float(x) for x in np.percentile(a, [50, 90, 99, 99.9, 100])]
    res.append({"gaussians": P, "visible": int(vis.sum()), "instances": int(R), "instances_per_visible_gaussian_p50_p90_p99_p99.9_max": q(n[vis]),
                "mean_instances_per_visible_gaussian": float(n[vis].mean()),
                "wave_sum_p50_p90_p99_p99.9_max": q(nw.sum(1)), "wave_max_lane_p50_p90_p99_p99.9_max": q(nw.max(1)),
                "mean_wave_max_lane": float(nw.max(1).mean()), "block_sum_p50_p90_p99_p99.9_max": q(nb.sum(1)),
                "share_of_instances_in_gaussians_above_32": float(n[n > 32].sum() / max(R, 1)),
                "share_of_instances_in_gaussians_above_128": float(n[n > 128].sum() / max(R, 1)),
                "empty_quadrant_mask_share": empty_mask / max(R, 1),
                "behind_last_contributor_share": float((lens - np.minimum(lens, maxc)).sum() / max(R, 1)),
                "tiles_nonempty": int((lens > 0).sum()), "tile_len_p50_p90_p99_max": [float(x) for x in np.percentile(lens[lens > 0], [50, 90, 99, 100])]})
    print(res[-1], flush=True)
sys.stdout = _out
print(json.dumps({"workload": wl, "state": info, "views": res}, indent=1))
