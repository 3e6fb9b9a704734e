"""Fraction of 32x32 SSIM blocks whose halo tile is exactly zero in render AND target (forward fast path), and whose 3x3
block neighbourhood is (backward skip), on the bench workload."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import torch
import torch.nn.functional as F
from gaussian_renderer import render
from synthetic import build_workload
name = sys.argv[1] if len(sys.argv) > 1 else "north_star"
model, cams, extent = build_workload(name, device=torch.device("cuda"), seed=0, n_views=8)
bg = torch.zeros(3, device="cuda")
zs, ss = [], []
with torch.no_grad():
    for c in cams:
        img = render(c, model, bg)["render"]
        nzp = ((img != 0).any(0) | (c.original_image.cuda() != 0).any(0)).float()[None, None]
        nzp = F.max_pool2d(nzp, 17, 1, 8)                       # halo (5 rows, 8 columns of staging; conservative)
        H, W = nzp.shape[-2:]
        nzp = F.pad(nzp, (0, (-W) % 32, 0, (-H) % 32))
        nzb = F.max_pool2d(nzp, 32, 32)
        z = 1 - nzb
        s = 1 - F.max_pool2d(nzb, 3, 1, 1)
        zs.append(z.mean().item()); ss.append(s.mean().item())
print(name, "zero-tile blocks: %.3f   skippable (3x3 zero): %.3f" % (sum(zs) / len(zs), sum(ss) / len(ss)))
