"""Runs only the fused SSIM/L1 forward + backward at 1080p (for rocprofv3 --pmc passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import torch
from hgs_runtime.fused import ssim_l1
H, W = 1080, 1920
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.rand(3, H, W, device="cuda", generator=g).requires_grad_(True)
b = torch.rand(3, H, W, device="cuda", generator=g)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    s, l = ssim_l1(a, b)
    (0.2 * (1 - s) + 0.8 * l).backward()
    a.grad = None
torch.cuda.synchronize()
print("done")
