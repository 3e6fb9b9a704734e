"""FrameRenderer over all views of a workload, N rounds (for rocprofv3 --kernel-trace --stats)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import torch
from gaussian_renderer.frames import FrameRenderer
from synthetic import build_workload
wl = sys.argv[1] if len(sys.argv) > 1 else "north_star"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
K = int(sys.argv[3]) if len(sys.argv) > 3 else 8
model, cams, _ = build_workload(wl, device="cuda", with_targets=False)
bg = torch.zeros(3, device="cuda")
fr = FrameRenderer(model, cams, bg, frames_per_launch=K)
fr.render(0)
for r in range(rounds + 1):
    if r == 1:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    if K > 1:
        for i in range(0, len(cams), K):
            fr.render_batch(list(range(i, i + K)), check=False)
    else:
        for i in range(len(cams)):
            fr.render(i, check=False)
    assert fr.validate() == []
torch.cuda.synchronize()
print("ms/view", (time.perf_counter() - t0) * 1e3 / (rounds * len(cams)))
