"""Host-side cost of the drop-in render() under no_grad (cProfile over all views, capacity mode)."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import torch
from diff_gaussian_rasterization import _C as raster
from gaussian_renderer import render
from synthetic import build_workload
model, cams, _ = build_workload(sys.argv[1] if len(sys.argv) > 1 else "north_star", device="cuda", with_targets=False)
bg = torch.zeros(3, device="cuda")
raster.set_async(True)
with torch.no_grad():
    for c in cams[:3]:
        render(c, model, bg)
    raster.check_async()
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for c in cams:
            render(c, model, bg)
        t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"ms/view: host {(t1 - t0) * 1e3 / len(cams):.4f}  total {(t2 - t0) * 1e3 / len(cams):.4f}")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5):
        for c in cams:
            render(c, model, bg)
    pr.disable()
    torch.cuda.synchronize()
    raster.check_async()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
