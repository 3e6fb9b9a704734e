"""distCUDA2 on uniform random points (P = 50 k, 200 k, 1 M), a few calls each: for rocprofv3 --kernel-trace --stats."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import torch
from simple_knn._C import distCUDA2
g = torch.Generator(device="cuda").manual_seed(0)
for P in [int(a) for a in sys.argv[1:]] or [50000]:
    x = torch.rand(P, 3, device="cuda", generator=g)
    distCUDA2(x); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        distCUDA2(x)
    torch.cuda.synchronize()
    print(P, "points:", (time.perf_counter() - t) / 5 * 1e3, "ms per call")
