"""Drop-in for `simple_knn._C` (submodules/simple-knn/ext.cpp:15-17): distCUDA2(points[P,3]) -> float[P]."""
import torch

import hgs_runtime as rt


def distCUDA2(points):
    """Mean squared distance of every point to its 3 nearest neighbours (spatial.cu:15-26)."""
    points = rt.require_gpu_tensor(points, "points", torch.float32)
    P = points.shape[0]
    means = torch.zeros((P,), dtype=torch.float32, device=points.device)  # torch::full({P}, 0.0)
    if P == 0:
        return means
    L = rt.lib()
    nbytes = L.hgs_dist2_scratch_bytes(P)
    scratch = torch.empty((nbytes,), dtype=torch.uint8, device=points.device)
    with torch.cuda.device(points.device):
        rt.check(L.hgs_dist2(rt.current_stream(), P, rt.ptr(points), rt.ptr(means), rt.ptr(scratch), nbytes))
    return means
