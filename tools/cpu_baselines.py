#!/usr/bin/env python3
"""CPU-only paths of the repo timed on the host cores (reported baselines, not targets).
  B2  c_utils.filter_strand_list_segments      S strands x 100 segments  (the reference .pyx: tools/ref_cython_timing.py, container only)
  B3  Stage-II merge plumbing (merge.py)       1k Gaussians -> to_hair_gaussian_model -> strands info -> one
                                               compute_endpoint_pair_to_merge round -> merge_endpoint_pairs
  B4  strand metrics compute_metrics           200k-point prediction vs 200k-point GT, 4 threshold pairs
Prints one JSON object."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import numpy as np
import torch


def med(fn, n=5):
    ts = []
    for _ in range(n):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return float(np.median(ts))


def main():
    out = {"cores": os.cpu_count()}
    import c_utils
    rng = np.random.default_rng(0)
    for S in (2000, 10000):
        strands = np.empty(S, dtype=object)
        for j in range(S):
            strands[j] = rng.integers(0, 10**6, size=(100, 2)).astype(np.int64)
        out[f"B2_filter_strand_segments_S{S}_ms"] = med(lambda: c_utils.filter_strand_list_segments(strands), 7) * 1e3
    # ---- B3: merge plumbing on 1k Gaussians (config C1), CPU tensors
    from arguments import OptimizationParams
    from scene.gaussian_model import GaussianModel
    from torch import nn
    n = 1000
    g = torch.Generator().manual_seed(0)
    pts = torch.rand(n, 3, generator=g) * 0.05
    m = GaussianModel(sh_degree=0, device="cpu")
    m._xyz = nn.Parameter(pts)
    m._features_dc = nn.Parameter(torch.zeros(n, 1, 3)); m._features_rest = nn.Parameter(torch.zeros(n, 0, 3))
    sc = torch.full((n, 3), 1e-4); sc[:, 0] = 1.5e-3
    m._scaling = nn.Parameter(torch.log(sc)); q = torch.randn(n, 4, generator=g); m._rotation = nn.Parameter(q)
    m._opacity = nn.Parameter(torch.full((n, 1), 2.0)); m._mask = nn.Parameter(torch.full((n, 1), 2.0))
    m.max_radii2D = torch.zeros(n)
    m.ref_strand_root = pts[:50].numpy().copy()
    m.training_setup(OptimizationParams())
    t = time.perf_counter(); hair = m.to_hair_gaussian_model(); out["B3_to_hair_gaussian_model_ms"] = (time.perf_counter() - t) * 1e3
    hair.merge_dist_th, hair.merge_angle_th = 4e-3, 40
    t = time.perf_counter(); pairs = hair.compute_endpoint_pair_to_merge(); out["B3_compute_endpoint_pair_to_merge_ms"] = (time.perf_counter() - t) * 1e3
    t = time.perf_counter(); hair.merge_endpoint_pairs(pairs); hair.compute_strands_info(); out["B3_merge_and_strands_info_ms"] = (time.perf_counter() - t) * 1e3
    out["B3_pairs_merged"] = int(pairs.shape[0])
    # ---- B4: metrics on 200k vs 200k oriented points
    from loss.metrics import HairEvalData, compute_metrics
    from synthetic import strand_polylines
    sp = strand_polylines(2000, 100, seed=0)
    mid = 0.5 * (sp[:, 1:] + sp[:, :-1]).reshape(-1, 3)
    d = (sp[:, 1:] - sp[:, :-1]).reshape(-1, 3); d /= np.linalg.norm(d, axis=1, keepdims=True)
    sid = np.repeat(np.arange(2000), 100)
    gt = HairEvalData(mid.astype(np.float64), d.astype(np.float64), sid)
    pred = HairEvalData(mid + rng.normal(size=mid.shape) * 1e-3, d, sid)
    t = time.perf_counter(); res, labels = compute_metrics(pred, gt, bidirectional=True); out["B4_compute_metrics_200k_s"] = time.perf_counter() - t
    out["B4_f1"] = [float(x) for x in res["f1(b)"]]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
