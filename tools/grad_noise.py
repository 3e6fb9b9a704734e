"""Diagnostic: how far are GPU gradients / fp32-oracle gradients from the fp64 oracle on a strand scene?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import numpy as np
from oracle import hgs_oracle as O
from tests import scenes, gpu_util as G
from diff_gaussian_rasterization import _C
_C.set_tile_cull(False)   # the reference's lists: n_contrib counts positions in them (gradients are bit-identical either way)
for name, s in (("strands", scenes.strand_scene(n_strands=60, n_seg=60, W=256, H=144, seed=3)),
                ("blobs", scenes.random_scene(P=1500, W=160, H=96, seed=1, sh_degree=0)),
                # (found by tools/dev/fuzz_parity.py: screen-filling opaque Gaussians with exact depth ties -- 0.4 % of the cov3D
                # gradient's elements outside the parity test's element-wise bound, its budget is 0.2 %)
                ("huge_opaque_ties", scenes.random_scene(P=300, W=131, H=176, seed=120006742, sh_degree=1, scale_lo=0.03, scale_hi=0.6,
                                                         opacity_lo=0.9, opacity_hi=0.99, behind_frac=0.05, fovx_deg=100.0, depth_levels=4))):
    dpix = np.random.default_rng(123).normal(size=(3, s["H"], s["W"])).astype(np.float32)
    f32 = O.forward(s); f64 = O.forward(s, f64=True)
    fw = G.run_forward(s); got = G.intermediates(s, fw)
    print(name, "n_contrib mismatches gpu/f32:", (got["n_contrib"] != f32["n_contrib"]).sum(), " f64/f32:", (f64["n_contrib"] != f32["n_contrib"]).sum())
    g32 = O.backward(s, f32, dpix); g64 = O.backward(s, f64, dpix, f64=True)
    g = G.run_backward(s, fw, dpix)
    for k in ("dL_dmeans2D", "dL_dconic", "dL_dopacity", "dL_dcolors", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations"):
        ref = g64[k].astype(np.float64); sc = np.abs(ref).max()
        e_gpu = np.abs(g[k].reshape(ref.shape) - ref); e_32 = np.abs(g32[k].astype(np.float64) - ref)
        print(f"  {k:14s} scale {sc:10.3e}  gpu max {e_gpu.max()/sc:9.2e} mean {e_gpu.mean()/sc:9.2e} | oracle32 max {e_32.max()/sc:9.2e} mean {e_32.mean()/sc:9.2e}")
