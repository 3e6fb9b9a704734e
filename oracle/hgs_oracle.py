"""ctypes front-end of the CPU oracle (oracle/*.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (hair-gs_amd/) never imports this module.

All arrays are numpy, C-contiguous.  `f64=True` selects the double-precision build of the same
C source (used for finite-difference checks); depth keys are always fp32 bits.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libhgs_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("raster_oracle.c", "knn_oracle.c", "strand_oracle.c", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.hgs_oracle_filter_strand_segments.restype = C.c_int64
    return _LIB


def set_threads(n):
    """OpenMP thread count used by the oracle (cpu_baseline reports this as `cores`)."""
    omp = C.CDLL("libgomp.so.1")
    omp.omp_set_num_threads(int(n))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _arr(x, dt, shape=None):
    if x is None:
        return None
    a = np.ascontiguousarray(np.asarray(x, dtype=dt))
    if shape is not None:
        a = a.reshape(shape)
    return a


def _real(f64):
    return (np.float64, C.c_double, "_f64") if f64 else (np.float32, C.c_float, "_f32")


def normalize_inputs(inp, f64=False):
    dt, _, _ = _real(f64)
    P = int(np.asarray(inp["means3D"]).shape[0])
    out = dict(inp)
    out["means3D"] = _arr(inp["means3D"], dt, (P, 3))
    shs = inp.get("shs")
    out["shs"] = None if shs is None else _arr(shs, dt)
    out["M"] = 0 if shs is None else int(out["shs"].shape[1])
    out["colors_precomp"] = _arr(inp.get("colors_precomp"), dt)
    out["opacities"] = _arr(inp["opacities"], dt, (P,))
    out["scales"] = _arr(inp.get("scales"), dt)
    out["rotations"] = _arr(inp.get("rotations"), dt)
    out["cov3D_precomp"] = _arr(inp.get("cov3D_precomp"), dt)
    out["viewmatrix"] = _arr(inp["viewmatrix"], dt, (16,))
    out["projmatrix"] = _arr(inp["projmatrix"], dt, (16,))
    out["campos"] = _arr(inp["campos"], dt, (3,))
    out["bg"] = _arr(inp["bg"], dt, (3,))
    out["scale_modifier"] = float(inp.get("scale_modifier", 1.0))
    out["sh_degree"] = int(inp.get("sh_degree", 0))
    out["P"] = P
    return out


def forward(inp, f64=False, render=True):
    """Full forward: preprocess -> binning -> blend.  Returns every intermediate."""
    dt, cr, sfx = _real(f64)
    L = lib()
    n = normalize_inputs(inp, f64)
    P, W, H = n["P"], int(n["W"]), int(n["H"])
    gx, gy = (W + 15) // 16, (H + 15) // 16
    o = {
        "radii": np.zeros(P, np.int32), "means2D": np.zeros((P, 2), dt), "depths": np.zeros(P, dt),
        "cov3D": np.zeros((P, 6), dt), "conic_opacity": np.zeros((P, 4), dt), "rgb": np.zeros((P, 3), dt),
        "clamped": np.zeros((P, 3), np.uint8), "tiles_touched": np.zeros(P, np.uint32),
        "point_offsets": np.zeros(P, np.uint32),
    }
    f = getattr(L, "hgs_oracle_preprocess" + sfx)
    f.restype = C.c_int
    R = f(C.c_int(P), C.c_int(n["sh_degree"]), C.c_int(n["M"]), C.c_int(W), C.c_int(H), _p(n["means3D"]), _p(n["shs"]),
          _p(n["colors_precomp"]), _p(n["opacities"]), _p(n["scales"]), cr(n["scale_modifier"]), _p(n["rotations"]),
          _p(n["cov3D_precomp"]), _p(n["viewmatrix"]), _p(n["projmatrix"]), _p(n["campos"]), cr(n["tanfovx"]),
          cr(n["tanfovy"]), _p(o["radii"]), _p(o["means2D"]), _p(o["depths"]), _p(o["cov3D"]), _p(o["conic_opacity"]),
          _p(o["rgb"]), _p(o["clamped"]), _p(o["tiles_touched"]), _p(o["point_offsets"])) if P > 0 else 0
    if n["cov3D_precomp"] is not None:
        o["cov3D"] = n["cov3D_precomp"].reshape(P, 6).copy()
    o["num_rendered"] = int(R)
    o["keys_sorted"] = np.zeros(R, np.uint64)
    o["point_list"] = np.zeros(R, np.uint32)
    o["ranges"] = np.zeros((gx * gy, 2), np.uint32)
    if P > 0:
        getattr(L, "hgs_oracle_bin" + sfx)(C.c_int(P), C.c_int(W), C.c_int(H), C.c_int(R), _p(o["radii"]),
                                           _p(o["means2D"]), _p(o["depths"]), _p(o["point_offsets"]),
                                           _p(o["keys_sorted"]), _p(o["point_list"]), _p(o["ranges"]))
    o["out_color"] = np.zeros((3, H, W), dt)
    o["final_T"] = np.zeros((H, W), dt)
    o["n_contrib"] = np.zeros((H, W), np.uint32)
    if render:
        feats = n["colors_precomp"] if n["colors_precomp"] is not None else o["rgb"]
        o["features"] = feats
        getattr(L, "hgs_oracle_render" + sfx)(C.c_int(W), C.c_int(H), _p(o["ranges"]), _p(o["point_list"]),
                                              _p(o["means2D"]), _p(feats), _p(o["conic_opacity"]), _p(n["bg"]),
                                              _p(o["final_T"]), _p(o["n_contrib"]), _p(o["out_color"]))
    return o


def backward(inp, fwd, dL_dpix, f64=False):
    """Full backward given the forward's intermediates.  Output names/shapes follow
    DGR/rasterize_points.cu:151-159 (dL_dconic is [P,2,2] with element [1,0] unused)."""
    dt, cr, sfx = _real(f64)
    L = lib()
    n = normalize_inputs(inp, f64)
    P, W, H, M = n["P"], int(n["W"]), int(n["H"]), n["M"]
    dpix = _arr(dL_dpix, dt, (3, H, W))
    feats = n["colors_precomp"] if n["colors_precomp"] is not None else fwd["rgb"]
    acc = np.zeros((P, 9), np.float64)
    fragile = np.zeros(P, np.uint8)   # Gaussians with a near-threshold (pixel, entry) decision (raster_oracle.c)
    touched = np.zeros(P, np.uint8)   # Gaussians blended at a pixel that holds such a decision (its transmittance chain moves)
    getattr(L, "hgs_oracle_render_backward" + sfx)(C.c_int(P), C.c_int(W), C.c_int(H), _p(fwd["ranges"]),
                                                   _p(fwd["point_list"]), _p(n["bg"]), _p(fwd["means2D"]),
                                                   _p(fwd["conic_opacity"]), _p(feats), _p(fwd["final_T"]),
                                                   _p(fwd["n_contrib"]), _p(dpix), _p(acc), _p(fragile), _p(touched))
    g = {
        "dL_dmeans2D": np.zeros((P, 3), dt), "dL_dconic": np.zeros((P, 4), dt), "dL_dopacity": np.zeros((P, 1), dt),
        "dL_dcolors": np.zeros((P, 3), dt), "dL_dmeans3D": np.zeros((P, 3), dt), "dL_dcov3D": np.zeros((P, 6), dt),
        "dL_dsh": np.zeros((P, M, 3), dt), "dL_dscales": np.zeros((P, 3), dt), "dL_drotations": np.zeros((P, 4), dt),
    }
    g["dL_dmeans2D"][:, 0:2] = acc[:, 0:2]
    g["dL_dconic"][:, 0] = acc[:, 2]
    g["dL_dconic"][:, 1] = acc[:, 3]
    g["dL_dconic"][:, 3] = acc[:, 4]
    g["dL_dopacity"][:, 0] = acc[:, 5]
    g["dL_dcolors"][:] = acc[:, 6:9]
    cov = n["cov3D_precomp"] if n["cov3D_precomp"] is not None else fwd["cov3D"]
    getattr(L, "hgs_oracle_preprocess_backward" + sfx)(
        C.c_int(P), C.c_int(n["sh_degree"]), C.c_int(M), C.c_int(W), C.c_int(H), _p(n["means3D"]), _p(fwd["radii"]),
        _p(n["shs"]), _p(fwd["clamped"]), _p(n["scales"]), _p(n["rotations"]), cr(n["scale_modifier"]),
        _p(np.ascontiguousarray(cov, dtype=dt)), _p(n["viewmatrix"]), _p(n["projmatrix"]), cr(n["tanfovx"]),
        cr(n["tanfovy"]), _p(n["campos"]), _p(g["dL_dmeans2D"]), _p(g["dL_dconic"]), _p(g["dL_dmeans3D"]),
        _p(g["dL_dcolors"]), _p(g["dL_dcov3D"]), _p(g["dL_dsh"]), _p(g["dL_dscales"]), _p(g["dL_drotations"]))
    g["acc"] = acc
    g["fragile"] = fragile.astype(bool)
    g["touched"] = touched.astype(bool) | g["fragile"]
    return g


def mark_visible(means3D, viewmatrix):
    m = _arr(means3D, np.float32)
    v = _arr(viewmatrix, np.float32, (16,))
    out = np.zeros(m.shape[0], np.uint8)
    lib().hgs_oracle_mark_visible_f32(C.c_int(m.shape[0]), _p(m), _p(v), _p(out))
    return out.astype(bool)


def dist2(points, want_debug=False):
    """simple_knn distCUDA2 restatement: mean squared distance to the 3 nearest neighbours."""
    p = _arr(points, np.float32)
    P = p.shape[0]
    out = np.zeros(P, np.float32)
    codes = np.zeros(P, np.uint32)
    idx = np.zeros(P, np.uint32)
    lib().hgs_oracle_dist2(C.c_int(P), _p(p), _p(out), _p(codes), _p(idx))
    return (out, codes, idx) if want_debug else out


def filter_strand_list_segments(strands_list):
    """Flat-array restatement of c_utils.filter_strand_list_segments (object array in, [pairs,2,2] out)."""
    lens = np.array([int(s.shape[0]) for s in strands_list], np.int64)
    offsets = np.zeros(len(lens) + 1, np.int64)
    np.cumsum(lens, out=offsets[1:])
    rows = (np.concatenate([np.asarray(s, np.int64).reshape(-1, 2) for s in strands_list], 0)
            if len(lens) else np.zeros((0, 2), np.int64))
    return filter_strand_segments_flat(offsets, rows)


def filter_strand_segments_flat(offsets, rows):
    offsets = _arr(offsets, np.int64)
    rows = _arr(rows, np.int64)
    S = offsets.shape[0] - 1
    f = lib().hgs_oracle_filter_strand_segments
    total = f(C.c_int64(S), _p(offsets), _p(rows), None)
    out = np.empty((total, 2, 2), np.int64)
    f(C.c_int64(S), _p(offsets), _p(rows), _p(out))
    return out
