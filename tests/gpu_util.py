"""Helpers for the GPU parity tests: run the HIP path through the drop-in `_C` module (ctypes -> C ABI) and view
the opaque workspace buffers as typed arrays for comparison with the oracle."""
import numpy as np
import torch

import hgs_runtime as rt
from diff_gaussian_rasterization import _C


def to_dev(x, dev="cuda"):
    if x is None:
        return torch.empty(0, device=dev)
    return torch.as_tensor(np.ascontiguousarray(x), device=dev)


def run_forward(scene, dev="cuda", debug=False):
    s = scene
    args = (to_dev(s["bg"]), to_dev(s["means3D"]), to_dev(s["colors_precomp"]), to_dev(s["opacities"]).reshape(-1, 1),
            to_dev(s["scales"]), to_dev(s["rotations"]), float(s["scale_modifier"]), to_dev(s["cov3D_precomp"]),
            to_dev(s["viewmatrix"]), to_dev(s["projmatrix"]), float(s["tanfovx"]), float(s["tanfovy"]), int(s["H"]),
            int(s["W"]), to_dev(s["shs"]), int(s["sh_degree"]), to_dev(s["campos"]), False, debug)
    R, color, radii, geom, binning, img = _C.rasterize_gaussians(*args)
    torch.cuda.synchronize()
    return dict(args=args, R=R, color=color, radii=radii, geom=geom, binning=binning, img=img)


def _view(buf, off, count, dtype):
    nbytes = count * np.dtype(dtype).itemsize
    return np.frombuffer(buf[off:off + nbytes].cpu().numpy().tobytes(), dtype=dtype, count=count)


def intermediates(scene, fw):
    P, W, H, R = scene["means3D"].shape[0], scene["W"], scene["H"], fw["R"]
    T = ((W + 15) // 16) * ((H + 15) // 16)
    g = rt.layout("geom", P)
    im = rt.layout("image", W, H)
    o = {}
    gb, ib, bb = fw["geom"], fw["img"], fw["binning"]
    if P:
        o["depths"] = _view(gb, g["depths"], P, np.float32)
        o["clamped"] = _view(gb, g["clamped"], 3 * P, np.uint8).reshape(P, 3)
        o["means2D"] = _view(gb, g["means2D"], 2 * P, np.float32).reshape(P, 2)
        o["cov3D"] = _view(gb, g["cov3D"], 6 * P, np.float32).reshape(P, 6)
        o["conic_opacity"] = _view(gb, g["conic_opacity"], 4 * P, np.float32).reshape(P, 4)
        o["rgb"] = _view(gb, g["rgb"], 3 * P, np.float32).reshape(P, 3)
        o["tiles_touched"] = _view(gb, g["tiles_touched"], P, np.uint32)
        o["point_offsets"] = _view(gb, g["point_offsets"], P, np.uint32)
    o["final_T"] = _view(ib, im["final_T"], W * H, np.float32).reshape(H, W)
    o["n_contrib"] = _view(ib, im["n_contrib"], W * H, np.uint32).reshape(H, W)
    o["ranges"] = _view(ib, im["ranges"], 2 * T, np.uint32).reshape(T, 2)
    o["status"] = _view(ib, im["status"], 16, np.uint32)   # csrc/hgs_common.h HGS_ST_*: [0] R, [1] overflow, [4] sort chunk items, [5] blend segment items, [6] segment length, [7] blend work items, [8] wait timeout
    o["tile_maxc"] = _view(ib, im["tile_maxc"], T, np.uint32)
    if R:
        b = rt.layout("binning", R)
        o["point_list"] = _view(bb, b["point_list"], R, np.uint32)
        ks = _view(bb, b["keys_sorted"], R, np.uint64)  # depth_bits<<32 | id, per tile
        # rebuild the reference's key format tile<<32|depth_bits (rasterizer_impl.cu:102-104) from ranges
        tile_of = np.zeros(R, np.uint64)
        for t in range(T):
            a, e = o["ranges"][t]
            tile_of[a:e] = t
        o["keys_sorted"] = (tile_of << np.uint64(32)) | (ks >> np.uint64(32))
        o["inv"] = _view(bb, b["inv"], R, np.uint32)
    else:
        o["point_list"] = np.zeros(0, np.uint32)
        o["keys_sorted"] = np.zeros(0, np.uint64)
    o["out_color"] = fw["color"].cpu().numpy()
    o["radii"] = fw["radii"].cpu().numpy()
    o["num_rendered"] = R
    return o


def in_tile_order(o, n):
    """Capacity mode (round 5) ALLOCATES the tiles' segments instead of scanning them in tile order: the binning buffer holds
    them in the order the allocating wavefronts' atomics arrived.  Returns (ranges, point_list, keys_sorted) of an
    intermediates() result re-laid in tile order -- what the blocking mode, i.e. the reference's layout, holds -- after checking
    that the segments partition [0, n) exactly."""
    r = o["ranges"].astype(np.int64)
    lens = r[:, 1] - r[:, 0]
    used = np.flatnonzero(lens > 0)
    assert np.all(r[lens == 0] == 0), "an empty tile's range is (0, 0)"
    order = used[np.argsort(r[used, 0], kind="stable")]
    assert int(lens.sum()) == n
    if len(order):                                          # back to back, from 0, without overlap
        assert r[order[0], 0] == 0 and np.array_equal(r[order[1:], 0], r[order[:-1], 1]) and r[order[-1], 1] == n
    start = np.concatenate([[0], np.cumsum(lens)[:-1]])
    ranges = np.stack([start, start + lens], 1)
    ranges[lens == 0] = 0
    idx = np.concatenate([np.arange(r[t, 0], r[t, 1]) for t in used]) if len(used) else np.zeros(0, np.int64)
    return ranges.astype(np.uint32), o["point_list"][idx], o["keys_sorted"][idx]


def run_backward(scene, fw, dL_dpix):
    a = fw["args"]
    (bg, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D, view, proj, tfx, tfy, H, W, sh, degree,
     campos, _, debug) = a
    out = _C.rasterize_gaussians_backward(bg, means3D, fw["radii"], colors, scales, rotations, scale_modifier, cov3D, view,
                                          proj, tfx, tfy, to_dev(dL_dpix), sh, degree, campos, fw["geom"], fw["R"],
                                          fw["binning"], fw["img"], debug)
    torch.cuda.synchronize()
    names = ["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales",
             "dL_drotations"]
    g = {n: t.cpu().numpy() for n, t in zip(names, out)}
    g["dL_dconic"] = _C.rasterize_gaussians_backward.last_dL_dconic.cpu().numpy().reshape(-1, 4)
    return g



def blend_work_list(scene, fw):
    """The blend kernels' work list (HgsImage.tile_order: tile | segment << 24; T + max(T, 1024) entries, sort_tiles_kernel)."""
    W, H = scene["W"], scene["H"]
    T = ((W + 15) // 16) * ((H + 15) // 16)
    lay = rt.layout("image", W, H)
    return _view(fw["img"], lay["tile_order"], T + max(T, 1024), np.uint32)


def free_port():
    """A TCP port nobody listens on right now (rendezvous of the multi-process tests: fixed numbers collide with whatever else
    runs on the box, or with the previous test's socket)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]

