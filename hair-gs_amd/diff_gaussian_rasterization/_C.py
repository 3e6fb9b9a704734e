"""Drop-in for the reference's pybind11 module `diff_gaussian_rasterization._C`
(submodules/diff-gaussian-rasterization/ext.cpp:15-19): same three functions, same argument order, same
return tuples -- implemented over the C ABI of libhgs.so (include/hgs.h) with ctypes."""
import ctypes as C

import torch

import hgs_runtime as rt

# ---- optional asynchronous mode -----------------------------------------------------------------------------------
# Default (async_mode False) = the reference's behaviour: every forward blocks once on `num_rendered`
# (cuda_rasterizer/rasterizer_impl.cu:280-281) to size the binning buffer exactly.
# With set_async(True) the forward never blocks: the binning buffer is sized from a CAPACITY (slack x the largest
# instance count seen so far), the true count and the overflow flag of every pass are copied to pinned host memory
# asynchronously, and the caller validates them ONCE per training step with check_async() (one synchronisation
# instead of three).  If a pass needed more than its capacity, check_async() raises HgsCapacityOverflow after growing
# the capacity: the caller discards the step's gradients and repeats it.  `num_rendered` returned by
# rasterize_gaussians is then the capacity (it only sizes/carves buffers downstream).
IMAGE_PREZEROED = 2   # include/hgs.h HGS_IMAGE_PREZEROED (flag in `prefiltered`)
COUNT_ROW_RUNS = 4    # include/hgs.h HGS_COUNT_ROW_RUNS
_state = {"last_R": 0, "last_counts_clean": False, "async": False, "cap": 0, "slack": 1.5, "dirty": False, "cap_used": None, "max_R": {}, "cull": None}


class HgsCapacityOverflow(RuntimeError):
    pass


def bucket_capacity(n):
    """A binning capacity of at least `n` instances with four significant bits (m x 2^e, 8 <= m < 16: at most 12.5 % above n).
    The workspaces carved for a capacity are the largest allocations of a pass (~330 B per instance); a model that grows by a few
    per cent per topology event would otherwise ask for a slightly larger block at every re-capture, which no cached block can
    serve -- the caching allocator then holds one retired block per event (tools/dev/recapture_memory.py).  In buckets, successive
    captures ask for the same sizes and take the blocks the dropped graph has freed."""
    n = int(n)
    if n <= 16:
        return max(n, 0)
    e = n.bit_length() - 4
    return ((n + (1 << e) - 1) >> e) << e


def set_tile_cull(enabled=True):
    """Tile culling (include/hgs.h hgs_set_tile_cull: drop the (Gaussian, tile) instances no pixel can blend; image,
    radii and every gradient are bit-identical with and without) for ALL entry points of this module: True / False, or
    None for the per-entry-point defaults:
      rasterize_gaussians         OFF -- the reference's own function: `num_rendered`, the tile lists in the returned
                                  buffers and `n_contrib` are the reference's, entry for entry;
      rasterize_gaussians_culled  (what diff_gaussian_rasterization.GaussianRasterizer, i.e. render(), calls) and
      rasterize_gaussians_multi   ON -- callers that only see the image, the radii and the gradients.
    Returns the previous setting."""
    was = _state["cull"]
    _state["cull"] = None if enabled is None else bool(enabled)
    return was


def set_row_reduce(mode):
    """How the single-pass backward sums the instance rows per Gaussian (include/hgs.h hgs_set_row_reduce): True / False, or
    None = decide per call -- from the exact instance count of a blocking-mode pass (R >= 4 P), from the library's capacity rule
    otherwise.  train.GraphedStep.capture() sets it from the instance counts its warm-up passes measured, so that the form
    follows the MODEL (and with it every replay and every eager iteration until the next capture), not the capacity a run
    happens to hold: a run that rolls back and raises its capacity keeps the arithmetic of one that never overflowed."""
    was = _state.get("row_reduce")
    _state["row_reduce"] = None if mode is None else bool(mode)
    return was


def _apply_row_reduce(L, P, R):
    import os
    env = os.environ.get("HGS_ROW_REDUCE")          # A/B aid: 0 / 1 pins the form for the whole process
    if env in ("0", "1"):
        L.hgs_set_row_reduce(int(env))
        return
    mode = _state.get("row_reduce")
    if mode is None:
        mode = (1 if R >= 4 * P else 0) if not _state["async"] else -1
    L.hgs_set_row_reduce(int(mode))


def set_row_runs(mode):
    """How the preprocess launch counts tile rectangles of more than 16 tiles (include/hgs.h HGS_COUNT_ROW_RUNS): True = by tile
    rows plus a one-workgroup launch, False = tile by tile, None = decide per pass from the instances per Gaussian seen so far
    (capacity mode: the capacity; blocking mode: the previous pass's exact count) -- the counts are the same integers either way."""
    was = _state.get("row_runs")
    _state["row_runs"] = None if mode is None else bool(mode)
    return was


def _row_runs_flag(P, use_async):
    import os
    env = os.environ.get("HGS_ROW_RUNS")            # A/B aid: 0 / 1 pins the form for the whole process
    mode = _state.get("row_runs")
    if env in ("0", "1"):
        mode = env == "1"
    if mode is None:
        mode = (_state["cap"] if use_async else _state.get("last_exact_R", 0)) >= 8 * P
    return COUNT_ROW_RUNS if (mode and P > 0) else 0


def set_async(enabled=True, slack=1.5):
    """Capacity mode: passes never wait for num_rendered; the library keeps a sticky device-side maximum of it
    (hgs_forward_preprocess max_rendered) which check_async() reads -- one synchronisation per check, no per-pass copy."""
    _state["async"], _state["slack"] = bool(enabled), float(slack)
    _state["dirty"] = False
    _state["cap_used"] = None


def _max_rendered(dev):
    t = _state["max_R"].get(dev)
    if t is None:
        t = torch.zeros(1, dtype=torch.int32, device=dev)
        _state["max_R"][dev] = t
    return t


def check_async():
    """Synchronise once and validate every pass issued since the last check against the capacity it ran with; returns
    [largest num_rendered seen] ([] if no pass was issued).  Raises HgsCapacityOverflow after raising the capacity."""
    if not _state.get("dirty"):
        return []
    worst, cap = 0, _state["cap_used"]
    for t in _state["max_R"].values():
        worst = max(worst, int(t.item()) & 0xFFFFFFFF)   # (.item() synchronises the stream the passes ran on; the word is unsigned)
        t.zero_()
    _state["dirty"] = False
    _state["cap_used"] = None
    _state["last_exact_R"] = worst        # (an instance COUNT, not a capacity: what set_row_reduce's callers decide by)
    if worst == 0xFFFFFFFF:   # include/hgs.h HGS_WAIT_TIMED_OUT
        raise rt.HgsError("a raster pass gave up an inter-workgroup wait (status word 8): its frame is invalid")
    _state["cap"] = max(_state["cap"], bucket_capacity(int(worst * _state["slack"]) + 4096))
    if cap is not None and worst > cap:
        raise HgsCapacityOverflow(f"a raster pass needed {worst} instances (capacity {cap}): capacity raised to {_state['cap']}, repeat the step")
    return [worst]


def _f32(t, name):
    if t is None or t.numel() == 0:
        return None
    return rt.require_gpu_tensor(t, name, torch.float32)


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                        prefiltered, debug):
    """RasterizeGaussiansCUDA (rasterize_points.cu:35-115).
    Returns (num_rendered, out_color[3,H,W], radii[P], geomBuffer, binningBuffer, imgBuffer); the tile lists are the
    reference's (see set_tile_cull)."""
    return _forward(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                    projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered, debug,
                    None, False)


def rasterize_gaussians_culled(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                               viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                               prefiltered, debug):
    """rasterize_gaussians with tile culling on (this module's extension; what GaussianRasterizer / render() call): same
    image, radii and gradients bit for bit, fewer instances in the buffers."""
    return _forward(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                    projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered, debug,
                    None, True)


def rasterize_gaussians_multi(background7, means3D, colors, extra4, opacity, scales, rotations, scale_modifier,
                              cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh,
                              degree, campos, prefiltered, debug, image_buffer=None, hair=None):
    """Single-pass 7-channel forward (hgs_forward_render_multi): RGB + `extra4` [P,4] unclamped channels blended with
    the same weights.  Returns (num_rendered, out_color[7,H,W], radii, geomBuffer, binningBuffer, imgBuffer).
    image_buffer: a uint8 tensor of hgs_image_bytes(W, H) whose counters the caller has cleared on this stream
    (hgs_iteration_prologue): used as the imgBuffer, and the pass skips its own clearing launch."""
    return _forward(background7, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                    projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered, debug,
                    extra4, True, image_buffer, None, hair)


class HairSource:
    """The strand parameters behind means3D / scales / rotations / opacity (/ extra4) of a pass: those five tensors are then
    OUTPUTS, written by the pass's first launch.  In capacity mode that launch is hgs_hair_forward_preprocess (parameters ->
    Gaussians -> preprocess in one kernel, `fusion`'s riders beside it); otherwise hgs_hair_params_forward runs in front of
    the ordinary preprocess launch.  `fusion`: hgs_runtime.StrandFusion (or None); `fill(fused)` is called once the form is
    known and must put the iteration prologue (if one rides) into `fusion` with the matching zero range."""

    kind = "hair"

    def __init__(self, endpoints, pairs, width, factor, opacity_raw, mask_raw, fusion=None, fill=None):
        self.endpoints, self.pairs, self.width, self.factor = endpoints, pairs, width, float(factor)
        self.opacity_raw, self.mask_raw, self.fusion, self.fill = opacity_raw, mask_raw, fusion, fill


class CloudSource:
    """The Stage-I counterpart of HairSource: the raw scaling / rotation / opacity / mask parameters behind scales / rotations /
    opacity / extra4 of a pass (means3D is the model's own parameter): hgs_cloud_forward_preprocess in capacity mode,
    hgs_cloud_params_forward in front of the ordinary preprocess launch otherwise."""
    kind = "cloud"

    def __init__(self, scaling_raw, rotation_raw, opacity_raw, mask_raw, fusion=None, fill=None):
        self.scaling_raw, self.rotation_raw = scaling_raw, rotation_raw
        self.opacity_raw, self.mask_raw, self.fusion, self.fill = opacity_raw, mask_raw, fusion, fill


def will_fuse_hair(W, H):
    """Would a pass with a HairSource at this size run the one-launch form now?  (capacity mode with a learnt capacity,
    at most HGS_FUSED_PREPROCESS_MAX_TILES tiles)"""
    import os
    return (_state["async"] and _state["cap"] > 0 and os.environ.get("HGS_FUSE_PREPROCESS", "1") != "0"
            and ((int(W) + 15) // 16) * ((int(H) + 15) // 16) <= rt.FUSED_PREPROCESS_MAX_TILES)


def rasterize_gaussians_prezeroed(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                                  viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                                  image_buffer, max_rendered=None, hair=None):
    """rasterize_gaussians_culled on an image buffer whose counters the caller has cleared on this stream (see
    rasterize_gaussians_multi); max_rendered: an int32[1] device tensor of the caller's that receives the sticky maximum of
    num_rendered in capacity mode instead of this module's (gaussian_renderer.frames validates its own frames with it)."""
    return _forward(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                    projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, False, False,
                    None, True, image_buffer, max_rendered, hair)


def _forward(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
             projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered, debug, extra4,
             cull, image_buffer=None, max_rendered=None, hair=None):
    if means3D.ndim != 2 or means3D.shape[1] != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")  # rasterize_points.cu:57-59
    L = rt.lib()
    L.hgs_set_tile_cull(int(_state["cull"] if _state["cull"] is not None else bool(cull)))   # host-side switch, read by this pass
    means3D = rt.require_gpu_tensor(means3D, "means3D", torch.float32)
    dev = means3D.device
    P, H, W = means3D.shape[0], int(image_height), int(image_width)
    M = sh.shape[1] if (sh is not None and sh.numel() != 0) else 0
    n_ch = 3 if extra4 is None else 7
    extra_ = None if extra4 is None else rt.require_gpu_tensor(extra4, "extra4", torch.float32)
    out_color = torch.empty((n_ch, H, W), dtype=torch.float32, device=dev)
    radii = torch.empty((P,), dtype=torch.int32, device=dev)
    u8 = dict(dtype=torch.uint8, device=dev)
    geom = torch.empty((L.hgs_geom_bytes(P),), **u8)
    flags = int(bool(prefiltered))
    if image_buffer is not None:
        if image_buffer.numel() != L.hgs_image_bytes(W, H) or image_buffer.dtype != torch.uint8 or image_buffer.device != dev:
            raise RuntimeError("image_buffer: need a uint8 tensor of hgs_image_bytes(W, H) on the inputs' device")
        img, flags = image_buffer, flags | IMAGE_PREZEROED
    else:
        img = torch.empty((L.hgs_image_bytes(W, H),), **u8)
    bg = _f32(background, "bg")
    colors_, opacity_, scales_, rots_, cov_, sh_ = (_f32(colors, "colors_precomp"), _f32(opacity, "opacities"),
                                                     _f32(scales, "scales"), _f32(rotations, "rotations"),
                                                     _f32(cov3D_precomp, "cov3D_precomp"), _f32(sh, "sh"))
    view, proj, cam = _f32(viewmatrix, "viewmatrix"), _f32(projmatrix, "projmatrix"), _f32(campos, "campos")
    stream = rt.current_stream()
    with torch.cuda.device(dev):
        use_async = _state["async"] and _state["cap"] > 0 and P > 0
        flags |= _row_runs_flag(P, use_async)
        n_host = C.c_int(0)
        fused_hair = hair is not None and use_async and will_fuse_hair(W, H)
        if hair is not None:
            if scale_modifier != 1.0 or colors_ is not None or cov_ is not None or sh_ is None:
                raise RuntimeError("a HairSource pass renders SH colours at scale_modifier 1")
            if hair.fill is not None:
                hair.fill(fused_hair)
            o_raw, m_raw = _f32(hair.opacity_raw, "opacity_raw"), _f32(hair.mask_raw, "mask_raw")
            fu_ = None if hair.fusion is None else C.byref(hair.fusion)
            ex_out = extra_ if extra_ is not None else torch.empty((P, 4), dtype=torch.float32, device=dev)
            mr_ = rt.ptr(max_rendered if max_rendered is not None else _max_rendered(dev)) if fused_hair else None
            if hair.kind == "hair":
                ep_, w_ = _f32(hair.endpoints, "endpoints"), _f32(hair.width, "width")
                pairs_ = rt.require_gpu_tensor(hair.pairs, "endpoint_pairs", torch.int64)
                if fused_hair:
                    rt.check(L.hgs_hair_forward_preprocess(stream, P, int(degree), M, W, H, rt.ptr(ep_), rt.ptr(pairs_), rt.ptr(w_),
                                                           hair.factor, rt.ptr(o_raw), rt.ptr(m_raw), rt.ptr(sh_), rt.ptr(means3D),
                                                           rt.ptr(scales_), rt.ptr(rots_), rt.ptr(opacity_), rt.ptr(ex_out),
                                                           rt.ptr(view), rt.ptr(proj), rt.ptr(cam), float(tan_fovx),
                                                           float(tan_fovy), flags, rt.ptr(geom), rt.ptr(img), rt.ptr(radii), mr_, fu_))
                else:
                    rt.check(L.hgs_hair_params_forward(stream, P, rt.ptr(ep_), rt.ptr(pairs_), rt.ptr(w_), hair.factor, rt.ptr(o_raw),
                                                       rt.ptr(m_raw), rt.ptr(means3D), rt.ptr(scales_), rt.ptr(rots_), None,
                                                       rt.ptr(opacity_), rt.ptr(ex_out), fu_))
            else:
                s_raw, r_raw = _f32(hair.scaling_raw, "scaling_raw"), _f32(hair.rotation_raw, "rotation_raw")
                if fused_hair:
                    rt.check(L.hgs_cloud_forward_preprocess(stream, P, int(degree), M, W, H, rt.ptr(means3D), rt.ptr(s_raw),
                                                            rt.ptr(r_raw), rt.ptr(o_raw), rt.ptr(m_raw), rt.ptr(sh_),
                                                            rt.ptr(scales_), rt.ptr(rots_), rt.ptr(opacity_), rt.ptr(ex_out),
                                                            rt.ptr(view), rt.ptr(proj), rt.ptr(cam), float(tan_fovx),
                                                            float(tan_fovy), flags, rt.ptr(geom), rt.ptr(img), rt.ptr(radii), mr_, fu_))
                else:
                    rt.check(L.hgs_cloud_params_forward(stream, P, rt.ptr(s_raw), rt.ptr(r_raw), rt.ptr(o_raw), rt.ptr(m_raw),
                                                        rt.ptr(scales_), rt.ptr(rots_), rt.ptr(opacity_), rt.ptr(ex_out), fu_))
        if not fused_hair:
            rt.check(L.hgs_forward_preprocess(stream, P, int(degree), M, W, H, rt.ptr(means3D), rt.ptr(sh_), rt.ptr(colors_),
                                              rt.ptr(opacity_), rt.ptr(scales_), float(scale_modifier), rt.ptr(rots_),
                                              rt.ptr(cov_), rt.ptr(view), rt.ptr(proj), rt.ptr(cam), float(tan_fovx),
                                              float(tan_fovy), flags, rt.ptr(geom), rt.ptr(img),
                                              rt.ptr(radii), None if use_async else C.addressof(n_host),
                                              rt.ptr(max_rendered if max_rendered is not None else _max_rendered(dev))
                                              if use_async else None))
        # the scan of a capacity-mode pass (scatter kernel) leaves the per-tile instance counters of `img` at zero; a blocking
        # pass leaves its counts there (hgs_runtime.strand_step.ViewTable.counts_clean follows this)
        if P > 0:
            _state["last_counts_clean"] = bool(use_async and ((W + 15) // 16) * ((H + 15) // 16) <= rt.FUSED_PREPROCESS_MAX_TILES)
        if use_async:
            R = _state["cap"]
        else:
            R = int(n_host.value)
            _state["last_exact_R"] = R
            if _state["async"]:  # first call: learn the scale of the scene with one blocking read
                _state["cap"] = max(_state["cap"], bucket_capacity(int(R * _state["slack"]) + 4096))
        if extra_ is None:
            binning = torch.empty((L.hgs_binning_bytes(R),), **u8)
            rt.check(L.hgs_forward_render(stream, P, W, H, R, rt.ptr(bg), rt.ptr(colors_), rt.ptr(geom), rt.ptr(binning),
                                          rt.ptr(img), rt.ptr(out_color)))
        else:
            binning = torch.empty((L.hgs_binning_bytes_multi(R),), **u8)
            rt.check(L.hgs_forward_render_multi(stream, P, W, H, R, rt.ptr(bg), rt.ptr(colors_), rt.ptr(extra_),
                                                rt.ptr(geom), rt.ptr(binning), rt.ptr(img), rt.ptr(out_color)))
        if use_async and max_rendered is None:
            _state["dirty"] = True
            _state["cap_used"] = R if _state["cap_used"] is None else min(_state["cap_used"], R)
        if debug:
            torch.cuda.synchronize(dev)  # surface asynchronous faults here, like CHECK_CUDA (auxiliary.h:166-173)
    _state["last_R"] = R
    return R, out_color, radii, geom, binning, img


def _scratch(nbytes, dev):
    """Backward scratch (per-instance partial-gradient rows).  It is deliberately NOT cleared by the library; with
    HGS_POISON_SCRATCH=1 (tests) it is filled with NaN so that any read of a row nobody wrote shows up."""
    import os
    buf = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    if os.environ.get("HGS_POISON_SCRATCH") == "1" and nbytes >= 4:
        buf[:nbytes // 4 * 4].view(torch.float32).fill_(float("nan"))
    return buf


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                 viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, sh, degree, campos,
                                 geomBuffer, R, binningBuffer, imageBuffer, debug):
    """RasterizeGaussiansBackwardCUDA (rasterize_points.cu:117-196).
    Returns (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)."""
    L = rt.lib()
    means3D = rt.require_gpu_tensor(means3D, "means3D", torch.float32)
    dev = means3D.device
    P = means3D.shape[0]
    H, W = int(dL_dout_color.shape[1]), int(dL_dout_color.shape[2])
    M = sh.shape[1] if (sh is not None and sh.numel() != 0) else 0
    f32 = dict(dtype=torch.float32, device=dev)
    new = torch.empty if P > 0 else torch.zeros  # the kernel writes every element when P > 0
    dL_dmeans3D, dL_dmeans2D, dL_dcolors = new((P, 3), **f32), new((P, 3), **f32), new((P, 3), **f32)
    dL_dconic, dL_dopacity, dL_dcov3D = new((P, 2, 2), **f32), new((P, 1), **f32), new((P, 6), **f32)
    dL_dsh, dL_dscales, dL_drotations = new((P, M, 3), **f32), new((P, 3), **f32), new((P, 4), **f32)
    if P == 0:
        return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations
    scratch = _scratch(L.hgs_backward_scratch_bytes(P, int(R)), dev)
    # keep every (possibly freshly made contiguous) input alive in a local until the launch has been enqueued:
    # a temporary released early would hand its block to the next temporary of the same size
    dpix = rt.require_gpu_tensor(dL_dout_color, "dL_dout_color", torch.float32)
    bg_, sh_, colors_, scales_, rots_, cov_ = (_f32(background, "bg"), _f32(sh, "sh"), _f32(colors, "colors_precomp"),
                                               _f32(scales, "scales"), _f32(rotations, "rotations"),
                                               _f32(cov3D_precomp, "cov3D_precomp"))
    view_, proj_, cam_ = _f32(viewmatrix, "viewmatrix"), _f32(projmatrix, "projmatrix"), _f32(campos, "campos")
    radii_ = rt.require_gpu_tensor(radii, "radii", torch.int32)
    with torch.cuda.device(dev):
        rt.check(L.hgs_backward(rt.current_stream(), P, int(degree), M, int(R), W, H, rt.ptr(bg_), rt.ptr(means3D),
                                rt.ptr(sh_), rt.ptr(colors_), rt.ptr(scales_), float(scale_modifier), rt.ptr(rots_),
                                rt.ptr(cov_), rt.ptr(view_), rt.ptr(proj_), rt.ptr(cam_), float(tan_fovx), float(tan_fovy),
                                rt.ptr(radii_), rt.ptr(geomBuffer), rt.ptr(binningBuffer), rt.ptr(imageBuffer),
                                rt.ptr(dpix), rt.ptr(scratch), rt.ptr(dL_dmeans2D), rt.ptr(dL_dconic), rt.ptr(dL_dopacity),
                                rt.ptr(dL_dcolors), rt.ptr(dL_dmeans3D), rt.ptr(dL_dcov3D), rt.ptr(dL_dsh),
                                rt.ptr(dL_dscales), rt.ptr(dL_drotations)))
        if debug:
            torch.cuda.synchronize(dev)
    rasterize_gaussians_backward.last_dL_dconic = dL_dconic  # kept for the parity tests
    return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations


def rasterize_gaussians_multi_backward(background7, means3D, radii, colors, scales, rotations, scale_modifier,
                                       cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, grad_planes, sh,
                                       degree, campos, geomBuffer, R, binningBuffer, imageBuffer, debug):
    """Backward of the single-pass mode (hgs_backward_multi).  `grad_planes`: list of 7 contiguous [H,W] tensors (views
    into larger gradient tensors are fine).  `background7=None` declares an all-zero background (include/hgs.h: selects the
    black-background specialisation of the blend backward; same gradients).  Returns (dL_dmeans2D_rgb, dL_dcolors, dL_dextra4, dL_dopacity,
    dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)."""
    L = rt.lib()
    means3D = rt.require_gpu_tensor(means3D, "means3D", torch.float32)
    dev, P = means3D.device, means3D.shape[0]
    H, W = int(grad_planes[0].shape[-2]), int(grad_planes[0].shape[-1])
    M = sh.shape[1] if (sh is not None and sh.numel() != 0) else 0
    f32 = dict(dtype=torch.float32, device=dev)
    new = torch.empty if P > 0 else torch.zeros
    dL_dmeans3D, dL_dmeans2D, dL_dcolors = new((P, 3), **f32), new((P, 3), **f32), new((P, 3), **f32)
    dL_dconic, dL_dopacity, dL_dcov3D = new((P, 2, 2), **f32), new((P, 1), **f32), new((P, 6), **f32)
    dL_dsh, dL_dscales, dL_drotations = new((P, M, 3), **f32), new((P, 3), **f32), new((P, 4), **f32)
    dL_dextra = new((P, 4), **f32)
    if P == 0:
        return dL_dmeans2D, dL_dcolors, dL_dextra, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations
    scratch = _scratch(L.hgs_backward_scratch_bytes_multi(P, int(R)), dev)
    planes = [rt.require_gpu_tensor(g, "grad plane", torch.float32) for g in grad_planes]
    plane_ptrs = (C.c_void_p * 7)(*[g.data_ptr() for g in planes])
    bg_, sh_, colors_, scales_, rots_, cov_ = (_f32(background7, "bg"), _f32(sh, "sh"), _f32(colors, "colors_precomp"),
                                               _f32(scales, "scales"), _f32(rotations, "rotations"),
                                               _f32(cov3D_precomp, "cov3D_precomp"))
    view_, proj_, cam_ = _f32(viewmatrix, "viewmatrix"), _f32(projmatrix, "projmatrix"), _f32(campos, "campos")
    radii_ = rt.require_gpu_tensor(radii, "radii", torch.int32)
    _apply_row_reduce(L, P, int(R))
    with torch.cuda.device(dev):
        rt.check(L.hgs_backward_multi(rt.current_stream(), P, int(degree), M, int(R), W, H, rt.ptr(bg_), rt.ptr(means3D),
                                      rt.ptr(sh_), rt.ptr(colors_), rt.ptr(scales_), float(scale_modifier), rt.ptr(rots_),
                                      rt.ptr(cov_), rt.ptr(view_), rt.ptr(proj_), rt.ptr(cam_), float(tan_fovx),
                                      float(tan_fovy), rt.ptr(radii_), rt.ptr(geomBuffer), rt.ptr(binningBuffer),
                                      rt.ptr(imageBuffer), plane_ptrs, rt.ptr(scratch), rt.ptr(dL_dextra),
                                      rt.ptr(dL_dmeans2D), rt.ptr(dL_dconic), rt.ptr(dL_dopacity), rt.ptr(dL_dcolors),
                                      rt.ptr(dL_dmeans3D), rt.ptr(dL_dcov3D), rt.ptr(dL_dsh), rt.ptr(dL_dscales),
                                      rt.ptr(dL_drotations)))
        if debug:
            torch.cuda.synchronize(dev)
    return dL_dmeans2D, dL_dcolors, dL_dextra, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations


def rasterize_gaussians_multi_backward_params(background7, means3D, radii, scales, rotations, viewmatrix, projmatrix, tan_fovx,
                                              tan_fovy, grad_planes, sh, degree, campos, geomBuffer, R, binningBuffer,
                                              imageBuffer, params):
    """hgs_backward_multi_params: the single-pass backward whose per-Gaussian launch also applies the parameters' backward
    (`params`: a filled hgs_runtime.ParamBackward; its output tensors are the caller's).  Returns dL_dsh [P,M,3]."""
    L = rt.lib()
    means3D = rt.require_gpu_tensor(means3D, "means3D", torch.float32)
    dev, P = means3D.device, means3D.shape[0]
    H, W = int(grad_planes[0].shape[-2]), int(grad_planes[0].shape[-1])
    M = sh.shape[1]
    dL_dsh = (torch.empty if P > 0 else torch.zeros)((P, M, 3), dtype=torch.float32, device=dev)
    scratch = _scratch(L.hgs_backward_scratch_bytes_multi(P, int(R)), dev)
    planes = [rt.require_gpu_tensor(g, "grad plane", torch.float32) for g in grad_planes]
    plane_ptrs = (C.c_void_p * 7)(*[g.data_ptr() for g in planes])
    bg_, sh_, scales_, rots_ = _f32(background7, "bg"), _f32(sh, "sh"), _f32(scales, "scales"), _f32(rotations, "rotations")
    view_, proj_, cam_ = _f32(viewmatrix, "viewmatrix"), _f32(projmatrix, "projmatrix"), _f32(campos, "campos")
    radii_ = rt.require_gpu_tensor(radii, "radii", torch.int32)
    _apply_row_reduce(L, P, int(R))
    with torch.cuda.device(dev):
        rt.check(L.hgs_backward_multi_params(rt.current_stream(), P, int(degree), M, int(R), W, H, rt.ptr(bg_), rt.ptr(means3D),
                                             rt.ptr(sh_), rt.ptr(scales_), rt.ptr(rots_), rt.ptr(view_), rt.ptr(proj_), rt.ptr(cam_),
                                             float(tan_fovx), float(tan_fovy), rt.ptr(radii_), rt.ptr(geomBuffer),
                                             rt.ptr(binningBuffer), rt.ptr(imageBuffer), plane_ptrs, rt.ptr(scratch),
                                             rt.ptr(dL_dsh), C.byref(params)))
    return dL_dsh


def mark_visible(means3D, viewmatrix, projmatrix):
    """markVisible (rasterize_points.cu:198-217) -> bool[P]."""
    means3D = rt.require_gpu_tensor(means3D, "means3D", torch.float32)
    P = means3D.shape[0]
    present = torch.zeros((P,), dtype=torch.bool, device=means3D.device)
    if P:
        view_, proj_ = _f32(viewmatrix, "viewmatrix"), _f32(projmatrix, "projmatrix")
        with torch.cuda.device(means3D.device):
            rt.check(rt.lib().hgs_mark_visible(rt.current_stream(), P, rt.ptr(means3D), rt.ptr(view_), rt.ptr(proj_),
                                               rt.ptr(present)))
    return present
