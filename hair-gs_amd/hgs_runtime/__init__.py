"""Loader of libhgs.so (the hand-written HIP kernels behind a C ABI, include/hgs.h).

There is NO fallback: if the library is missing or fails to load, every op raises.  The library is built
in-tree by `hgs_runtime.build()` (hipcc --offload-arch=gfx950, cross-compiles without a GPU) and travels
with the source tree.
"""
import ctypes as C
import os
import subprocess

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("HGS_LIB") or os.path.join(_PKG_ROOT, "libhgs.so")   # HGS_LIB: another build of the same ABI (A/B runs, tools/)
CSRC = os.path.join(_PKG_ROOT, "csrc")
_lib = None

vp, ci, cf, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t

# name -> (restype, argtypes); mirrors include/hgs.h one to one (tests/test_abi.py checks the header against this)
SIGNATURES = {
    "hgs_abi_version": (ci, []),
    "hgs_last_error": (C.c_char_p, []),
    "hgs_geom_bytes": (sz, [ci]),
    "hgs_image_bytes": (sz, [ci, ci]),
    "hgs_binning_bytes": (sz, [ci]),
    "hgs_backward_scratch_bytes": (sz, [ci, ci]),
    "hgs_forward_preprocess": (ci, [vp, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, cf, vp, vp, vp, vp, vp, cf, cf, ci,
                                    vp, vp, vp, vp, vp]),
    "hgs_forward_render": (ci, [vp, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp]),
    "hgs_backward": (ci, [vp, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, cf, vp, vp, vp, vp, vp, cf, cf, vp, vp, vp,
                          vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "hgs_binning_bytes_multi": (sz, [ci]),
    "hgs_backward_scratch_bytes_multi": (sz, [ci, ci]),
    "hgs_forward_render_multi": (ci, [vp, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp]),
    "hgs_backward_multi": (ci, [vp, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, cf, vp, vp, vp, vp, vp, cf, cf, vp, vp, vp,
                                vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "hgs_param_backward_bytes": (sz, []),
    "hgs_backward_multi_params": (ci, [vp, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, cf, cf, vp, vp, vp, vp, vp, vp,
                                       vp, vp]),
    "hgs_hair_endpoint_gather": (ci, [vp, ci, vp, vp, vp, vp, vp]),
    "hgs_mark_visible": (ci, [vp, ci, vp, vp, vp, vp]),
    "hgs_dist2_scratch_bytes": (sz, [ci]),
    "hgs_dist2": (ci, [vp, ci, vp, vp, vp, sz]),
    "hgs_ssim_l1_scratch_floats": (sz, [ci, ci, ci]),
    "hgs_ssim_l1_num_blocks": (ci, [ci, ci, ci]),
    "hgs_ssim_l1_forward": (ci, [vp, ci, ci, ci, vp, vp, vp, vp, vp]),
    "hgs_ssim_l1_backward": (ci, [vp, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp]),
    "hgs_orientation_loss_num_blocks": (ci, [ci, ci]),
    "hgs_orientation_loss_forward": (ci, [vp, ci, ci, vp, vp, vp, cf, vp, vp, vp, vp]),
    "hgs_orientation_loss_backward": (ci, [vp, ci, ci, vp, vp, vp, cf, vp, vp, vp, vp, vp, vp]),
    "hgs_adam_step": (ci, [vp, ci, vp, vp, vp, vp, vp, vp, vp, cf, cf, cf, vp]),
    "hgs_smoothness_num_blocks": (ci, [ci]),
    "hgs_smoothness_forward": (ci, [vp, ci, vp, vp, cf, cf, vp]),
    "hgs_smoothness_backward": (ci, [vp, ci, ci, vp, vp, cf, cf, vp, vp, vp]),
    "hgs_strand_geometry_forward": (ci, [vp, ci, vp, vp, vp, cf, vp, vp, vp, vp]),
    "hgs_strand_geometry_backward": (ci, [vp, ci, ci, vp, vp, vp, cf, vp, vp, vp, vp, vp, vp]),
    "hgs_view_targets_bytes": (sz, []),
    "hgs_head_params_bytes": (sz, []),
    "hgs_strand_fusion_bytes": (sz, []),
    "hgs_select_view": (ci, [vp, vp, ci, vp, cf, vp]),
    "hgs_iteration_prologue": (ci, [vp, vp, ci, vp, cf, vp, vp, sz, vp]),
    "hgs_adam_prep_bytes": (sz, []),
    "hgs_adam_inline_bytes": (sz, []),
    "hgs_image_zero_range": (ci, [ci, ci, vp, vp]),
    "hgs_graph_find_prologue": (ci, [vp, vp]),
    "hgs_graph_find_prologues": (ci, [vp, ci, vp, vp, vp]),
    "hgs_graph_set_prologue": (ci, [vp, vp, vp, ci, vp, cf, vp, vp, sz]),
    "hgs_set_view_queue": (ci, [vp, vp, ci, vp, cf, vp]),
    "hgs_select_view_queued": (ci, [vp, vp, ci, vp, vp, vp, vp]),
    "hgs_hair_params_forward": (ci, [vp, ci, vp, vp, vp, cf, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "hgs_hair_params_backward": (ci, [vp, ci, ci, vp, vp, vp, cf, vp, vp, vp, vp, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp]),
    "hgs_cloud_params_forward": (ci, [vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "hgs_cloud_forward_preprocess": (ci, [vp, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                          cf, cf, ci, vp, vp, vp, vp, vp]),
    "hgs_hair_forward_preprocess": (ci, [vp, ci, ci, ci, ci, ci, vp, vp, vp, cf, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                         cf, cf, ci, vp, vp, vp, vp, vp]),
    "hgs_cloud_params_backward": (ci, [vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "hgs_loss_head_scratch_floats": (sz, [vp]),
    "hgs_loss_head_tail": (ci, [vp, vp, vp, vp]),
    "hgs_loss_head_forward": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "hgs_loss_head_backward": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, vp, vp, vp, vp]),
    "hgs_densify_stats": (ci, [vp, ci, vp, vp, ci, vp, vp, vp]),
    "hgs_radius_pairs": (ci, [vp, ci, vp, vp, cf, cf, ci, ci, vp, vp, vp, ci]),
    "hgs_knn3": (ci, [vp, ci, vp, vp, vp]),
    "hgs_nearest_distance_f64": (ci, [vp, ci, ci, vp, vp, vp]),
    "hgs_strand_walk_ends": (ci, [vp, ci, ci, vp, vp, vp, vp, vp, vp]),
    "hgs_strand_walk_fill": (ci, [vp, ci, vp, vp, vp, vp, vp, vp, vp]),
    "hgs_set_tile_cull": (ci, [ci]),
    "hgs_set_segment_policy": (ci, [ci, ci, ci]),
    "hgs_set_row_reduce": (ci, [ci]),
    "hgs_set_lazy_records": (ci, [ci]),
    "hgs_debug_set_wg_trace": (ci, [vp, vp]),
    "hgs_prof_enable": (ci, [ci]),
    "hgs_prof_bracket_overhead_ms": (C.c_double, []),
    "hgs_prof_collect": (ci, [vp, vp]),
    "hgs_prof_kernel_name": (C.c_char_p, [ci]),
    "hgs_geom_layout": (ci, [ci, vp]),
    "hgs_image_layout": (ci, [ci, ci, vp]),
    "hgs_binning_layout": (ci, [ci, vp]),
}
GEOM_FIELDS = ["depths", "clamped", "means2D", "cov3D", "conic_opacity", "rgb", "tiles_touched", "point_offsets", "rect",
               "block_sums"]
IMG_FIELDS = ["final_T", "n_contrib", "ranges", "tile_count", "tile_cursor", "tile_maxc", "status", "tile_order"]
BIN_FIELDS = ["keys", "point_list", "packed", "inv", "keys_sorted"]
PACKED_FLOATS = 12
INST_GRAD_FLOATS = 12


class HgsError(RuntimeError):
    pass


class ViewTargets(C.Structure):
    """include/hgs.h HgsViewTargets (184 bytes)."""
    _fields_ = [("image", vp), ("float_mask", vp), ("orientation", vp), ("confidence", vp), ("mask", vp),
                ("viewmatrix", cf * 16), ("projmatrix", cf * 16), ("campos", cf * 3), ("mask_count", cf)]


class HeadParams(C.Structure):
    """include/hgs.h HgsHeadParams."""
    _fields_ = [("H", ci), ("W", ci), ("lambda_dssim", cf), ("lambda_mask", cf), ("lambda_orientation", cf),
                ("lambda_smooth", cf), ("bg", cf * 3), ("min_val", cf), ("window", cf * 11), ("n_smooth", ci),
                ("cos_threshold", cf), ("eps", cf), ("n_endpoints", ci), ("defer_tail", ci), ("tile_used", vp),
                ("tiles_x", ci), ("tiles_y", ci)]


class HeadTail(C.Structure):
    """include/hgs.h HgsHeadTail."""
    _fields_ = [("pix_partials", vp), ("nb_pix", ci), ("out", vp), ("inv_hw", cf), ("l_mask", cf), ("l_ori", cf),
                ("l_smooth", cf), ("bce", ci), ("ori", ci), ("smooth", ci)]


class Prologue(C.Structure):
    """include/hgs.h HgsPrologue."""
    _fields_ = [("table", vp), ("view", ci), ("slot", vp), ("lr", cf), ("lr_dst", vp), ("zero_ptr", vp), ("zero_bytes", sz),
                ("adam_prep", vp)]


ADAM_MAX_TENSORS = 8


class AdamSlot(C.Structure):
    """include/hgs.h HgsAdamSlot."""
    _fields_ = [("p", vp), ("m", vp), ("v", vp), ("coef", vp)]


class AdamPrep(C.Structure):
    """include/hgs.h HgsAdamPrep (lives in device memory: built here, uploaded as bytes)."""
    _fields_ = [("n", ci), ("lr", vp * ADAM_MAX_TENSORS), ("step", vp * ADAM_MAX_TENSORS), ("beta1", cf), ("beta2", cf), ("coef", vp)]


class AdamInline(C.Structure):
    """include/hgs.h HgsAdamInline."""
    _fields_ = [("slot", AdamSlot * 6), ("beta1", cf), ("beta2", cf), ("eps", cf)]


class StrandFusion(C.Structure):
    """include/hgs.h HgsStrandFusion."""
    _fields_ = [("smooth_pairs", vp), ("n_smooth", ci), ("cos_threshold", cf), ("eps", cf), ("smooth_partials", vp),
                ("smooth_pair_grads", vp), ("head_out", vp), ("grad_out", vp), ("radii", vp), ("dmean2D", vp), ("dmean2D_stride", ci),
                ("max_radii2D", vp), ("grad_accum", vp), ("denom", vp), ("ep_segments", vp), ("ep_pairs", vp),
                ("n_endpoints", ci), ("head_tail", HeadTail), ("prologue", Prologue)]


class ParamBackward(C.Structure):
    """include/hgs.h HgsParamBackward."""
    _fields_ = [("kind", ci), ("endpoints", vp), ("endpoint_pairs", vp), ("dist_to_scale_factor", cf), ("seg_contrib", vp),
                ("d_width", vp), ("rotation_raw", vp), ("d_means3D", vp), ("d_scaling_raw", vp), ("d_rotation_raw", vp),
                ("extra4", vp), ("d_opacity_raw", vp), ("d_mask_raw", vp), ("dL_dmeans2D_rgb", vp), ("max_radii2D", vp),
                ("grad_accum", vp), ("denom", vp), ("head_tail", HeadTail), ("adam", AdamInline)]


PARAMS_HAIR, PARAMS_CLOUD = 1, 2
HEAD_SKIP_PIXELS, HEAD_SKIP_SMOOTH = 1, 2
VIEW_QUEUE_MAX = 16   # include/hgs.h HGS_VIEW_QUEUE_MAX
HEAD_OUT = ["total", "l1", "dssim", "mask", "orientation", "smooth", "ori_count", "smooth_count", "g_ssim", "g_l1", "g_mask",
            "g_ori", "g_smooth", "total_fwd"]
HEAD_NOUT = 16
FUSED_PREPROCESS_MAX_TILES = 8192   # include/hgs.h HGS_FUSED_PREPROCESS_MAX_TILES
ABI_VERSION = 6   # include/hgs.h HGS_ABI_VERSION: bumped whenever a struct, a signature or a buffer layout changes


def build(verbose=False):
    """Compile csrc/*.hip into libhgs.so for gfx950 (no GPU needed)."""
    subprocess.check_call(["make", "-s", "-j8", "-C", CSRC], stdout=None if verbose else subprocess.DEVNULL)
    import c_utils
    c_utils.build()   # the host-side native helper (CPython extension, gcc)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HgsError(f"{LIB_PATH} not found: run hgs_runtime.build() (hipcc --offload-arch=gfx950). "
                           "There is no CPU fallback.")
        # torch first: its wheel carries its own libamdhip64; the process must end up with ONE HIP runtime, and it has to
        # be the one that owns the tensors' memory.  Loading libhgs.so before torch binds it to /opt/rocm's copy, and a
        # later kernel launch on torch's memory then fails with "no ROCm-capable device is detected".
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        # a stale library called through newer struct layouts corrupts memory: refuse it (version AND struct sizes)
        if L.hgs_abi_version() != ABI_VERSION:
            raise HgsError(f"{LIB_PATH}: ABI version {L.hgs_abi_version()}, this binding needs {ABI_VERSION}: "
                           "rebuild with hgs_runtime.build()")
        for fn, st in (("hgs_view_targets_bytes", ViewTargets), ("hgs_head_params_bytes", HeadParams),
                       ("hgs_strand_fusion_bytes", StrandFusion), ("hgs_param_backward_bytes", ParamBackward),
                       ("hgs_adam_prep_bytes", AdamPrep), ("hgs_adam_inline_bytes", AdamInline)):
            if getattr(L, fn)() != C.sizeof(st):
                raise HgsError(f"{LIB_PATH}: {fn}() = {getattr(L, fn)()} but the binding's struct has {C.sizeof(st)} bytes: "
                               "rebuild with hgs_runtime.build()")
        if os.environ.get("HGS_LAZY_RECORDS") in ("0", "1"):      # A/B aid: pins include/hgs.h hgs_set_lazy_records for the process
            L.hgs_set_lazy_records(int(os.environ["HGS_LAZY_RECORDS"]))
        _lib = L
    return _lib


def check(status):
    if status != 0:
        raise HgsError(lib().hgs_last_error().decode())


def ptr(t):
    """Device pointer of a tensor; None / empty tensors map to NULL (the reference passes 0-element tensors)."""
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


def current_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def require_gpu_tensor(t, name, dtype=None):
    import torch
    if not t.is_cuda:
        raise HgsError(f"{name} must be a CUDA(HIP) tensor: libhgs.so only runs on the GPU, there is no CPU path")
    if dtype is not None and t.dtype != dtype:
        raise HgsError(f"{name} must be {dtype}, got {t.dtype}")
    return t.contiguous()


KERNEL_COUNT = 18


def prof_enable(on=True):
    lib().hgs_prof_enable(int(bool(on)))


def prof_collect():
    """{kernel name: (total_ms, launches)} since the last collect (synchronises the recorded events)."""
    ms = (C.c_double * KERNEL_COUNT)()
    n = (C.c_longlong * KERNEL_COUNT)()
    check(lib().hgs_prof_collect(ms, n))
    return {lib().hgs_prof_kernel_name(i).decode(): (ms[i], n[i]) for i in range(KERNEL_COUNT)}


def layout(kind, *dims):
    n = {"geom": len(GEOM_FIELDS), "image": len(IMG_FIELDS), "binning": len(BIN_FIELDS)}[kind]
    arr = (C.c_size_t * n)()
    getattr(lib(), f"hgs_{kind}_layout")(*dims, arr)
    names = {"geom": GEOM_FIELDS, "image": IMG_FIELDS, "binning": BIN_FIELDS}[kind]
    return dict(zip(names, list(arr)))
