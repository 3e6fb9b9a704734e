"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU and exports exactly what
include/hgs.h declares; the Python surface mirrors the reference's names and validation."""
import os
import re

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "hgs.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(hgs_[a-z0-9_]+)\s*\(", src))


def test_library_exports_every_declared_symbol():
    import hgs_runtime as rt
    rt.build()
    L = rt.lib()
    declared = _header_functions()
    assert declared == set(rt.SIGNATURES), declared ^ set(rt.SIGNATURES)
    for name in declared:
        assert hasattr(L, name), name
    # the version the header declares, the library reports and the binding demands are one number
    header = open(os.path.join(ROOT, "include", "hgs.h")).read()
    declared_version = int(re.search(r"#define\s+HGS_ABI_VERSION\s+(\d+)", header).group(1))
    assert L.hgs_abi_version() == declared_version == rt.ABI_VERSION


def test_workspace_sizes_and_layouts_are_consistent():
    import hgs_runtime as rt
    L = rt.lib()
    for P in (0, 1, 255, 256, 257, 100000):
        n = L.hgs_geom_bytes(P)
        lay = rt.layout("geom", P)
        assert all(v % 256 == 0 for v in lay.values()) and max(lay.values()) < n
    for (W, H) in ((1, 1), (16, 16), (17, 33), (1920, 1080)):
        n = L.hgs_image_bytes(W, H)
        lay = rt.layout("image", W, H)
        T = ((W + 15) // 16) * ((H + 15) // 16)
        assert lay["n_contrib"] - lay["final_T"] >= 4 * W * H
        slots = 64
        while slots < T:
            slots *= 2                       # tile_count / tile_cursor: a power-of-two table of scattered slots
        # (between them: the T + 1 row-run marks of HGS_COUNT_ROW_RUNS passes, an even number of words)
        assert lay["tile_cursor"] - lay["tile_count"] == 4 * (slots + ((T + 2) & ~1)) and lay["status"] + 16 <= n
    for R in (0, 1, 1000, 441042):
        n = L.hgs_binning_bytes(R)
        lay = rt.layout("binning", R)
        assert lay["packed"] - lay["point_list"] >= 4 * R and max(lay.values()) + 8 * R <= n
    assert L.hgs_backward_scratch_bytes(10, 1000) >= 1000 * 48


def test_struct_mirrors_match_the_library():
    import ctypes as C
    import hgs_runtime as rt
    L = rt.lib()
    assert L.hgs_view_targets_bytes() == C.sizeof(rt.ViewTargets) == 184
    assert L.hgs_head_params_bytes() == C.sizeof(rt.HeadParams)
    assert L.hgs_strand_fusion_bytes() == C.sizeof(rt.StrandFusion)
    assert rt.ViewTargets.viewmatrix.offset == 40 and rt.ViewTargets.campos.offset == 168
    src = open(os.path.join(ROOT, "include", "hgs.h")).read()
    names = re.search(r"enum \{ HGS_HEAD_TOTAL = 0,(.*?)HGS_HEAD_NOUT = (\d+)", src, re.S)
    listed = [n.strip() for n in ("HGS_HEAD_TOTAL," + names.group(1)).split(",") if n.strip()]
    assert [n[len("HGS_HEAD_"):].lower() for n in listed] == [
        {"orientation": "orientation"}.get(k, k) for k in rt.HEAD_OUT]
    assert int(names.group(2)) == rt.HEAD_NOUT


def test_python_surface_matches_reference_names():
    import inspect
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import _C
    assert dgr.GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "sh_degree", "campos", "prefiltered", "debug")
    sig = inspect.signature(dgr.GaussianRasterizer.forward)
    assert list(sig.parameters) == ["self", "means3D", "means2D", "opacities", "shs", "colors_precomp", "scales",
                                    "rotations", "cov3D_precomp"]
    assert len(inspect.signature(_C.rasterize_gaussians).parameters) == 19
    assert len(inspect.signature(_C.rasterize_gaussians_backward).parameters) == 21
    from simple_knn._C import distCUDA2  # noqa: F401


def test_argument_validation_matches_reference():
    import torch
    import diff_gaussian_rasterization as dgr
    rs = dgr.GaussianRasterizationSettings(8, 8, 1.0, 1.0, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 0,
                                           torch.zeros(3), False, False)
    r = dgr.GaussianRasterizer(rs)
    x = torch.zeros(4, 3)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(x, x, torch.zeros(4, 1), scales=x, rotations=torch.zeros(4, 4))
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(x, x, torch.zeros(4, 1), shs=torch.zeros(4, 1, 3), colors_precomp=x, scales=x, rotations=torch.zeros(4, 4))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(x, x, torch.zeros(4, 1), colors_precomp=x, scales=x)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(x, x, torch.zeros(4, 1), colors_precomp=x, scales=x, rotations=torch.zeros(4, 4), cov3D_precomp=torch.zeros(4, 6))


def test_no_cpu_fallback():
    """CPU tensors must be rejected loudly: the product has no CPU path."""
    import torch
    import hgs_runtime as rt
    from simple_knn._C import distCUDA2
    with pytest.raises(rt.HgsError):
        distCUDA2(torch.zeros(8, 3))
    import diff_gaussian_rasterization as dgr
    rs = dgr.GaussianRasterizationSettings(8, 8, 1.0, 1.0, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 0,
                                           torch.zeros(3), False, False)
    x = torch.zeros(4, 3)
    with pytest.raises(rt.HgsError):
        dgr.GaussianRasterizer(rs)(x, x, torch.zeros(4, 1), colors_precomp=x, scales=x, rotations=torch.zeros(4, 4))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "hair-gs_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "hgs_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f
