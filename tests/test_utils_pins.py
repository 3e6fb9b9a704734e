"""Product-side Python helpers pinned against the reference's own outputs (tests/golden/ref_python_pins.npz)."""
import numpy as np
import torch


def test_eval_sh_matches_reference(golden):
    from utils.sh import RGB2SH, SH2RGB, eval_sh
    feats = torch.from_numpy(golden["sh_feats"])
    dirs = torch.from_numpy(golden["sh_xyz"]) - torch.from_numpy(golden["sh_campos"])
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    sv = feats.transpose(1, 2).reshape(-1, 3, 16)
    for deg in range(4):
        got = eval_sh(deg, sv, dirs) + 0.5
        np.testing.assert_allclose(got.numpy(), golden[f"sh_raw_deg{deg}"], rtol=1e-5, atol=1e-6)
    c = torch.from_numpy(golden["rgb2sh_in"])
    np.testing.assert_allclose(RGB2SH(c).numpy(), golden["rgb2sh_out"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(SH2RGB(c).numpy(), golden["sh2rgb_out"], rtol=1e-6, atol=1e-7)


def test_graphics_matches_reference(golden):
    from utils.graphics import focal2fov, fov2focal, getProjectionMatrix, getWorld2View2
    for i in range(golden["cam_R"].shape[0]):
        R, T = golden["cam_R"][i], golden["cam_T"][i]
        W, H, f = int(golden["cam_W"][i]), int(golden["cam_H"][i]), float(golden["cam_focal"][i])
        fx, fy = focal2fov(f, W), focal2fov(f, H)
        assert fx == float(golden["cam_fovx"][i]) and fy == float(golden["cam_fovy"][i])
        assert abs(fov2focal(fx, W) - float(golden["cam_focal_back"][i])) < 1e-9
        wv = torch.tensor(getWorld2View2(R, T)).transpose(0, 1)
        np.testing.assert_array_equal(wv.numpy(), golden["cam_wv"][i])
        pr = getProjectionMatrix(0.01, 100.0, fx, fy).transpose(0, 1)
        np.testing.assert_array_equal(pr.numpy(), golden["cam_proj"][i])
        np.testing.assert_allclose((wv @ pr).numpy(), golden["cam_full"][i], rtol=1e-6, atol=1e-6)


def test_camera_class_matches_reference_recipe(golden):
    from scene.cameras import Camera
    i = 0
    cam = Camera(1, golden["cam_R"][i], golden["cam_T"][i], float(golden["cam_fovx"][i]), float(golden["cam_fovy"][i]),
                 None, None, "x", 0, data_device="cpu", image_width=int(golden["cam_W"][i]),
                 image_height=int(golden["cam_H"][i]))
    np.testing.assert_array_equal(cam.world_view_transform.numpy(), golden["cam_wv"][i])
    np.testing.assert_allclose(cam.full_proj_transform.numpy(), golden["cam_full"][i], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(cam.camera_center.numpy(), golden["cam_center"][i], rtol=1e-5, atol=1e-6)


def test_c_utils_matches_reference_cython(golden):
    import c_utils
    lens, rows = golden["strand_lens"], golden["strand_rows"]
    off = np.concatenate([[0], np.cumsum(lens)])
    sl = np.empty(len(lens), dtype=object)
    for j in range(len(lens)):
        sl[j] = rows[off[j]:off[j + 1]]
    out = c_utils.filter_strand_list_segments(sl)
    assert out.dtype == np.int64
    np.testing.assert_array_equal(out, golden["strand_pairs"])
    assert c_utils.filter_strand_list_segments(np.empty(0, dtype=object)).shape == (0, 2, 2)


def test_c_utils_native_module_contract():
    """The native half (c_utils/_c_utils.c): strided / Fortran-ordered strands, plain lists, the flat form, and the
    reference's error behaviour (None -> TypeError; a strand of >= 2 rows that is not 2-D int64 -> ValueError, which is what
    the reference's typed memoryview raises; shorter strands are never looked at)."""
    import pytest
    import c_utils
    assert c_utils._load().__name__ == "c_utils._c_utils"            # the compiled module, not a Python loop
    rng = np.random.default_rng(3)
    base = [rng.integers(0, 1000, (n, 2)).astype(np.int64) for n in (5, 0, 1, 7, 2)]
    want = c_utils.filter_strand_segments_flat_numpy(*c_utils.strands_to_flat(base))
    assert want.shape == (4 + 0 + 0 + 6 + 1, 2, 2)
    np.testing.assert_array_equal(c_utils.filter_strand_list_segments(base), want)                       # a plain list
    views = [np.asfortranarray(a) for a in base]
    views[3] = np.concatenate([base[3], base[3]], axis=1)[:, ::2][:, :2].copy()[:, :]                    # fresh copy ...
    wide = np.zeros((7, 6), np.int64)
    wide[:, ::3] = base[3]
    views[3] = wide[:, ::3]                                                                               # ... and a strided view
    np.testing.assert_array_equal(c_utils.filter_strand_list_segments(np.array(views + [None], dtype=object)[:-1]), want)
    np.testing.assert_array_equal(c_utils.filter_strand_segments_flat(*c_utils.strands_to_flat(base)), want)
    with pytest.raises(TypeError):
        c_utils.filter_strand_list_segments(None)
    with pytest.raises(ValueError):
        c_utils.filter_strand_list_segments([base[0].astype(np.int32)])
    with pytest.raises(ValueError):
        c_utils.filter_strand_list_segments([np.zeros((3, 2, 2), np.int64)])
    assert c_utils.filter_strand_list_segments([np.zeros((1, 2), np.int32)]).shape == (0, 2, 2)         # too short to matter
    with pytest.raises(ValueError):
        c_utils.filter_strand_segments_flat(np.array([0, 5], np.int64), np.zeros((3, 2), np.int64))     # offsets beyond rows


def test_argument_defaults_match_reference(golden):
    from argparse import ArgumentParser
    from arguments import GeneralParams, ModelParams, OptimizationParams
    p = ArgumentParser()
    op, mp, gp = OptimizationParams(p), ModelParams(p), GeneralParams(p)
    for k in golden.files:
        if k.startswith("opt_"):
            assert float(getattr(op, k[4:])) == float(golden[k]), k
    assert mp.sh_degree == int(golden["model_sh_degree"])
    a = p.parse_args(["-s", "/tmp/x", "--iterations", "7", "--eval"])
    assert a.source_path == "/tmp/x" and a.iterations == 7 and a.eval is True and a.densify_grad_threshold == 0.0002
    assert op.extract(a).iterations == 7 and gp.save_frequency == 5000
