"""Pins the oracle (and the synthetic camera helper) against outputs of the reference's own Python
code stored in tests/golden/ref_python_pins.npz (generator: tests/golden/make_ref_python_pins.py)."""
import numpy as np

from oracle import hgs_oracle as O
from tests import scenes


def _sh_scene(golden, deg):
    xyz, campos, feats = golden["sh_xyz"], golden["sh_campos"], golden["sh_feats"]
    P = xyz.shape[0]
    # camera far away looking at the cloud so every Gaussian survives culling; SH uses campos only
    cam = scenes.make_camera(eye=(0, 0, -40.0), target=(0, 0, 0), W=64, H=64, fovx_deg=40.0)
    s = dict(cam)
    s.update(means3D=xyz, opacities=np.full(P, 0.5, np.float32), bg=np.zeros(3, np.float32), sh_degree=deg,
             scale_modifier=1.0, shs=feats, colors_precomp=None, scales=np.full((P, 3), 0.05, np.float32),
             rotations=np.tile(np.array([1, 0, 0, 0], np.float32), (P, 1)), cov3D_precomp=None)
    s["campos"] = campos  # the SH direction uses campos as given (CR/forward.cu:26)
    return s


def test_sh_colors_match_reference_eval_sh(golden):
    for deg in range(4):
        s = _sh_scene(golden, deg)
        f = O.forward(s, render=False)
        vis = f["radii"] > 0
        assert vis.all()
        ref = golden[f"sh_rgb_deg{deg}"]
        np.testing.assert_allclose(f["rgb"], ref, rtol=2e-5, atol=2e-6)
        raw = golden[f"sh_raw_deg{deg}"]
        safe = np.abs(raw) > 1e-5
        assert ((f["clamped"] > 0) == (raw < 0))[safe].all()


def test_camera_matrices_match_reference_graphics(golden):
    n = golden["cam_R"].shape[0]
    for i in range(n):
        R, T = golden["cam_R"][i], golden["cam_T"][i]
        wv = scenes.world2view(R, T).T
        np.testing.assert_allclose(wv, golden["cam_wv"][i], rtol=0, atol=1e-6)
        pr = scenes.projection(0.01, 100.0, float(golden["cam_fovx"][i]), float(golden["cam_fovy"][i])).T
        np.testing.assert_allclose(pr, golden["cam_proj"][i], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose((wv @ pr), golden["cam_full"][i], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(np.linalg.inv(wv)[3, :3], golden["cam_center"][i], rtol=1e-4, atol=1e-5)


def test_strand_filter_matches_reference_cython(golden):
    lens, rows = golden["strand_lens"], golden["strand_rows"]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    out = O.filter_strand_segments_flat(off, rows)
    assert out.dtype == np.int64
    np.testing.assert_array_equal(out, golden["strand_pairs"])
    strands = [rows[off[j]:off[j + 1]] for j in range(len(lens))]
    np.testing.assert_array_equal(O.filter_strand_list_segments(strands), golden["strand_pairs"])
    assert O.filter_strand_segments_flat(np.zeros(1, np.int64), np.zeros((0, 2), np.int64)).shape == (0, 2, 2)
    assert golden["strand_pairs_empty"].shape == (0, 2, 2)
