"""The loss kernels and the strand kernels AT THE NORTH-STAR SIZE against the reference's own loss/losses.py and model, run on
the CPU of the authoring container at 1920 x 1080 / 100 k segments (tests/golden/ref_fullsize_pins.npz, generator
tests/golden/make_ref_fullsize_pins.py).  Inputs are re-created from the generator's seeds; the fixture holds the scalars and
4096 sampled elements of every gradient."""
import importlib.util
import os
import types

import numpy as np
import pytest
import torch

from arguments import OptimizationParams

_HERE = os.path.dirname(os.path.abspath(__file__))
PINS = np.load(os.path.join(_HERE, "golden", "ref_fullsize_pins.npz"))
_spec = importlib.util.spec_from_file_location("make_ref_fullsize_pins", os.path.join(_HERE, "golden", "make_ref_fullsize_pins.py"))
GEN = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(GEN)          # (only its seeded input builders are used: images(), head_inputs())
H, W = GEN.H, GEN.W


def _samples(got, key, rel, what):
    idx, want = PINS[key + "_idx"], PINS[key + "_val"].astype(np.float64)
    g = got.detach().reshape(-1)[torch.from_numpy(idx).to(got.device)].double().cpu().numpy()
    scale = float(PINS[key + "_abs_max"]) if (key + "_abs_max") in PINS.files else float(np.abs(want).max())
    err = float(np.abs(g - want).max())
    assert err <= rel * scale, (what, err, scale)


def test_torch_ssim_at_1080p_reproduces_the_reference_run():
    """The op-by-op statements (the checker of the fused kernels) at full size, hair-like pair."""
    from loss import losses as Ls
    a, b = GEN.images(int(PINS["ssim1_seed"]), True)
    x, y = torch.from_numpy(a).requires_grad_(True), torch.from_numpy(b)
    s = Ls.ssim(x, y)
    g, = torch.autograd.grad(s, x)
    assert abs(float(s) - float(PINS["ssim1_ssim"])) <= 1e-7
    _samples(g, "ssim1_d_ssim", 1e-6, "d ssim / d image")
    assert abs(float(g.double().abs().sum()) - float(PINS["ssim1_d_ssim_abs_sum"])) <= 1e-6 * float(PINS["ssim1_d_ssim_abs_sum"])


@pytest.mark.gpu
@pytest.mark.parametrize("ci", [0, 1])
def test_hip_ssim_l1_at_1080p_against_the_reference_run(ci):
    """hgs_ssim_l1_* at 1920 x 1080 vs loss/losses.py:16-17,43-84 as executed: SSIM to 2e-5 absolute, L1 to 1e-6 relative,
    4096 sampled gradient elements to 2e-4 of the largest, the sum of |gradient| to 1e-4 relative."""
    from hgs_runtime.fused import ssim_l1
    k = f"ssim{ci}_"
    a, b = GEN.images(int(PINS[k + "seed"]), bool(PINS[k + "hair"]))
    x, y = torch.from_numpy(a).cuda().requires_grad_(True), torch.from_numpy(b).cuda()
    s, l = ssim_l1(x, y)
    gs, = torch.autograd.grad(s, x, retain_graph=True)
    gl, = torch.autograd.grad(l, x)
    assert abs(float(s) - float(PINS[k + "ssim"])) <= 2e-5
    assert abs(float(l) - float(PINS[k + "l1"])) <= 1e-6 * float(PINS[k + "l1"])
    _samples(gs, k + "d_ssim", 2e-4, "d ssim / d image")
    _samples(gl, k + "d_l1", 1e-6, "d l1 / d image")
    assert abs(float(gs.double().abs().sum()) - float(PINS[k + "d_ssim_abs_sum"])) <= 1e-4 * float(PINS[k + "d_ssim_abs_sum"])


def _north_star_model(device):
    from synthetic import build_workload
    model, _, _ = build_workload("north_star", device=device, seed=0, with_targets=False, n_views=1)
    rng = np.random.default_rng(5)
    ep = model._endpoints.detach().cpu().numpy()
    ep = (ep + rng.normal(0, 2e-4, ep.shape)).astype(np.float32)
    with torch.no_grad():
        model._endpoints.copy_(torch.from_numpy(ep).to(device))
    opt = OptimizationParams()
    opt.spatial_sort = False
    model.training_setup(opt)
    model.compute_strands_info()
    assert model.endpoint_pairs.shape[0] == int(PINS["model_segments"])
    return model, opt


@pytest.mark.gpu
def test_hip_strand_kernels_at_north_star_size_against_the_reference_run():
    """hgs_strand_geometry_* and hgs_smoothness_* on 100 k segments vs the reference's getters and angle_smoothness_loss."""
    from loss import losses as Ls
    m, _ = _north_star_model("cuda")
    xyz, scaling, _, orientation = m.derived_gaussians()
    _samples(xyz, "get_xyz", 1e-6, "xyz")
    _samples(scaling, "get_scaling", 2e-6, "scaling")
    _samples(orientation, "get_orientation", 2e-6, "orientation")
    for th in (30, 3):
        v = Ls.angle_smoothness_loss(m, threshold=float(th))
        want = float(PINS[f"smooth{th}_value"])
        assert abs(float(v) - want) <= 2e-5 * want, th
        g, = torch.autograd.grad(v, m._endpoints)
        _samples(g, f"smooth{th}", 2e-4, f"smoothness gradient, threshold {th}")
        assert abs(float(g.double().abs().sum()) - float(PINS[f"smooth{th}_abs_sum"])) <= 1e-4 * float(PINS[f"smooth{th}_abs_sum"])


@pytest.mark.gpu
def test_hip_loss_head_at_1080p_against_the_reference_run():
    """hgs_loss_head_forward / _backward at 1920 x 1080 vs loss_function (loss/losses.py:319-355) as executed: the five terms,
    the total, sampled elements of the three image gradients."""
    import ctypes as C
    import hgs_runtime as rt
    from hgs_runtime.strand_step import head_params
    from loss import losses as Ls
    dev = torch.device("cuda")
    a, b = GEN.images(13, True)
    omap, mlog, ori, conf, mask, wvt = GEN.head_inputs(14)
    m, opt = _north_star_model("cuda")
    image, gt = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    omap_t, mask_img = torch.from_numpy(omap).to(dev), torch.from_numpy(mlog[0].copy()).to(dev)
    ori_t, conf_t = torch.from_numpy(ori).to(dev), torch.from_numpy(conf).to(dev)
    m8 = torch.from_numpy(mask.astype(np.uint8)).to(dev)
    fmask = m8.float()
    row = rt.ViewTargets()
    row.image, row.orientation, row.confidence, row.float_mask, row.mask = (t.data_ptr() for t in (gt, ori_t, conf_t, fmask, m8))
    eye = np.eye(4, dtype=np.float32).reshape(-1)
    for j in range(16):
        row.viewmatrix[j] = float(wvt.reshape(-1)[j])
        row.projmatrix[j] = float(eye[j])
    row.mask_count = float(m8.sum().item())
    targets = torch.from_numpy(np.frombuffer(bytes(row), dtype=np.uint8).copy()).to(dev)
    hp = head_params(H, W, opt, 0, 0, float(m.min_val), True)
    L = rt.lib()
    scratch = torch.empty(L.hgs_loss_head_scratch_floats(C.byref(hp)), device=dev)
    out = torch.zeros(rt.HEAD_NOUT, device=dev)
    d_img, d_mask, d_omap = torch.empty(3, H, W, device=dev), torch.empty(H, W, device=dev), torch.empty(3, H, W, device=dev)
    one = torch.ones(1, device=dev)
    rt.check(L.hgs_loss_head_forward(rt.current_stream(), C.byref(hp), image.data_ptr(), mask_img.data_ptr(), omap_t.data_ptr(),
                                     targets.data_ptr(), None, None, scratch.data_ptr(), out.data_ptr(), None, None))
    rt.check(L.hgs_loss_head_backward(rt.current_stream(), C.byref(hp), image.data_ptr(), mask_img.data_ptr(), omap_t.data_ptr(),
                                      targets.data_ptr(), None, None, scratch.data_ptr(), out.data_ptr(), one.data_ptr(), 0,
                                      d_img.data_ptr(), d_mask.data_ptr(), d_omap.data_ptr(), None))
    o = dict(zip(rt.HEAD_OUT, out.tolist()))
    assert abs(o["l1"] - float(PINS["head_term_l1"])) <= 1e-6 * float(PINS["head_term_l1"])
    assert abs(o["dssim"] - float(PINS["head_term_dssim"])) <= 2e-5
    assert abs(o["mask"] - float(PINS["head_term_mask"])) <= 1e-5 * float(PINS["head_term_mask"])
    assert abs(o["orientation"] - float(PINS["head_term_orientation"])) <= 1e-5 * float(PINS["head_term_orientation"])
    smooth = Ls.angle_smoothness_loss(m)
    assert abs(float(smooth) - float(PINS["head_term_smooth"])) <= 2e-5 * float(PINS["head_term_smooth"])
    want_total = float(PINS["head_total"]) - opt.lambda_smooth * float(PINS["head_term_smooth"])     # (the head was given no smoothness pairs)
    assert abs(o["total"] - want_total) <= 2e-5 * want_total
    empty = torch.from_numpy(~omap.any(axis=0)).to(dev)
    d_omap[:, empty] = 0.0        # (masked pixels the render left at 0: the generator zeroes the reference's ~1e10 there too)
    _samples(d_img, "head_d_image", 2e-4, "d total / d image")
    _samples(d_omap, "head_d_omap", 2e-4, "d total / d direction image")
    _samples(d_mask, "head_d_mask", 1e-5, "d total / d mask channel")
