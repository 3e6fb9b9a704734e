"""The loss terms, the strand model's getters and the schedules against what the REFERENCE's own code produced on the CPU of
the authoring container (tests/golden/ref_loss_pins.npz, generator tests/golden/make_ref_loss_pins.py: loss/losses.py and
scene/hair_gaussian_model.py executed unedited, values and autograd gradients).

  * CPU (`-m "not gpu"`): this package's op-by-op torch statements -- the checker the GPU tests of the fused kernels use -- must
    reproduce the reference run: the checker is pinned.
  * GPU (`-m gpu`): the HIP kernels (hgs_ssim_l1_*, hgs_loss_head_*, hgs_strand_geometry_*, hgs_smoothness_*) against the same
    vectors directly, through the C ABI.
Tolerances are written where they apply; fp32 throughout.
"""
import os
import types

import numpy as np
import pytest
import torch

from arguments import OptimizationParams

_HERE = os.path.dirname(os.path.abspath(__file__))
PINS = np.load(os.path.join(_HERE, "golden", "ref_loss_pins.npz"))
TOPO = np.load(os.path.join(_HERE, "golden", "ref_topology_pins.npz"))
SSIM_CASES = [tuple(int(v) for v in r) for r in PINS["meta_ssim_cases"]]
MODEL_SEEDS = [int(s) for s in PINS["meta_smooth_seeds"]]
HEAD_CASES = [tuple(int(v) for v in r) for r in PINS["meta_head_cases"]]


def _t(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def _model(seed, endpoints, device):
    """This package's HairGaussianModel holding state `seed` of ref_topology_pins.npz with the given endpoints -- built the way
    tests/golden/make_ref_loss_pins.py builds the reference's."""
    from scene.hair_gaussian_model import HairGaussianModel
    k = f"s{seed}_"
    m = HairGaussianModel(sh_degree=3, device=device)
    m.ref_strand_root = TOPO[k + "ref_strand_root"]
    m.strand_root_endpoint_idx = _t(TOPO[k + "root_idx"], device)
    m.endpoint_pairs = _t(TOPO[k + "pairs"], device)
    P = lambda a: torch.nn.Parameter(_t(a, device).clone().requires_grad_(True))
    m._endpoints, m._features_dc, m._features_rest = P(endpoints), P(TOPO[k + "f_dc"]), P(TOPO[k + "f_rest"])
    m._opacity, m._mask, m._width = P(TOPO[k + "opacity"]), P(TOPO[k + "mask"]), P(TOPO[k + "width"])
    opt = OptimizationParams()
    opt.spatial_sort = False                      # keep the fixture's storage order (gradients are compared row by row)
    m.training_setup(opt)
    m.compute_strands_info()
    return m


def _close(got, want, rel, what, floor=0.0, where=None):
    got = got.detach().cpu().double().numpy() if torch.is_tensor(got) else np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    ok = np.isfinite(want)        # (the reference's own NaNs -- smoothness gradient next to a zero-length segment -- are not a target)
    if where is not None:
        ok = ok & np.broadcast_to(where, want.shape)
    scale = max(float(np.abs(want[ok]).max()) if ok.any() else 0.0, floor)
    err = float(np.abs(got[ok] - want[ok]).max()) if ok.any() else 0.0
    assert err <= rel * scale, (what, err, scale)


# ---- CPU: the op-by-op statements (the GPU tests' checker) reproduce the reference run --------------------------------------
@pytest.mark.parametrize("ci", range(len(SSIM_CASES)))
def test_torch_ssim_and_l1_reproduce_the_reference_run(ci):
    from loss import losses as Ls
    k = f"ssim{ci}_"
    x = _t(PINS[k + "img"], "cpu").requires_grad_(True)
    y = _t(PINS[k + "gt"], "cpu")
    s = Ls.ssim(x, y)
    gs, = torch.autograd.grad(s, x)
    l = Ls.l1_loss(x, y)
    gl, = torch.autograd.grad(l, x)
    # same operations on the same CPU: equal to the last bits of a float
    assert abs(float(s) - float(PINS[k + "ssim"])) <= 1e-7 and abs(float(l) - float(PINS[k + "l1"])) <= 1e-8
    _close(gs, PINS[k + "d_ssim"], 1e-6, "d ssim / d image")
    _close(gl, PINS[k + "d_l1"], 1e-7, "d l1 / d image")


@pytest.mark.parametrize("seed", MODEL_SEEDS)
def test_getters_smoothness_and_schedules_reproduce_the_reference_run(seed):
    from loss import losses as Ls
    k = f"model{seed}_"
    m = _model(seed, PINS[k + "endpoints"], "cpu")
    assert abs(float(m.dist_to_scale_factor) - float(PINS[k + "dist_to_scale_factor"])) <= 1e-7      # set_pval, gaussian_model.py:696-704
    vals = {"scaling": m.get_scaling, "xyz": m.get_xyz, "orientation": m.get_orientation, "opacity": m.get_opacity, "mask": m.get_mask}
    for n, v in vals.items():
        assert np.array_equal(v.detach().numpy(), PINS[k + n]), n                                     # same torch statements: bit-equal
    total = sum((vals[n] * _t(PINS[k + "up_" + n], "cpu")).sum() for n in vals)
    grads = torch.autograd.grad(total, [m._endpoints, m._width, m._opacity, m._mask])
    for n, g in zip(("endpoints", "width", "opacity", "mask"), grads):
        _close(g, PINS[k + "d_" + n], 1e-6, "d / d " + n)
    Ls.fused_losses = False
    try:
        for th in (30, 3, 179):
            v = Ls.angle_smoothness_loss(m, threshold=float(th))
            want = float(PINS[k + f"smooth{th}_value"])
            assert abs(float(v) - want) <= 1e-6 * max(want, 1.0), th
            if torch.is_tensor(v) and v.requires_grad:
                g, = torch.autograd.grad(v, m._endpoints)
                _close(g, PINS[k + f"smooth{th}_d_endpoints"], 1e-5, f"smoothness gradient, threshold {th}", floor=1e-12)
            else:
                assert not PINS[k + f"smooth{th}_d_endpoints"].any()
    finally:
        Ls.fused_losses = True
    for it, want in zip(PINS["meta_lr_iterations"], PINS[k + "schedules"]):
        m.update_learning_rate(int(it))
        lr = [g["lr"] for g in m.optimizer.param_groups if g["name"] == "endpoints"][0]
        assert np.allclose([lr, m.merge_dist_th, m.merge_angle_th], want, rtol=1e-12, atol=0), int(it)


def _close_direction_gradient(got, k, rel, rel_empty):
    """d total / d direction image in two parts: pixels the render left at exactly (0, 0, 0) inside the orientation mask get
    ~confidence / (count min_val^2) from the reference's statements (x / (|x| + min_val) and atan2(0, min_val): 1e10-1e12 here)
    and would hide every other pixel in a comparison relative to the largest element."""
    empty = ~PINS[k + "omap"].any(axis=0)
    want = PINS[k + "d_omap"]
    _close(got, want, rel, "d total / d direction image (blended pixels)", where=~empty[None])
    if (want[:, empty] != 0).any():
        _close(got, want, rel_empty, "d total / d direction image (empty masked pixels)", where=empty[None])


def _head_inputs(ci, device):
    H, W, with_mask = HEAD_CASES[ci]
    k = f"head{ci}_"
    cam = types.SimpleNamespace(
        original_image=_t(PINS[k + "gt"], device), world_view_transform=_t(PINS[k + "world_view_transform"], device),
        orientation_field=_t(PINS[k + "orientation_field"], device), orientation_confidence=_t(PINS[k + "orientation_confidence"], device),
        mask=_t(PINS[k + "mask"], device) if with_mask else None, float_mask=None)
    cam.float_mask = cam.mask.float() if with_mask else None
    return k, H, W, with_mask, cam


@pytest.mark.parametrize("ci", range(len(HEAD_CASES)))
def test_torch_loss_function_reproduces_the_reference_run(ci, monkeypatch):
    """loss_function (loss/losses.py:319-355): five terms, the weighted total and its gradients w.r.t. the RGB image, the
    rendered mask channel, the rendered direction image and the endpoints.  `render` returns the fixture's prescribed images
    on both sides (the generator binds the reference's `render` the same way)."""
    from loss import losses as Ls
    k, H, W, with_mask, cam = _head_inputs(ci, "cpu")
    m = _model(int(PINS[k + "seed"]), PINS[k + "endpoints"], "cpu")
    x = _t(PINS[k + "image"], "cpu").requires_grad_(True)
    xo = _t(PINS[k + "omap"], "cpu").requires_grad_(True)
    xm = _t(PINS[k + "mask_render"], "cpu").requires_grad_(True)
    calls = []

    def render(camera, pc, bg, scaling_modifier=1.0, override_color=None, debug=False):
        calls.append(1)
        return {"render": xm if len(calls) == 1 and with_mask else xo}
    monkeypatch.setattr(Ls, "render", render)
    monkeypatch.setattr(Ls, "fused_losses", False)
    loss, terms = Ls.loss_function(m, x, cam, OptimizationParams())
    assert abs(float(loss) - float(PINS[k + "total"])) <= 1e-6 * float(PINS[k + "total"])
    for n in ("l1", "dssim", "mask", "orientation", "smooth"):
        want = float(PINS[k + "term_" + n])
        if want != want:
            assert n not in terms
        else:
            assert abs(float(terms[n]) - want) <= 1e-6 * max(abs(want), 1e-3), n
    gi, go = torch.autograd.grad(loss, [x, xo], retain_graph=True)
    _close(gi, PINS[k + "d_image"], 1e-6, "d total / d image")
    _close_direction_gradient(go, k, 1e-5, 1e-5)
    if with_mask:
        gm, = torch.autograd.grad(loss, xm, retain_graph=True)
        _close(gm, PINS[k + "d_mask_render"], 1e-6, "d total / d mask channel")
    ge, = torch.autograd.grad(loss, m._endpoints, allow_unused=True)
    _close(ge if ge is not None else torch.zeros_like(m._endpoints), PINS[k + "d_endpoints"], 1e-5, "d total / d endpoints", floor=1e-12)


# ---- GPU: the HIP kernels against the reference run ------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("ci", range(len(SSIM_CASES)))
def test_hip_ssim_l1_against_the_reference_run(ci):
    """hgs_ssim_l1_forward / _backward vs loss/losses.py:16-17,43-84 as executed: SSIM to 2e-5 absolute (an fp32 mean of ~1e4
    window quotients, hardware reciprocals), L1 to 1e-6 relative, gradients to 2e-4 of their largest element."""
    from hgs_runtime.fused import ssim_l1
    k = f"ssim{ci}_"
    x = _t(PINS[k + "img"], "cuda").requires_grad_(True)
    y = _t(PINS[k + "gt"], "cuda")
    s, l = ssim_l1(x, y)
    gs, = torch.autograd.grad(s, x, retain_graph=True)
    gl, = torch.autograd.grad(l, x)
    assert abs(float(s) - float(PINS[k + "ssim"])) <= 2e-5
    assert abs(float(l) - float(PINS[k + "l1"])) <= 1e-6 * float(PINS[k + "l1"])
    _close(gs, PINS[k + "d_ssim"], 2e-4, "d ssim / d image")
    _close(gl, PINS[k + "d_l1"], 1e-6, "d l1 / d image")


@pytest.mark.gpu
@pytest.mark.parametrize("seed", MODEL_SEEDS)
def test_hip_strand_geometry_and_smoothness_against_the_reference_run(seed):
    """hgs_strand_geometry_* (midpoint, scale, direction; forward and backward) and hgs_smoothness_* vs the reference getters
    scene/hair_gaussian_model.py:135-201 and loss/losses.py:175-221 as executed."""
    from loss import losses as Ls
    k = f"model{seed}_"
    m = _model(seed, PINS[k + "endpoints"], "cuda")
    assert m.fused_geometry
    xyz, scaling, _, orientation = m.derived_gaussians()
    _close(xyz, PINS[k + "xyz"], 1e-6, "xyz")
    _close(scaling, PINS[k + "scaling"], 2e-6, "scaling")
    # collapsed segments (|d| < min_val): (1, 0, 0) on both sides; everywhere else unit vectors to 2e-6
    _close(orientation, PINS[k + "orientation"], 2e-6, "orientation")
    total = sum((v * _t(PINS[k + "up_" + n], "cuda")).sum() for n, v in (("xyz", xyz), ("scaling", scaling), ("orientation", orientation)))
    ge, gw = torch.autograd.grad(total, [m._endpoints, m._width])
    # the fixture's gradient holds all five getters; opacity and mask do not reach endpoints / width
    _close(ge, PINS[k + "d_endpoints"], 2e-4, "d / d endpoints")
    _close(gw, PINS[k + "d_width"], 1e-5, "d / d width")
    _close(m.get_opacity, PINS[k + "opacity"], 1e-6, "opacity")
    _close(m.get_mask, PINS[k + "mask"], 1e-6, "mask")
    assert Ls.fused_losses
    for th in (30, 3, 179):
        v = Ls.angle_smoothness_loss(m, threshold=float(th))
        want = float(PINS[k + f"smooth{th}_value"])
        assert abs(float(v) - want) <= 2e-5 * max(want, 1e-3), th
        if torch.is_tensor(v) and v.requires_grad:
            g, = torch.autograd.grad(v, m._endpoints)
            _close(g, PINS[k + f"smooth{th}_d_endpoints"], 2e-4, f"smoothness gradient, threshold {th}", floor=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("ci", range(len(HEAD_CASES)))
def test_hip_loss_head_against_the_reference_run(ci):
    """hgs_loss_head_forward / _backward (SSIM + L1 + mask BCE + orientation + the weighted total, one fused head) vs
    loss_function, loss/losses.py:319-355, as executed: every term, the total and the three image gradients."""
    import ctypes as C
    import hgs_runtime as rt
    from hgs_runtime.strand_step import head_params
    from loss import losses as Ls
    k, H, W, with_mask, cam = _head_inputs(ci, "cuda")
    dev = torch.device("cuda")
    m = _model(int(PINS[k + "seed"]), PINS[k + "endpoints"], "cuda")
    image, omap = _t(PINS[k + "image"], dev), _t(PINS[k + "omap"], dev)
    mask_img = _t(PINS[k + "mask_render"][0], dev).contiguous()
    opt = OptimizationParams()
    row = rt.ViewTargets()
    m8 = cam.mask.to(torch.uint8).contiguous() if with_mask else None
    fmask = cam.float_mask.contiguous() if with_mask else None
    row.image, row.orientation, row.confidence = cam.original_image.data_ptr(), cam.orientation_field.data_ptr(), cam.orientation_confidence.data_ptr()
    row.float_mask, row.mask = (fmask.data_ptr(), m8.data_ptr()) if with_mask else (0, 0)
    vm = cam.world_view_transform.cpu().numpy().reshape(-1)
    for j in range(16):
        row.viewmatrix[j] = float(vm[j])
        row.projmatrix[j] = float(np.eye(4).reshape(-1)[j])
    row.mask_count = float(m8.sum().item()) if with_mask else 0.0
    targets = torch.from_numpy(np.frombuffer(bytes(row), dtype=np.uint8).copy()).to(dev)
    n_smooth = int(m.smoothness_index_pairs().shape[0])
    hp = head_params(H, W, opt, 0, 0, float(m.min_val), with_mask)
    L = rt.lib()
    scratch = torch.empty(L.hgs_loss_head_scratch_floats(C.byref(hp)), device=dev)
    out = torch.zeros(rt.HEAD_NOUT, device=dev)
    d_img, d_mask, d_omap = torch.empty(3, H, W, device=dev), torch.zeros(H, W, device=dev), torch.empty(3, H, W, device=dev)
    one = torch.ones(1, device=dev)
    rt.check(L.hgs_loss_head_forward(rt.current_stream(), C.byref(hp), image.data_ptr(), mask_img.data_ptr(), omap.data_ptr(),
                                     targets.data_ptr(), None, None, scratch.data_ptr(), out.data_ptr(), None, None))
    rt.check(L.hgs_loss_head_backward(rt.current_stream(), C.byref(hp), image.data_ptr(), mask_img.data_ptr(), omap.data_ptr(),
                                      targets.data_ptr(), None, None, scratch.data_ptr(), out.data_ptr(), one.data_ptr(), 0,
                                      d_img.data_ptr(), d_mask.data_ptr(), d_omap.data_ptr(), None))
    o = dict(zip(rt.HEAD_OUT, out.tolist()))
    assert abs(o["l1"] - float(PINS[k + "term_l1"])) <= 1e-6 * float(PINS[k + "term_l1"])
    assert abs(o["dssim"] - float(PINS[k + "term_dssim"])) <= 2e-5
    assert abs(o["orientation"] - float(PINS[k + "term_orientation"])) <= 1e-5 * float(PINS[k + "term_orientation"])
    if with_mask:
        assert abs(o["mask"] - float(PINS[k + "term_mask"])) <= 1e-5 * float(PINS[k + "term_mask"])
    # the head was given no smoothness pairs (n_smooth = 0 in head_params): its total is the reference's minus that term
    want_total = float(PINS[k + "total"]) - opt.lambda_smooth * float(PINS[k + "term_smooth"])
    assert abs(o["total"] - want_total) <= 2e-5 * want_total, (o["total"], want_total)
    _close(d_img, PINS[k + "d_image"], 2e-4, "d total / d image")
    _close_direction_gradient(d_omap, k, 2e-4, 1e-5)
    if with_mask:
        _close(d_mask, PINS[k + "d_mask_render"][0], 1e-5, "d total / d mask channel")
    assert n_smooth > 0
    # the smoothness term of the same model through its own kernel: the piece the head was not given
    v = Ls.angle_smoothness_loss(m)
    assert abs(float(v) - float(PINS[k + "term_smooth"])) <= 2e-5 * float(PINS[k + "term_smooth"])
