"""End-to-end on the GPU: render() drop-in with the model containers, autograd through 3 raster passes,
one-view training steps."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_render_dict_and_autograd_against_oracle():
    """render() on a HairGaussianModel: image / radii match the oracle, and d(loss)/d(endpoints) obtained through
    autograd (rasterizer backward + torch chain rule) matches oracle raster gradients pushed through the same chain."""
    from gaussian_renderer import render
    from oracle import hgs_oracle as O
    from synthetic import build_workload
    model, cams, _ = build_workload("tiny", device="cuda", with_targets=False)
    cam = cams[1]
    bg = torch.tensor([0.1, 0.2, 0.3], device="cuda")
    pkg = render(cam, model, bg)
    assert set(pkg) == {"render", "viewspace_points", "visibility_filter", "radii"}
    H, W = cam.image_height, cam.image_width
    assert pkg["render"].shape == (3, H, W) and pkg["radii"].dtype == torch.int32
    w = torch.randn(3, H, W, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0))
    (pkg["render"] * w).sum().backward()
    import math
    s = dict(means3D=model.get_xyz.detach().cpu().numpy(), opacities=model.get_opacity.detach().cpu().numpy().reshape(-1),
             scales=model.get_scaling.detach().cpu().numpy(), rotations=model.get_rotation.detach().cpu().numpy(),
             shs=model.get_features.detach().cpu().numpy(), colors_precomp=None, cov3D_precomp=None,
             viewmatrix=cam.world_view_transform.cpu().numpy(), projmatrix=cam.full_proj_transform.cpu().numpy(),
             campos=cam.camera_center.cpu().numpy(), bg=bg.cpu().numpy(), tanfovx=math.tan(cam.FoVx * 0.5),
             tanfovy=math.tan(cam.FoVy * 0.5), W=W, H=H, sh_degree=0, scale_modifier=1.0)
    f = O.forward(s)
    np.testing.assert_array_equal(pkg["radii"].cpu().numpy(), f["radii"])
    assert np.abs(pkg["render"].detach().cpu().numpy() - f["out_color"]).max() < 1e-3
    g = O.backward(s, f, w.cpu().numpy())
    vs = pkg["viewspace_points"].grad.cpu().numpy()
    sc = np.abs(g["dL_dmeans2D"]).max()
    assert np.abs(vs - g["dL_dmeans2D"]).max() <= 2e-3 * sc
    # chain rule to the shared endpoints with torch on the oracle's raster gradients
    m2 = model
    ep = m2._endpoints.detach().clone().requires_grad_(True)
    saved = m2._endpoints
    m2._endpoints = ep
    chain = (m2.get_xyz * torch.from_numpy(g["dL_dmeans3D"]).cuda()).sum() + \
        (m2.get_scaling * torch.from_numpy(g["dL_dscales"]).cuda()).sum() + \
        (m2.get_rotation * torch.from_numpy(g["dL_drotations"]).cuda()).sum()
    chain.backward()
    m2._endpoints = saved
    ref = ep.grad
    got = saved.grad
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max() <= 2e-3 * ref.abs().max()


def test_training_steps_reduce_loss_and_update_stats():
    from arguments import OptimizationParams
    from synthetic import build_workload
    from train import ViewSampler, training_step
    from utils.general import safe_state
    safe_state(True)
    model, cams, extent = build_workload("tiny", device="cuda", with_targets=True)
    opt = OptimizationParams()
    opt.enable_topology = False
    model.training_setup(opt)
    # start away from the optimum: perturb colours and opacity, keep the geometry
    with torch.no_grad():
        model._features_dc.add_(0.8)
        model._opacity.sub_(1.0)
    bg = torch.zeros(3, device="cuda")
    sampler = ViewSampler(cams, seed=0)
    losses = []
    for it in range(1, 61):
        loss, terms, pkg = training_step(model, sampler.next(), opt, bg, it, extent=extent)
        losses.append(float(loss))
        assert set(terms) >= {"l1", "dssim", "mask", "orientation", "smooth"}
    assert np.isfinite(losses).all()
    assert np.mean(losses[-10:]) < np.mean(losses[:10])
    assert float(model.denom.sum()) > 0 and float(model.xyz_gradient_accum.sum()) > 0
    assert float(model.max_radii2D.max()) > 0
    for g in model.optimizer.param_groups:
        assert torch.isfinite(g["params"][0]).all(), g["name"]


def test_stage1_cloud_model_step():
    from arguments import OptimizationParams
    from synthetic import attach_targets, cameras_extent, make_cameras, make_cloud_model
    from train import training_step
    cams = make_cameras(3, 200, 120, device="cuda")
    model = make_cloud_model(3000, device="cuda", spatial_lr_scale=cameras_extent(cams))
    assert float(model.get_scaling.min()) > 0  # distCUDA2-initialised scales
    attach_targets(cams, model)
    opt = OptimizationParams()
    opt.enable_topology = False
    model.training_setup(opt)
    bg = torch.zeros(3, device="cuda")
    for it in range(1, 6):
        loss, terms, _ = training_step(model, cams[it % 3], opt, bg, it)
        assert torch.isfinite(loss)
    assert "smooth" not in terms


def test_spatially_sorted_cloud_is_the_same_cloud():
    """Stage-I cloud: training_setup() re-orders a GPU cloud along a Morton curve (GaussianModel.sort_spatially; the
    binning kernels' atomics then meet on a handful of tile counters per workgroup instead of all of them).  The order of
    a cloud's Gaussians carries no meaning: the render agrees to rounding (depth ties aside, blending order is the
    depth order) and the gradients are the unsorted model's, permuted."""
    from arguments import OptimizationParams
    from gaussian_renderer import render
    from synthetic import attach_targets, cameras_extent, make_cameras, make_cloud_model
    cams = make_cameras(3, 200, 120, device="cuda")
    bg = torch.zeros(3, device="cuda")
    out = {}
    for mode in ("unsorted", "sorted"):
        torch.manual_seed(0)
        model = make_cloud_model(3000, device="cuda", spatial_lr_scale=cameras_extent(cams))
        opt = OptimizationParams()
        opt.spatial_sort = mode == "sorted"
        xyz0 = model.get_xyz.detach().clone()
        model.training_setup(opt)
        if mode == "sorted":
            # which old Gaussian sits at each new position (centres are distinct)
            same = (model.get_xyz.detach()[:, None, :] == xyz0[None, :, :]).all(dim=-1)
            assert bool((same.sum(dim=1) == 1).all())
            perm = same.to(torch.int8).argmax(dim=1)
            assert sorted(perm.tolist()) == list(range(3000))
            assert not torch.equal(perm, torch.arange(3000, device="cuda"))
        img = render(cams[1], model, bg)["render"]
        (img ** 2).sum().backward()
        out[mode] = (img.detach().clone(), model._xyz.grad.clone(), model._opacity.grad.clone(), model._features_dc.grad.clone())
    assert (out["sorted"][0] - out["unsorted"][0]).abs().max() <= 2e-6
    for a, b in zip(out["sorted"][1:], out["unsorted"][1:]):
        assert (a - b[perm]).abs().max() <= 1e-4 * max(float(b.abs().max()), 1e-12)


def test_derived_geometry_is_never_stale():
    """The getters recompute from the parameters on every call (reference scene/hair_gaussian_model.py:134-172): writes
    that autograd's version counters do not see -- `.data` mutation, raw-pointer kernels -- are visible at once, with and
    without grad mode, and a render after such a write shows the moved geometry."""
    from gaussian_renderer import render
    from synthetic import build_workload
    model, cams, _ = build_workload("tiny", device="cuda", with_targets=False)
    bg = torch.zeros(3, device="cuda")

    def means():
        return model._endpoints.detach()[model.endpoint_pairs].mean(dim=1)
    for grad in (False, True):
        with torch.set_grad_enabled(grad):
            a = model.get_xyz.detach().clone()
            img_a = render(cams[1], model, bg)["render"].detach().clone()
            v = model._endpoints._version
            model._endpoints.data.add_(0.01)
            assert model._endpoints._version == v          # the write is invisible to the version counter
            b = model.get_xyz.detach().clone()
            img_b = render(cams[1], model, bg)["render"].detach().clone()
        assert torch.allclose(b, means(), atol=1e-7) and torch.allclose(b - a, torch.full_like(a, 0.01), atol=1e-6)
        assert not torch.equal(img_a, img_b)
        w0 = model.get_scaling.detach().clone()
        model._width.data.add_(0.5)
        assert torch.allclose(model.get_scaling[:, 1], w0[:, 1] * float(np.exp(0.5)), rtol=1e-5)
    assert not hasattr(model, "_derived")


def test_fused_strand_geometry_matches_torch_formulas():
    """hgs_strand_geometry_* vs the op-by-op getters (which restate the reference's), forward and backward."""
    from synthetic import make_strand_model
    m = make_strand_model(50, 20, device="cuda")
    with torch.no_grad():  # include one collapsed segment and one anti-parallel-to-x segment
        m._endpoints[1] = m._endpoints[0]
        m._endpoints[5] = m._endpoints[4] + torch.tensor([-0.002, 0.0, 0.0], device="cuda")
    w = [torch.randn(s, device="cuda", generator=torch.Generator(device="cuda").manual_seed(i)) for i, s in
         enumerate([(1000, 3), (1000, 3), (1000, 4), (1000, 3)])]
    outs, grads = {}, {}
    for fused in (False, True):
        m.fused_geometry = fused
        m._endpoints.grad = None
        m._width.grad = None
        o = (m.get_xyz, m.get_scaling, m.get_rotation, m.get_orientation)
        sum((a * b).sum() for a, b in zip(o, w)).backward()
        outs[fused] = [t.detach().clone() for t in o]
        grads[fused] = (m._endpoints.grad.clone(), m._width.grad.clone())
    # rows where d is nearly -x_hat: the reference formula R = I + K + K^2/(1+c) cancels catastrophically there
    # (error ~ 1e-7/(1+c)); compare those through the rotation's action instead of against the torch quaternion
    dirs = outs[True][3]
    well = (1 + dirs[:, 0]) > 0.05
    for a, b, name in zip(outs[True], outs[False], ("xyz", "scale", "quat", "dir")):
        assert (a - b)[well].abs().max() <= 5e-6 * max(1.0, float(b.abs().max())), name
    from utils.transform import build_rotation
    Rq = build_rotation(outs[True][2])
    live = outs[True][1][:, 0] > 2e-7  # not collapsed
    assert (Rq[live][:, :, 0] - dirs[live]).abs().max() < 2e-6      # R(q) x_hat = d for every non-collapsed segment
    assert torch.equal(outs[True][2][0], torch.tensor([1.0, 0, 0, 0], device="cuda"))  # collapsed -> identity
    ge_t, gw_t = grads[False]
    ge_f, gw_f = grads[True]
    touched = torch.zeros(ge_t.shape[0], dtype=torch.bool, device="cuda")
    touched[m.endpoint_pairs[~well].flatten()] = True
    assert (ge_f - ge_t)[~touched].abs().max() <= 2e-4 * ge_t.abs().max()
    assert (gw_f - gw_t).abs().max() <= 1e-5 * max(1e-12, float(gw_t.abs().max()))
    assert torch.isfinite(ge_f).all()


def test_fused_losses_match_torch_ops():
    """hgs_ssim_l1_* and hgs_orientation_loss_* vs the PyTorch implementations (which restate loss/losses.py)."""
    from arguments import OptimizationParams
    from loss import losses as Ls
    from synthetic import build_workload
    model, cams, _ = build_workload("tiny", device="cuda", with_targets=True)
    cam = cams[2]
    opt = OptimizationParams()
    g = torch.Generator(device="cuda").manual_seed(3)
    img = (cam.original_image + 0.1 * torch.randn(cam.original_image.shape, device="cuda", generator=g)).clamp(0, 1)
    res = {}
    for fused in (False, True):
        Ls.fused_losses = fused
        x = img.clone().requires_grad_(True)
        if fused:
            from hgs_runtime.fused import ssim_l1
            s, l = ssim_l1(x, cam.original_image)
        else:
            s, l = Ls.ssim(x, cam.original_image), Ls.l1_loss(x, cam.original_image)
        (0.3 * (1 - s) + 0.7 * l).backward()
        res[fused] = (float(s), float(l), x.grad.clone())
    Ls.fused_losses = True
    assert abs(res[True][0] - res[False][0]) < 2e-5 and abs(res[True][1] - res[False][1]) < 1e-6
    gt_, gf_ = res[False][2], res[True][2]
    assert (gf_ - gt_).abs().max() <= 2e-4 * gt_.abs().max()
    # orientation loss through the full render, both implementations, gradients w.r.t. the endpoints
    out = {}
    for fused in (False, True):
        Ls.fused_losses = fused
        model._endpoints.grad = None
        v = Ls.orientation_loss_rast(model, cam, opt)
        v.backward()
        out[fused] = (float(v), model._endpoints.grad.clone())
    Ls.fused_losses = True
    assert abs(out[True][0] - out[False][0]) <= 1e-5 * abs(out[False][0])
    assert (out[True][1] - out[False][1]).abs().max() <= 5e-4 * out[False][1].abs().max()


def test_async_capacity_mode_matches_blocking_and_recovers_from_overflow():
    """set_async: no per-pass host sync; results identical to the blocking mode; an under-sized capacity is detected
    by check_async() and the step is repeated by training_step."""
    from arguments import OptimizationParams
    from diff_gaussian_rasterization import _C as raster
    from gaussian_renderer import render
    from synthetic import build_workload
    from train import training_step
    model, cams, extent = build_workload("tiny", device="cuda", with_targets=True)
    bg = torch.zeros(3, device="cuda")
    with torch.no_grad():
        ref = [render(c, model, bg)["render"].clone() for c in cams]
    try:
        raster.set_async(True)
        with torch.no_grad():
            got = [render(c, model, bg)["render"].clone() for c in cams]   # first call learns the capacity (blocking)
            counts = raster.check_async()
        assert len(counts) == 1 and counts[0] > 0          # the sticky device-side maximum over the async passes
        assert raster.check_async() == []                   # nothing issued since
        for a, b in zip(got, ref):
            assert torch.equal(a, b)
        # force an overflow: shrink the capacity below what the scene needs
        raster._state["cap"] = 64
        with torch.no_grad():
            render(cams[0], model, bg)
        with pytest.raises(raster.HgsCapacityOverflow):
            raster.check_async()
        assert raster._state["cap"] > 64
        # training_step repeats the step transparently
        opt = OptimizationParams()
        opt.enable_topology = False
        model.training_setup(opt)
        raster._state["cap"] = 64
        before = model._endpoints.detach().clone()
        loss, _, _ = training_step(model, cams[1], opt, bg, 1, extent=extent)
        assert torch.isfinite(loss) and not torch.equal(before, model._endpoints.detach())
        # the per-ITERATION bookkeeping is not repeated with the step: an overflow on an iteration that bumps the SH degree
        # (every 1000th, train.py:136-137) bumps it once (round 3 bumped it per attempt and skipped a degree)
        bumps = []
        model.oneupSHdegree = lambda: bumps.append(1)
        raster._state["cap"] = 64
        training_step(model, cams[2], opt, bg, 1000, extent=extent)
        assert raster._state["cap"] > 64 and len(bumps) == 1
    finally:
        raster.set_async(False)


def test_graphed_step_matches_eager_steps():
    """HIP-graph replay of the whole iteration == the eager training_step sequence (same views, same updates)."""
    from arguments import OptimizationParams
    from diff_gaussian_rasterization import _C as raster
    from synthetic import build_workload
    from train import GraphedStep, training_step
    from utils.general import safe_state
    results = {}
    order = [1, 3, 0, 2, 1, 0, 3, 2]
    try:
        for mode in ("eager", "graph"):
            safe_state(True)
            model, cams, extent = build_workload("tiny", device="cuda", with_targets=True)
            opt = OptimizationParams()
            opt.enable_topology = False
            model.training_setup(opt)
            bg = torch.zeros(3, device="cuda")
            losses = []
            if mode == "eager":
                from hgs_runtime.strand_step import FusedStrandStep
                fused = FusedStrandStep(model, cams, opt, bg)  # the iteration the graph captures, launched eagerly
                for it, ci in enumerate(order, 1):
                    loss, _, _ = training_step(model, cams[ci], opt, bg, it, extent=extent, fused=fused)
                    losses.append(float(loss))
            else:
                gs = GraphedStep(model, cams, opt, bg, extent=extent)
                gs.capture(cams)
                for it, ci in enumerate(order, 1):
                    losses.append(float(gs.step(cams[ci], it)))
                counts = gs.check()
                assert len(counts) == 1 and min(counts) > 0   # single-pass mode: one raster traversal per step
                raster.set_async(False)
            results[mode] = (losses, model._endpoints.detach().clone(), model._opacity.detach().clone(),
                             model.denom.clone(), model.xyz_gradient_accum.clone())
    finally:
        raster.set_async(False)
    le, lg = results["eager"][0], results["graph"][0]
    assert np.allclose(le, lg, rtol=1e-4, atol=1e-6), (le, lg)
    for a, b in zip(results["eager"][1:], results["graph"][1:]):
        assert (a - b).abs().max() <= 1e-4 * max(1e-6, float(a.abs().max()))
    assert float(results["graph"][3].sum()) > 0


def test_eager_step_after_graph_replays_starts_without_gradients():
    """training() leaves its captured graph for an eager training_step whenever a topology operator is due.  The graph's
    static gradient tensors stay in `.grad` after a replay; the eager iteration must not accumulate onto them (the
    reference zeroes the gradients at the end of every iteration, train.py:203): eight replays + one eager step end
    bit-identical to eight replays + one more replay."""
    from arguments import OptimizationParams
    from diff_gaussian_rasterization import _C as raster
    from hgs_runtime.strand_step import ViewTable, fused_step_for
    from synthetic import build_workload
    from train import GraphedStep, training_step
    from utils.general import safe_state
    order = [1, 3, 0, 2, 1, 0, 3, 2, 3]
    res = {}
    try:
        for mode in ("replay", "eager_last"):
            safe_state(True)
            model, cams, extent = build_workload("tiny", device="cuda", with_targets=True)
            opt = OptimizationParams()
            opt.enable_topology = False
            model.training_setup(opt)
            bg = torch.zeros(3, device="cuda")
            views = ViewTable(cams)
            fused = fused_step_for(model, views, opt, bg)
            fused.defer_tail = True
            gs = GraphedStep(model, cams, opt, bg, extent=extent, views=views)
            gs.capture(cams)
            for it, ci in enumerate(order[:-1], 1):
                gs.step(cams[ci], it)
            if mode == "replay":
                gs.step(cams[order[-1]], len(order))
            else:
                assert model._endpoints.grad is not None          # what the replays left behind
                training_step(model, cams[order[-1]], opt, bg, len(order), extent=extent, fused=fused)
            torch.cuda.synchronize()
            raster.set_async(False)
            res[mode] = [g["params"][0].detach().clone() for g in model.optimizer.param_groups] + \
                        [model.optimizer.state[g["params"][0]]["exp_avg"].clone() for g in model.optimizer.param_groups
                         if g["params"][0].numel()]
    finally:
        raster.set_async(False)
    for a, b in zip(res["replay"], res["eager_last"]):
        assert torch.equal(a, b)


def test_iteration_prologue_equals_select_then_forward():
    """ViewTable.prologue() (view select + clearing of the image buffer in one launch, HGS_IMAGE_PREZEROED) against
    select() + a forward that clears its own buffer: identical loss, planes, gradients; the buffer is handed over once."""
    from arguments import OptimizationParams
    from hgs_runtime.strand_step import FusedStrandStep
    from synthetic import build_workload
    from utils.general import safe_state
    import ctypes
    import hgs_runtime as rt
    safe_state(True)
    model, cams, extent = build_workload("tiny", device="cuda", with_targets=True)
    opt = OptimizationParams()
    opt.enable_topology = False
    model.training_setup(opt)
    bg = torch.zeros(3, device="cuda")
    fused = FusedStrandStep(model, cams, opt, bg)
    params = [model._endpoints, model._width, model._opacity, model._mask, model._features_dc]
    lr = torch.zeros((), device="cuda")
    out = {}
    # (each twice: the buffer still holds the previous pass's counters; "ride": no launch of its own, the prologue rides in
    # spare workgroups of the iteration's first launch, HgsStrandFusion.prologue)
    for mode in ("select", "prologue", "prologue", "ride", "ride", "prologue"):
        for p in params:
            p.grad = None
        if mode == "select":
            fused.views.select(1, lr=0.5, lr_dst=lr)
            assert fused.views.take_image() is None
        elif mode == "prologue":
            fused.views.prologue(1, lr=0.25, lr_dst=lr)
            assert float(lr) == 0.25
        else:
            fused.views.select(0, lr=0.75, lr_dst=lr)    # (the slot holds another view until the rider has run)
            fused.views.prologue(1, lr=0.125, lr_dst=lr, ride=True)
            assert float(lr) == 0.75                     # nothing launched yet
        loss, _ = fused.loss()
        if mode == "ride":
            assert float(lr) == 0.125
        assert fused.views.take_image() is None          # consumed by the forward
        fused.backward(loss)
        cur = [loss.detach().clone(), fused.last["planes"].clone()] + [p.grad.clone() for p in params]
        if mode in out or "select" in out:
            for a, b in zip(out["select"], cur):
                assert torch.equal(a, b)
        out[mode] = cur
    off, nbytes = ctypes.c_size_t(0), ctypes.c_size_t(0)
    rt.check(rt.lib().hgs_image_zero_range(64, 48, ctypes.addressof(off), ctypes.addressof(nbytes)))
    assert off.value % 4 == 0 and 0 < nbytes.value <= rt.lib().hgs_image_bytes(64, 48) - off.value


def test_deferred_head_tail_gives_the_same_terms_and_gradients():
    """HgsHeadParams.defer_tail: the head's last sums ride in a spare workgroup of the backward's parameter launch instead
    of a launch of their own.  Every loss term, the planes and the gradients are bit-identical to the undeferred iteration --
    strand model and Stage-I cloud, unit and non-unit upstream gradient (the latter runs the tail before its per-pixel
    pass); before the backward the deferred terms are NOT complete (that is the contract)."""
    from arguments import OptimizationParams
    from hgs_runtime.strand_step import fused_step_for
    from synthetic import attach_targets, build_workload, cameras_extent, make_cameras, make_cloud_model
    from utils.general import safe_state
    safe_state(True)
    for kind in ("strands", "cloud"):
        if kind == "strands":
            model, cams, extent = build_workload("tiny", device="cuda", with_targets=True)
        else:
            cams = make_cameras(3, 200, 120, device="cuda")
            model = make_cloud_model(3000, device="cuda", spatial_lr_scale=cameras_extent(cams))
            attach_targets(cams, model)
        opt = OptimizationParams()
        opt.enable_topology = False
        model.training_setup(opt)
        bg = torch.zeros(3, device="cuda")
        fused = fused_step_for(model, cams, opt, bg)
        params = [g["params"][0] for g in model.optimizer.param_groups if g["params"][0].numel() > 0]
        for scale in (None, 0.5):
            ref = None
            for defer in (False, True, True):
                for p in params:
                    p.grad = None
                fused.defer_tail = defer
                fused.views.prologue(1, ride=defer)      # (and the prologue as a rider of the parameter forward launch)
                loss, terms = fused.loss()
                if scale is None:
                    fused.backward(loss)
                else:
                    (loss * scale).backward()
                torch.cuda.synchronize()
                cur = [terms[:14].clone(), fused.last["planes"].clone()] + [p.grad.clone() for p in params]   # (14 entries used)
                assert torch.isfinite(cur[0]).all() and float(cur[0][0]) > 0
                if ref is None:
                    ref = cur
                else:
                    for a, b in zip(ref, cur):
                        assert torch.equal(a, b)
        # the contract: with defer_tail the total is complete only after the backward
        fused.defer_tail = True
        fused.views.prologue(0)
        loss, terms = fused.loss()
        early = terms.clone()
        fused.backward(loss)
        torch.cuda.synchronize()
        assert float(early[0]) < float(terms[0])       # (the per-pixel terms were still missing from the total)
        assert torch.equal(early[1:3], terms[1:3])     # L1 and DSSIM are the forward's own


def test_several_steps_per_graph_equal_single_step_replays():
    """GraphedStep(steps_per_graph=4).step_many == four step() replays, bit for bit (same kernels, same order); the two
    graphs of one GraphedStep can be mixed (8 steps as 4 + 1 + 1 + ... )."""
    from arguments import OptimizationParams
    from diff_gaussian_rasterization import _C as raster
    from synthetic import build_workload
    from train import GraphedStep
    from utils.general import safe_state
    order = [1, 3, 0, 2, 1, 0, 3, 2, 2, 1]
    results = {}
    try:
        for mode in ("single", "many"):
            safe_state(True)
            model, cams, extent = build_workload("tiny", device="cuda", with_targets=True)
            opt = OptimizationParams()
            opt.enable_topology = False
            model.training_setup(opt)
            bg = torch.zeros(3, device="cuda")
            gs = GraphedStep(model, cams, opt, bg, extent=extent, steps_per_graph=4 if mode == "many" else 1)
            gs.capture(cams)
            losses = []
            if mode == "single":
                for it, ci in enumerate(order, 1):
                    losses.append(gs.step(cams[ci], it).clone())
            else:
                losses += [l.clone() for l in gs.step_many([cams[ci] for ci in order[0:4]], 1)]
                losses.append(gs.step(cams[order[4]], 5).clone())
                losses.append(gs.step(cams[order[5]], 6).clone())
                losses += [l.clone() for l in gs.step_many([cams[ci] for ci in order[6:10]], 7)]
            assert min(gs.check()) > 0
            raster.set_async(False)
            results[mode] = (torch.stack(losses), model._endpoints.detach().clone(), model._opacity.detach().clone(),
                             model._features_dc.detach().clone(), model.denom.clone(), model.xyz_gradient_accum.clone(),
                             model.max_radii2D.clone())
    finally:
        raster.set_async(False)
    for a, b in zip(results["single"], results["many"]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("use_graph", [False, True])
def test_training_loop_with_topology_changes(use_graph):
    """training(): densification + merging + opacity reset at their intervals, with graph re-capture after each."""
    from arguments import OptimizationParams
    from synthetic import build_workload
    from train import training
    from utils.general import safe_state
    safe_state(True)
    model, cams, extent = build_workload("tiny", device="cuda", with_targets=True)
    opt = OptimizationParams()
    opt.densify_from_iter, opt.densification_interval, opt.merge_interval, opt.opacity_reset_interval = 3, 6, 8, 12
    model.training_setup(opt)
    P0 = model.get_xyz.shape[0]
    ema = training(model, cams, opt, iterations=20, extent=extent, use_graph=use_graph)
    assert torch.isfinite(ema)
    P1 = model.get_xyz.shape[0]
    assert P1 != P0                                           # the topology did change
    pairs = model.endpoint_pairs
    assert int(pairs.max()) == model._endpoints.shape[0] - 1
    for g in model.optimizer.param_groups:
        p = g["params"][0]
        assert torch.isfinite(p).all() and p.shape[0] == (model._endpoints.shape[0] if g["name"] == "endpoints" else P1)
    assert float(model.get_opacity.max()) < 0.9               # opacity reset happened at it 12
    # sort_spatially (what the operators end with; the strands have moved since, so their order along the curve may have
    # changed) renumbers the strand bookkeeping instead of walking the chains again: the result is what a fresh walk gives,
    # and a second sort finds nothing to move
    model.compute_strands_info()          # (a walk on the CURRENT geometry: the root -> tip orientation is decided by it)
    model.sort_spatially()
    assert model.sort_spatially() is None
    kept = model.strands_info
    model.compute_strands_info()
    for a in ("offsets", "rows", "segment_rows", "id_to_strand_id", "strand_endpoint_id_to_complementary"):
        assert np.array_equal(np.asarray(getattr(kept, a)), np.asarray(getattr(model.strands_info, a))), a


def test_op_by_op_graph_loop_recaptures_cleanly_after_eager_iterations(recwarn):
    """training() with the op-by-op iteration (fused_step off) under graph replay: iterations on which a topology operator
    is due run eagerly on the current stream and the graph is captured again on its side stream afterwards.  Nothing of
    the eager iteration's autograd graph may survive it (a merge round that merges nothing keeps the parameters, whose
    AccumulateGrad nodes would then belong to the wrong stream: seen as a crash in capture_end at 100 k segments)."""
    from arguments import OptimizationParams
    from synthetic import build_workload
    from train import training
    from utils.general import safe_state
    safe_state(True)
    model, cams, extent = build_workload("tiny", device="cuda", with_targets=True)
    opt = OptimizationParams()
    opt.fused_step = False
    opt.merge_interval, opt.merge_dist_th_init, opt.merge_dist_th_final = 5, 1e-9, 1e-9     # merge rounds that find nothing
    model.training_setup(opt)
    P0 = model.get_xyz.shape[0]
    for chunk in range(2):
        ema = training(model, cams, opt, iterations=12, extent=extent, start_iteration=12 * chunk)
        assert torch.isfinite(ema)
    assert model.get_xyz.shape[0] == P0
    assert not [w for w in recwarn.list if "AccumulateGrad" in str(w.message)]


def test_capacity_overflow_mid_run_is_rolled_back_exactly(capsys, monkeypatch):
    """training() with a captured graph whose binning capacity is too small: the replays that overflow have zero gradients
    (include/hgs.h), the loop notices at its next headroom check, returns to its last checkpoint, re-captures with a larger
    capacity and runs those iterations again -- the run ends BIT-IDENTICAL to one that never overflowed (the reference
    computes every step with its gradient, train.py:146-204).  Case 1: the capacity is too small from the first replay on.
    Case 2: the strands become 100x wider at iteration 73, under the feet of a graph captured for thin ones -- the overflow
    starts after the checkpoint of iteration 64, which is what the loop returns to."""
    from arguments import OptimizationParams
    from diff_gaussian_rasterization import _C as raster
    from synthetic import attach_targets, cameras_extent, make_cameras, make_strand_model
    import train as T
    from utils.general import safe_state
    safe_state(True)
    cams = make_cameras(4, 400, 240, device="cuda")
    extent = cameras_extent(cams)

    def run(slack, iterations, widen_at=None):
        raster._state["cap"] = 0
        model = make_strand_model(300, 40, device="cuda", spatial_lr_scale=extent)
        model.compute_strands_info(only_foreground=True)
        attach_targets(cams, model)
        opt = OptimizationParams()
        opt.enable_topology = False
        opt.capacity_slack = slack
        model.training_setup(opt)
        set_lr = T.GraphedStep._set_lr

        def set_lr_and_widen(self, iteration):
            if iteration == widen_at:          # (host code of the step: runs again when the iteration is run again)
                with torch.no_grad():
                    model._width.add_(float(np.log(100.0)))
            return set_lr(self, iteration)
        monkeypatch.setattr(T.GraphedStep, "_set_lr", set_lr_and_widen)
        try:
            ema = T.training(model, cams, opt, iterations=iterations, extent=extent)
        finally:
            monkeypatch.setattr(T.GraphedStep, "_set_lr", set_lr)
        state = [g["params"][0].detach().clone() for g in model.optimizer.param_groups]
        for g in model.optimizer.param_groups:
            st = model.optimizer.state[g["params"][0]]
            state += [st["exp_avg"].clone(), st["exp_avg_sq"].clone(), st["step"].clone()]
        state += [model.max_radii2D.clone(), model.xyz_gradient_accum.clone(), model.denom.clone(), ema.clone()]
        return state, T.training.last_rollbacks

    try:
        small, rb_small = run(0.3, 100)          # capacity 0.3 x the busiest view: every replay overflows until the first check
        roomy, rb_roomy = run(4.0, 100)
        assert rb_small >= 1 and rb_roomy == 0
        assert "are run again from the last checkpoint" in capsys.readouterr().out
        for a, b in zip(small, roomy):
            assert torch.equal(a, b)
        small2, rb2 = run(1.0, 160, widen_at=73)
        out2 = capsys.readouterr().out
        roomy2, rb2r = run(200.0, 160, widen_at=73)
        assert rb2 >= 1 and rb2r == 0
        assert "iterations 65.." in out2         # returned to the checkpoint of iteration 64, not to the start
        for a, b in zip(small2, roomy2):
            assert torch.equal(a, b)
    finally:
        raster._state["cap"] = 0
        raster.set_async(False)


def test_single_pass_equals_three_passes():
    """render_multi (7 channels, one traversal) vs the reference's three render() calls: identical images, gradients
    equal up to fp32 summation order, RGB-only screen-space gradient for the densification statistics."""
    from arguments import OptimizationParams
    from gaussian_renderer import render, render_multi
    from loss import losses as Ls
    from synthetic import build_workload
    model, cams, _ = build_workload("tiny", device="cuda", with_targets=True)
    cam = cams[1]
    bg = torch.zeros(3, device="cuda")
    extra = torch.cat((model.get_mask, model.get_orientation), dim=1)
    with torch.no_grad():
        a = render(cam, model, bg)["render"]
        m = render(cam, model, bg, override_color=model.get_mask.repeat(1, 3))["render"][0]
        o = render(cam, model, bg, override_color=model.get_orientation)["render"]
        pk = render_multi(cam, model, bg, extra)
    assert torch.equal(pk["render"], a) and torch.equal(pk["extra"][0], m) and torch.equal(pk["extra"][1:4], o)
    with torch.no_grad():
        pk2 = render_multi(cam, model, bg, extra, splits=(1, 3))
    assert pk2["extra"][0].shape == m.shape and torch.equal(pk2["extra"][0], m) and torch.equal(pk2["extra"][1], o)
    opt = OptimizationParams()
    res = {}
    for single in (False, True):
        for g in model.optimizer.param_groups if model.optimizer else []:
            pass
        for p in (model._endpoints, model._features_dc, model._opacity, model._mask, model._width):
            p.grad = None
        if single:
            loss, terms, pkg = Ls.loss_function_single_pass(model, cam, opt, bg)
        else:
            pkg = render(cam, model, bg)
            loss, terms = Ls.loss_function(model, pkg["render"], cam, opt)
        loss.backward()
        res[single] = (float(loss), {k: float(v) for k, v in terms.items()}, pkg["viewspace_points"].grad.clone(),
                       [p.grad.clone() for p in (model._endpoints, model._features_dc, model._opacity, model._mask, model._width)])
    assert abs(res[True][0] - res[False][0]) <= 1e-5 * abs(res[False][0])
    for k in res[False][1]:
        assert abs(res[True][1][k] - res[False][1][k]) <= 1e-5 * max(1e-6, abs(res[False][1][k])), k
    vs_t, vs_f = res[True][2], res[False][2]
    assert (vs_t - vs_f).abs().max() <= 2e-4 * vs_f.abs().max()          # RGB-only dL/dmean2D
    for gt_, gf_ in zip(res[True][3], res[False][3]):
        assert (gt_ - gf_).abs().max() <= 2e-4 * gf_.abs().max()


def test_fused_adam_matches_torch_adam():
    from hgs_runtime.fused import FusedAdam
    g = torch.Generator(device="cuda").manual_seed(0)
    shapes = [(1000, 3), (777, 1, 3), (777, 1), (5,), (64, 4)]
    ref_p = [torch.randn(s, device="cuda", generator=g).requires_grad_(True) for s in shapes]
    my_p = [p.detach().clone().requires_grad_(True) for p in ref_p]
    lrs = [1.6e-4, 0.025, 0.05, 0.01, 0.005]
    ref = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(ref_p, lrs)], lr=0.0, eps=1e-15)
    mine = FusedAdam([{"params": [p], "lr": lr} for p, lr in zip(my_p, lrs)], lr=0.0, eps=1e-15)
    for it in range(25):
        for a, b in zip(ref_p, my_p):
            gr = torch.randn(a.shape, device="cuda", generator=g) * (10.0 ** ((it % 5) - 3))
            a.grad, b.grad = gr.clone(), gr.clone()
        if it == 10:
            ref.param_groups[0]["lr"] = mine.param_groups[0]["lr"] = 3e-5     # lr schedule change
        ref.step()
        mine.step()
    for a, b in zip(ref_p, my_p):
        assert (a - b).abs().max() <= 2e-6 * max(1.0, float(a.abs().max()))
    assert float(mine.state[my_p[0]]["step"]) == 25.0


def test_fused_smoothness_matches_torch_ops():
    from loss import losses as Ls
    from synthetic import make_strand_model
    m = make_strand_model(60, 25, device="cuda")
    m.compute_strands_info()
    out = {}
    for fused in (False, True):
        Ls.fused_losses = fused
        m._endpoints.grad = None
        v = Ls.angle_smoothness_loss(m, threshold=3.0)
        v.backward()
        out[fused] = (float(v), m._endpoints.grad.clone())
    Ls.fused_losses = True
    assert out[False][0] > 0 and abs(out[True][0] - out[False][0]) <= 2e-5 * out[False][0]
    assert (out[True][1] - out[False][1]).abs().max() <= 2e-4 * out[False][1].abs().max()
    Ls.fused_losses = True
    assert float(Ls.angle_smoothness_loss(m, threshold=179.0)) == 0.0      # nothing selected -> 0


@pytest.mark.parametrize("hw", [(61, 97), (64, 100), (33, 17)])
def test_fused_ssim_odd_sizes(hw):
    """Image sizes that are not multiples of the 32-px block / of 4 (scalar halo-load path)."""
    from hgs_runtime.fused import ssim_l1
    from loss import losses as Ls
    H, W = hw
    g = torch.Generator(device="cuda").manual_seed(H * 1000 + W)
    a = torch.rand(3, H, W, device="cuda", generator=g).requires_grad_(True)
    b = torch.rand(3, H, W, device="cuda", generator=g)
    s, l = ssim_l1(a, b)
    (s + 2 * l).backward()
    ga = a.grad.clone()
    a.grad = None
    s2, l2 = Ls.ssim(a, b), Ls.l1_loss(a, b)
    (s2 + 2 * l2).backward()
    assert abs(float(s) - float(s2)) < 2e-5 and abs(float(l) - float(l2)) < 1e-6
    assert (ga - a.grad).abs().max() <= 2e-4 * a.grad.abs().max()


@pytest.mark.parametrize("workload", ["tiny", "north_star"])
def test_fused_iteration_matches_op_by_op_path(workload):
    """hgs_runtime.strand_step (view table + fused parameter/loss-head kernels, one autograd node) against the
    op-by-op path (getters, render_multi, loss_function_single_pass, update_densification_stats): same loss terms,
    same gradients of every parameter group, same densification statistics.  north_star: the BASELINE size (100 000
    strand-Gaussians at 1080p), where the tile culling, the SSIM block lists and the capacity-mode binning all have
    something to do."""
    from arguments import OptimizationParams
    from hgs_runtime.strand_step import FusedStrandStep
    from loss.losses import loss_function_single_pass
    from synthetic import build_workload
    from utils.general import safe_state
    safe_state(True)
    model, cams, _ = (build_workload("tiny", device="cuda", with_targets=True) if workload == "tiny"
                      else build_workload(workload, device="cuda", with_targets=True, n_views=3))
    opt = OptimizationParams()
    model.training_setup(opt)
    bg = torch.zeros(3, device="cuda")
    params = [model._endpoints, model._width, model._opacity, model._mask, model._features_dc, model._features_rest]
    # the synthetic targets are renders of this very model: move the parameters off that point, where the L1 term's
    # sign(image - gt) is not differentiable (a 1-ulp difference of the image would flip whole gradients)
    g = torch.Generator(device="cuda").manual_seed(5)
    with torch.no_grad():
        model._features_dc.add_(0.2 * torch.randn(model._features_dc.shape, device="cuda", generator=g))
        model._endpoints.add_(0.003 * torch.randn(model._endpoints.shape, device="cuda", generator=g))
        model._opacity.add_(0.3 * torch.randn(model._opacity.shape, device="cuda", generator=g))
        model._mask.add_(0.3 * torch.randn(model._mask.shape, device="cuda", generator=g))
    fused = FusedStrandStep(model, cams, opt, bg)
    for ci in (2, 0):
        cam = cams[ci]
        # ---- op-by-op
        for p in params:
            p.grad = None
        for t in (model.max_radii2D, model.xyz_gradient_accum, model.denom):
            t.zero_()
        loss, terms, pkg = loss_function_single_pass(model, cam, opt, bg)
        loss.backward()
        with torch.no_grad():
            model.update_densification_stats(pkg["viewspace_points"], pkg["radii"], pkg["visibility_filter"])
        ref = dict(loss=float(loss), terms={k: float(v) for k, v in terms.items()},
                   grads=[p.grad.clone() for p in params],
                   stats=[t.clone() for t in (model.max_radii2D, model.xyz_gradient_accum, model.denom)],
                   image=pkg["render"].detach().clone())
        # ---- fused
        for p in params:
            p.grad = None
        for t in (model.max_radii2D, model.xyz_gradient_accum, model.denom):
            t.zero_()
        fused.views.select(fused.views.index[id(cam)])
        floss, _ = fused.loss()
        if ci == 2:
            floss.backward()               # generic upstream gradient: the per-pixel terms' backward kernel runs
        else:
            fused.backward(floss)          # known unit gradient: the planes written by the forward pass are used
        fused.update_densification_stats()
        fterms = {k: float(v) for k, v in fused.terms().items()}
        assert abs(float(floss) - ref["loss"]) <= 2e-5 * abs(ref["loss"]), (float(floss), ref["loss"])
        for k, v in ref["terms"].items():
            assert abs(fterms[k] - v) <= 2e-5 * max(abs(v), 1e-3), (k, fterms[k], v)
        assert (fused.last["planes"][:3] - ref["image"]).abs().max() <= 2e-6
        for name, p, gref in zip(("endpoints", "width", "opacity", "mask", "f_dc", "f_rest"), params, ref["grads"]):
            if gref.numel() == 0:
                continue
            scale = float(gref.abs().max())
            assert (p.grad - gref).abs().max() <= 2e-4 * max(scale, 1e-12), (name, float((p.grad - gref).abs().max()), scale)
        for t, sref in zip((model.max_radii2D, model.xyz_gradient_accum, model.denom), ref["stats"]):
            assert (t - sref).abs().max() <= 1e-4 * max(float(sref.abs().max()), 1e-12)
        assert float(model.denom.sum()) > 0


def test_view_table_select_switches_targets():
    from hgs_runtime.strand_step import ViewTable
    from synthetic import build_workload
    _, cams, _ = build_workload("tiny", device="cuda", with_targets=True)
    vt = ViewTable(cams)
    lr = torch.zeros((), device="cuda")
    for i in (1, 3):
        vt.select(i, lr=0.125 * i, lr_dst=lr)
        torch.cuda.synchronize()
        assert torch.equal(vt.viewmatrix.reshape(4, 4), cams[i].world_view_transform)
        assert torch.equal(vt.projmatrix.reshape(4, 4), cams[i].full_proj_transform)
        assert torch.equal(vt.campos, cams[i].camera_center)
        assert float(lr) == 0.125 * i
        ptrs = vt.slot[:40].view(torch.int64).tolist()
        assert ptrs[0] == cams[i].original_image.data_ptr() and ptrs[2] == cams[i].orientation_field.data_ptr()
    with pytest.raises(Exception):
        vt.select(len(cams))


def test_fused_cloud_iteration_matches_op_by_op_path():
    """Stage-I Gaussian cloud: hgs_cloud_params_* + the shared raster / loss head, one autograd node, against the getters
    (exp / normalize / sigmoid / build_rotation @ one-hot), render_multi, loss_function_single_pass and
    update_densification_stats: same loss terms, gradients of all 7 parameter groups, statistics."""
    from arguments import OptimizationParams
    from hgs_runtime.strand_step import FusedCloudStep, ViewTable
    from loss.losses import loss_function_single_pass
    from scene.gaussian_model import GaussianModel
    from synthetic import build_workload
    from utils.general import safe_state
    safe_state(True)
    _, cams, _ = build_workload("tiny", device="cuda", with_targets=True)       # cameras + targets of the strand scene
    g = torch.Generator(device="cuda").manual_seed(11)
    P = 3000
    m = GaussianModel(sh_degree=1, device="cuda")
    m._xyz = torch.nn.Parameter((torch.rand(P, 3, device="cuda", generator=g) - 0.5) * 0.25)
    m._features_dc = torch.nn.Parameter(torch.randn(P, 1, 3, device="cuda", generator=g) * 0.5)
    m._features_rest = torch.nn.Parameter(torch.randn(P, 3, 3, device="cuda", generator=g) * 0.1)
    m._scaling = torch.nn.Parameter(torch.log(torch.rand(P, 3, device="cuda", generator=g) * 0.01 + 0.002))
    m._rotation = torch.nn.Parameter(torch.randn(P, 4, device="cuda", generator=g))       # not unit: normalisation matters
    m._opacity = torch.nn.Parameter(torch.randn(P, 1, device="cuda", generator=g))
    m._mask = torch.nn.Parameter(torch.randn(P, 1, device="cuda", generator=g))
    m.active_sh_degree = 1
    for name in ("max_radii2D",):
        setattr(m, name, torch.zeros(P, device="cuda"))
    m.xyz_gradient_accum, m.denom = torch.zeros(P, 1, device="cuda"), torch.zeros(P, 1, device="cuda")
    opt = OptimizationParams()
    bg = torch.zeros(3, device="cuda")
    params = [m._xyz, m._scaling, m._rotation, m._opacity, m._mask, m._features_dc, m._features_rest]
    fused = FusedCloudStep(m, ViewTable(cams), opt, bg)
    for ci, unit in ((1, False), (3, True)):
        cam = cams[ci]
        for p in params:
            p.grad = None
        for t in (m.max_radii2D, m.xyz_gradient_accum, m.denom):
            t.zero_()
        loss, terms, pkg = loss_function_single_pass(m, cam, opt, bg)
        loss.backward()
        with torch.no_grad():
            m.update_densification_stats(pkg["viewspace_points"], pkg["radii"], pkg["visibility_filter"])
        ref = dict(loss=float(loss), terms={k: float(v) for k, v in terms.items()}, grads=[p.grad.clone() for p in params],
                   stats=[t.clone() for t in (m.max_radii2D, m.xyz_gradient_accum, m.denom)])
        for p in params:
            p.grad = None
        for t in (m.max_radii2D, m.xyz_gradient_accum, m.denom):
            t.zero_()
        fused.views.select(ci)
        floss, _ = fused.loss()
        fused.backward(floss) if unit else floss.backward()
        fused.update_densification_stats()
        assert abs(float(floss) - ref["loss"]) <= 2e-5 * abs(ref["loss"]), (float(floss), ref["loss"])
        fterms = {k: float(v) for k, v in fused.terms().items()}
        for k, v in ref["terms"].items():
            assert abs(fterms[k] - v) <= 2e-5 * max(abs(v), 1e-3), (k, fterms[k], v)
        for name, p, gref in zip(("xyz", "scaling", "rotation", "opacity", "mask", "f_dc", "f_rest"), params, ref["grads"]):
            scale = float(gref.abs().max())
            # expf / the reciprocal norm differ from torch.exp / F.normalize by an ulp in the rasterizer's INPUTS, which
            # moves a few alpha >= 1/255 decisions at footprint edges: 1e-3 of the tensor's scale at the worst element,
            # 1e-5 in the mean
            d = (p.grad - gref).abs()
            assert d.max() <= 1e-3 * max(scale, 1e-12), (name, float(d.max()), scale)
            assert d.mean() <= 1e-5 * max(scale, 1e-12), (name, float(d.mean()), scale)
        for t, sref in zip((m.max_radii2D, m.xyz_gradient_accum, m.denom), ref["stats"]):
            assert (t - sref).abs().max() <= 1e-3 * max(float(sref.abs().max()), 1e-12)
        assert float(m.denom.sum()) > 0


def test_fused_iteration_without_view_masks():
    """Views without masks: no BCE term, the orientation term masks by any(direction != background); the fused head then
    takes its two-pass route (the mask count depends on the render).  Loss and gradients against the op-by-op path."""
    from arguments import OptimizationParams
    from hgs_runtime.strand_step import FusedStrandStep
    from loss.losses import loss_function_single_pass
    from synthetic import build_workload
    from utils.general import safe_state
    safe_state(True)
    model, cams, _ = build_workload("tiny", device="cuda", with_targets=True)
    for c in cams:
        c.mask = None
        c.float_mask = None
    opt = OptimizationParams()
    model.training_setup(opt)
    bg = torch.zeros(3, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(7)
    with torch.no_grad():
        model._features_dc.add_(0.2 * torch.randn(model._features_dc.shape, device="cuda", generator=g))
        model._endpoints.add_(0.003 * torch.randn(model._endpoints.shape, device="cuda", generator=g))
    params = [model._endpoints, model._width, model._opacity, model._mask, model._features_dc]
    fused = FusedStrandStep(model, cams, opt, bg)
    assert not fused.views.has_mask and fused.head.lambda_mask == 0.0
    cam = cams[1]
    loss, terms, _ = loss_function_single_pass(model, cam, opt, bg)
    assert "mask" not in terms
    loss.backward()
    ref = [p.grad.clone() for p in params]
    for p in params:
        p.grad = None
    fused.views.select(1)
    floss, _ = fused.loss()
    fused.backward(floss)
    assert abs(float(floss) - float(loss)) <= 2e-5 * abs(float(loss))
    assert abs(float(fused.terms()["orientation"]) - float(terms["orientation"])) <= 2e-5 * abs(float(terms["orientation"]))
    for p, gref in zip(params, ref):
        if gref.abs().max() == 0:
            assert p.grad.abs().max() == 0
            continue
        assert (p.grad - gref).abs().max() <= 3e-4 * float(gref.abs().max())


@pytest.mark.parametrize("workload", ["tiny", "north_star"])
def test_fused_iteration_is_bitwise_reproducible(workload):
    """No float atomics anywhere in the fused strand iteration (blend backward: fixed-order partial sums; endpoints:
    gather over the adjacency tables): two evaluations from the same state give bit-identical loss and gradients
    (north_star: at the BASELINE size, 100 000 strand-Gaussians at 1080p)."""
    from arguments import OptimizationParams
    from hgs_runtime.strand_step import FusedStrandStep
    from synthetic import build_workload
    from utils.general import safe_state
    safe_state(True)
    model, cams, _ = (build_workload("tiny", device="cuda", with_targets=True) if workload == "tiny"
                      else build_workload(workload, device="cuda", with_targets=True, n_views=3))
    opt = OptimizationParams()
    model.training_setup(opt)
    with torch.no_grad():
        model._endpoints.add_(0.002 * torch.randn_like(model._endpoints))
    fused = FusedStrandStep(model, cams, opt, torch.zeros(3, device="cuda"))
    assert fused.ep_segments is not None and fused.ep_pairs is not None
    deg = (fused.ep_segments >= 0).sum(dim=1)
    assert int(deg.min()) >= 1 and int(deg.max()) == 2
    params = [model._endpoints, model._width, model._opacity, model._mask, model._features_dc]
    runs = []
    for _ in range(2):
        for p in params:
            p.grad = None
        fused.views.select(2)
        loss, _ = fused.loss()
        fused.backward(loss)
        runs.append((loss.detach().clone(), [p.grad.clone() for p in params]))
    assert torch.equal(runs[0][0], runs[1][0])
    for a, b in zip(runs[0][1], runs[1][1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("workload", ["tiny", "north_star"])
def test_unread_image_gradient_blocks_change_no_parameter_gradient(workload):
    """HgsHeadParams.tile_used: the SSIM backward neither filters nor zero-fills the 32 x 32 blocks of dL/dimage whose
    tiles the blend backward never reads (no pixel there blended an entry).  Loss terms, every parameter gradient and the
    densification statistics are bit-identical with and without; the number of filtered blocks drops."""
    from arguments import OptimizationParams
    from hgs_runtime.strand_step import FusedStrandStep
    from synthetic import build_workload
    from utils.general import safe_state
    safe_state(True)
    model, cams, _ = (build_workload("tiny", device="cuda", with_targets=True) if workload == "tiny"
                      else build_workload(workload, device="cuda", with_targets=True, n_views=2))
    opt = OptimizationParams()
    model.training_setup(opt)
    with torch.no_grad():
        model._endpoints.add_(0.002 * torch.randn_like(model._endpoints))
    fused = FusedStrandStep(model, cams, opt, torch.zeros(3, device="cuda"))
    params = [model._endpoints, model._width, model._opacity, model._mask, model._features_dc]
    runs = {}
    fused.poison_unwritten = True     # dL/dimage starts as NaN: a block left unwritten and read anyway would show
    for skip in (False, True):
        fused.skip_unread_blocks = skip
        for p in params:
            p.grad = None
        for t in (model.max_radii2D, model.xyz_gradient_accum, model.denom):
            t.zero_()
        fused.views.select(1)
        loss, terms = fused.loss()
        fused.backward(loss)
        fused.update_densification_stats()
        # the block lists sit behind the zero flags in the head's scratch: [n_work, n_zero_fill, ...]
        runs[skip] = (loss.detach().clone(), terms.clone(), [p.grad.clone() for p in params],
                      [model.max_radii2D.clone(), model.xyz_gradient_accum.clone(), model.denom.clone()])
    assert all(bool(torch.isfinite(g_).all()) for g_ in runs[True][2])
    n_terms = 14     # (hgs.h HGS_HEAD_TOTAL_FWD + 1: the two words behind are padding nobody writes)
    assert torch.equal(runs[False][0], runs[True][0]) and torch.equal(runs[False][1][:n_terms], runs[True][1][:n_terms])
    for a, b in zip(runs[False][2] + runs[False][3], runs[True][2] + runs[True][3]):
        assert torch.equal(a, b)


def test_black_background_backward_specialisation_is_exact():
    """hgs_backward_multi(bg = NULL) -- the blend backward with the background terms compiled out, what the fused iteration
    passes for a black background -- gives the gradients of bg = (0, ..., 0), bit for bit."""
    from arguments import OptimizationParams
    from hgs_runtime.strand_step import FusedStrandStep
    from synthetic import build_workload
    model, cams, _ = build_workload("tiny", device="cuda", with_targets=True)
    opt = OptimizationParams()
    model.training_setup(opt)
    fused = FusedStrandStep(model, cams, opt, torch.zeros(3, device="cuda"))
    assert fused.bg7_backward is None
    grey = FusedStrandStep(model, cams, opt, torch.full((3,), 0.25, device="cuda"))
    assert grey.bg7_backward is grey.bg7                     # (only an all-zero background takes the specialisation)
    params = [model._endpoints, model._width, model._opacity, model._mask, model._features_dc]
    runs = []
    for bgb in (None, fused.bg7):
        fused.bg7_backward = bgb
        for p in params:
            p.grad = None
        fused.views.select(2)
        loss, _ = fused.loss()
        fused.backward(loss)
        runs.append([p.grad.clone() for p in params] + [fused.last["dmean2D"].clone()])
    for a, b in zip(*runs):
        assert torch.equal(a, b) and bool(a.abs().sum() > 0)


def test_op_by_op_black_background_is_the_callers_statement():
    """render_multi / loss_function_single_pass: the black-background backward runs only when the CALLER says the
    background is black (nothing is cached per tensor: a new background tensor that reuses a freed one's address, or one
    rewritten in place, is the one rendered and differentiated); with a black background both backwards agree bit for bit."""
    from arguments import OptimizationParams
    from loss.losses import loss_function_single_pass
    from synthetic import build_workload
    model, cams, _ = build_workload("tiny", device="cuda", with_targets=True)
    opt = OptimizationParams()
    model.training_setup(opt)
    params = [model._endpoints, model._width, model._opacity, model._mask, model._features_dc]

    def grads(bg, black):
        for p in params:
            p.grad = None
        loss, _, pkg = loss_function_single_pass(model, cams[1], opt, bg, black_background=black)
        loss.backward()
        return float(loss), [p.grad.clone() for p in params], pkg["render"].detach().clone()

    zero = torch.zeros(3, device="cuda")
    l0, g0, _ = grads(zero, False)
    l1, g1, _ = grads(zero, True)
    assert l0 == l1 and all(torch.equal(a, b) for a, b in zip(g0, g1))
    # a background rewritten in place between two calls (also through .data, which leaves the version counter alone)
    bg = torch.zeros(3, device="cuda")
    _, _, img_black = grads(bg, False)
    bg.data.fill_(0.6)
    l2, g2, img_grey = grads(bg, False)
    fresh = torch.full((3,), 0.6, device="cuda")
    l3, g3, img_fresh = grads(fresh, False)
    assert torch.equal(img_grey, img_fresh) and not torch.equal(img_grey, img_black)
    assert l2 == l3 and all(torch.equal(a, b) for a, b in zip(g2, g3))
    assert not all(torch.equal(a, b) for a, b in zip(g0, g2))


@pytest.mark.parametrize("use_graph", [False, True])
def test_stage1_training_loop_with_densification(use_graph):
    """training() on the Stage-I cloud through the fused cloud iteration: densification (clone / split / prune) and the
    opacity reset change the number of Gaussians; the optimizer state, the statistics and -- in graph mode -- the
    captured graph follow."""
    from arguments import OptimizationParams
    from synthetic import attach_targets, cameras_extent, make_cameras, make_cloud_model
    from train import training
    from utils.general import safe_state
    safe_state(True)
    cams = make_cameras(4, 200, 120, device="cuda")
    model = make_cloud_model(3000, device="cuda", spatial_lr_scale=cameras_extent(cams))
    attach_targets(cams, model)
    opt = OptimizationParams()
    opt.densify_from_iter, opt.densification_interval, opt.opacity_reset_interval = 2, 5, 12
    opt.densify_grad_threshold = 1e-7            # make the tiny scene densify
    model.training_setup(opt)
    P0 = model.get_xyz.shape[0]
    ema = training(model, cams, opt, iterations=16, extent=cameras_extent(cams), use_graph=use_graph)
    assert torch.isfinite(ema)
    P1 = model.get_xyz.shape[0]
    assert P1 != P0
    for grp in model.optimizer.param_groups:
        p = grp["params"][0]
        st = model.optimizer.state[p]
        assert p.shape[0] == P1
        if p.numel() and "exp_avg" in st:     # (empty f_rest at SH degree 0 is never stepped)
            assert st["exp_avg"].shape == p.shape and st["exp_avg_sq"].shape == p.shape
    assert model.max_radii2D.shape[0] == P1 and model.denom.shape[0] == P1


@pytest.mark.gpu
@pytest.mark.parametrize("hw", [(200, 328), (96, 64), (70, 132)])
def test_loss_head_skips_all_zero_blocks_exactly(hw):
    """The loss head's SSIM kernels skip blocks that are exactly zero in render and target (forward: no filter passes;
    backward: blocks whose 3x3 block neighbourhood is zero are only zero-filled, the rest is split evenly over the XCDs
    through a compacted list).  The result has to be what the stand-alone hgs_ssim_l1_* kernels (which walk every block)
    produce, and the block lists have to partition the frame."""
    import ctypes as C
    import numpy as np
    import hgs_runtime as rt
    from arguments import OptimizationParams
    from hgs_runtime.fused import ssim_l1
    from hgs_runtime.strand_step import head_params
    H, W = hw
    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(11)
    rnd = lambda *s: torch.rand(*s, device=dev, generator=g)
    image, gt = torch.zeros(3, H, W, device=dev), torch.zeros(3, H, W, device=dev)
    y0, y1, x0, x1 = H // 3, H // 3 + H // 4, W // 4, W // 4 + W // 5
    image[:, y0:y1, x0:x1] = rnd(3, y1 - y0, x1 - x0)
    gt[:, y0 + 2:y1 + 2, x0 - 3:x1 - 3] = rnd(3, y1 - y0, x1 - x0)
    mask_img, omap = rnd(H, W) * 4 - 2, rnd(3, H, W)
    fmask, ori, conf = (rnd(H, W) > 0.5).float(), rnd(H, W) * 3.14159, rnd(H, W)
    m8 = (rnd(H, W) > 0.3).to(torch.uint8)
    row = rt.ViewTargets()
    row.image, row.float_mask, row.orientation, row.confidence, row.mask = (t.data_ptr() for t in (gt, fmask, ori, conf, m8))
    for k, v in enumerate(np.eye(4, dtype=np.float32).reshape(-1)):
        row.viewmatrix[k] = row.projmatrix[k] = float(v)
    row.mask_count = float(m8.sum().item())
    targets = torch.from_numpy(np.frombuffer(bytes(row), dtype=np.uint8).copy()).to(dev)
    opt = OptimizationParams()
    hp = head_params(H, W, opt, 0, 0, 1e-6, True)
    L = rt.lib()
    scratch = torch.empty(L.hgs_loss_head_scratch_floats(C.byref(hp)), device=dev)
    out = torch.zeros(rt.HEAD_NOUT, device=dev)
    d_img, d_mask, d_omap = torch.full((3, H, W), 7.0, device=dev), torch.empty(H, W, device=dev), torch.empty(3, H, W, device=dev)
    one = torch.ones(1, device=dev)
    rt.check(L.hgs_loss_head_forward(rt.current_stream(), C.byref(hp), image.data_ptr(), mask_img.data_ptr(), omap.data_ptr(),
                                     targets.data_ptr(), None, None, scratch.data_ptr(), out.data_ptr(), None, None))
    rt.check(L.hgs_loss_head_backward(rt.current_stream(), C.byref(hp), image.data_ptr(), mask_img.data_ptr(), omap.data_ptr(),
                                      targets.data_ptr(), None, None, scratch.data_ptr(), out.data_ptr(), one.data_ptr(), 0,
                                      d_img.data_ptr(), d_mask.data_ptr(), d_omap.data_ptr(), None))
    # stand-alone kernels: every block is filtered
    a = image.clone().requires_grad_(True)
    s, l1 = ssim_l1(a, gt)
    ((1.0 - opt.lambda_dssim) * l1 + opt.lambda_dssim * (1.0 - s)).backward()
    o = dict(zip(rt.HEAD_OUT, out.tolist()))
    assert abs(o["l1"] - float(l1)) <= 1e-6 * max(float(l1), 1e-6)
    assert abs(o["dssim"] - float(1.0 - s)) <= 2e-6
    assert torch.equal(d_img, a.grad)
    # block lists (white box: [n_work, n_skip, -, -][work ids][skipped ids] close the scratch buffer)
    nbs = 3 * ((H + 31) // 32) * ((W + 31) // 32)
    lists = scratch.view(torch.int32)[scratch.numel() - (2 * nbs + 16):].cpu()     # (4 header words, 2 nbs ids, 12 spare)
    n_work, n_skip = int(lists[0]), int(lists[1])
    assert n_work + n_skip == nbs
    work, skipped = lists[4:4 + n_work], lists[4 + nbs:4 + nbs + n_skip]
    assert sorted(work.tolist() + skipped.tolist()) == list(range(nbs))
    assert work.tolist() == sorted(work.tolist())
    if (H, W) == (200, 328):
        assert n_skip > 0 and n_work > 0
    # skipped blocks carry no gradient; every block with gradient is on the work list
    gb = torch.nn.functional.max_pool2d(torch.nn.functional.pad((a.grad != 0).float(), (0, (-W) % 32, 0, (-H) % 32))[None], 32)[0]
    assert not gb.reshape(-1)[skipped.long()].any()
    # ---- with the consumer's tile hint (HgsHeadParams.tile_used): blocks none of whose 16 x 16 tiles is read are left alone
    # (neither filtered nor zero-filled), the others are exactly as before.  Odd tile counts per row / column included
    # (W, H not multiples of 32: the last block of a row / column has one tile, not two).
    tx_n, ty_n = (W + 15) // 16, (H + 15) // 16
    used = (torch.rand(ty_n, tx_n, device=dev, generator=g) > 0.5)
    used[:, -1] = True                                   # (the odd last column / row matter)
    used[-1, :] = ~used[-1, :]
    tile_used = (used.to(torch.int32) * 5).contiguous()
    hp.tile_used, hp.tiles_x, hp.tiles_y = tile_used.data_ptr(), tx_n, ty_n
    d2 = torch.full((3, H, W), 7.0, device=dev)
    out2 = torch.zeros(rt.HEAD_NOUT, device=dev)
    rt.check(L.hgs_loss_head_forward(rt.current_stream(), C.byref(hp), image.data_ptr(), mask_img.data_ptr(), omap.data_ptr(),
                                     targets.data_ptr(), None, None, scratch.data_ptr(), out2.data_ptr(), None, None))
    rt.check(L.hgs_loss_head_backward(rt.current_stream(), C.byref(hp), image.data_ptr(), mask_img.data_ptr(), omap.data_ptr(),
                                      targets.data_ptr(), None, None, scratch.data_ptr(), out2.data_ptr(), one.data_ptr(), 0,
                                      d2.data_ptr(), d_mask.data_ptr(), d_omap.data_ptr(), None))
    assert torch.equal(out2[:14], out[:14])
    # a block is read iff one of its 2 x 2 tiles is
    bu = torch.nn.functional.max_pool2d(torch.nn.functional.pad(used.float(), (0, tx_n % 2, 0, ty_n % 2))[None, None], 2)[0, 0] > 0
    px_used = bu.repeat_interleave(32, dim=0).repeat_interleave(32, dim=1)[:H, :W]
    assert torch.equal(d2[:, px_used], a.grad[:, px_used])            # read blocks: the full result (zero-filled where it is zero)
    assert bool((d2[:, ~px_used] == 7.0).all())                        # the others: untouched
    lists2 = scratch.view(torch.int32)[scratch.numel() - (2 * nbs + 16):].cpu()
    read_blocks = int(bu.sum()) * 3
    assert int(lists2[0]) + int(lists2[1]) == read_blocks and int(lists2[0]) <= n_work


def test_one_launch_parameters_and_preprocess_equals_two_launches():
    """hgs_hair_forward_preprocess (strand parameters -> Gaussians -> preprocess in ONE kernel, the riders beside it) against
    hgs_hair_params_forward + hgs_forward_preprocess: derived Gaussians, radii, the geometry buffer, the image and every
    gradient bit for bit (the smoothness partial sums ride in a kernel of another translation unit: that term to 1e-6)."""
    from arguments import OptimizationParams
    from diff_gaussian_rasterization import _C as raster
    from hgs_runtime.strand_step import FusedStrandStep, ViewTable
    from synthetic import build_workload
    model, cams, _ = build_workload("tiny", device="cuda", with_targets=True)
    opt = OptimizationParams()
    model.training_setup(opt)
    params = [model._endpoints, model._width, model._opacity, model._mask, model._features_dc]
    runs = {}
    try:
        raster.set_async(True, slack=2.0)
        for fuse in (False, True):
            views = ViewTable(cams)
            step = FusedStrandStep(model, views, opt, torch.zeros(3, device="cuda"))
            step.fuse_preprocess = fuse
            seen = []
            for it, v in enumerate([0, 2, 1, 3, 2]):
                for p in params:
                    p.grad = None
                views.prologue(v, ride=True)
                fused_now = fuse and raster.will_fuse_hair(views.W, views.H)
                loss, terms = step.loss()
                step.backward(loss)
                raster.check_async()
                seen.append((fused_now, views.counts_clean, float(loss.detach()), terms.clone(), step.last["planes"].clone(),
                             step.last["radii"].clone(), [p.grad.clone() for p in params]))
            runs[fuse] = seen
    finally:
        raster.set_async(False)
    assert all(s[0] for s in runs[True][1:]) and not any(s[0] for s in runs[False])   # (a first pass may be the blocking one that learns the capacity)
    assert all(s[1] for s in runs[True][1:])            # capacity-mode passes leave the tile counters at zero
    for a, b in zip(runs[False], runs[True]):
        assert torch.equal(a[4], b[4]) and torch.equal(a[5], b[5])
        assert abs(a[2] - b[2]) <= 1e-6 * abs(a[2])
        for ga, gb in zip(a[6], b[6]):
            assert torch.equal(ga, gb) or float((ga - gb).abs().max()) <= 1e-6 * float(ga.abs().max())
        assert bool(a[6][0].abs().sum() > 0)
    # without the smoothness term nothing is left that could differ
    opt2 = OptimizationParams()
    opt2.lambda_smooth = 0.0
    out = {}
    try:
        raster.set_async(True, slack=2.0)
        for fuse in (False, True):
            views = ViewTable(cams)
            step = FusedStrandStep(model, views, opt2, torch.zeros(3, device="cuda"))
            step.fuse_preprocess = fuse
            for v in (1, 3):
                for p in params:
                    p.grad = None
                views.prologue(v, ride=True)
                loss, terms = step.loss()
                step.backward(loss)
            raster.check_async()
            out[fuse] = [loss.detach().clone(), terms[:14].clone()] + [p.grad.clone() for p in params]
    finally:
        raster.set_async(False)
    for a, b in zip(out[False], out[True]):
        assert torch.equal(a, b)


def test_one_launch_cloud_parameters_and_preprocess_equals_two_launches():
    """hgs_cloud_forward_preprocess against hgs_cloud_params_forward + hgs_forward_preprocess (Stage-I cloud iteration): loss
    terms, image, radii and every parameter gradient bit for bit."""
    from arguments import OptimizationParams
    from diff_gaussian_rasterization import _C as raster
    from hgs_runtime.strand_step import FusedCloudStep, ViewTable
    from synthetic import attach_targets, cameras_extent, make_cameras, make_cloud_model
    cams = make_cameras(4, 200, 120, device="cuda")
    model = make_cloud_model(3000, device="cuda", spatial_lr_scale=cameras_extent(cams))
    attach_targets(cams, model)
    opt = OptimizationParams()
    model.training_setup(opt)
    params = [model._xyz, model._scaling, model._rotation, model._opacity, model._mask, model._features_dc]
    out = {}
    try:
        raster.set_async(True, slack=2.0)
        for fuse in (False, True):
            views = ViewTable(cams)
            step = FusedCloudStep(model, views, opt, torch.zeros(3, device="cuda"))
            step.fuse_preprocess = fuse
            seen = []
            for v in (0, 2, 1, 3):
                for p in params:
                    p.grad = None
                views.prologue(v, ride=True)
                fused_now = fuse and raster.will_fuse_hair(views.W, views.H)
                loss, terms = step.loss()
                step.backward(loss)
                raster.check_async()
                seen.append([fused_now, loss.detach().clone(), terms[:14].clone(), step.last["planes"].clone(),
                             step.last["radii"].clone()] + [p.grad.clone() for p in params])
            out[fuse] = seen
    finally:
        raster.set_async(False)
    assert all(s[0] for s in out[True][1:]) and not any(s[0] for s in out[False])
    for a, b in zip(out[False], out[True]):
        for x, y in zip(a[1:], b[1:]):
            assert torch.equal(x, y)
        assert bool(a[5].abs().sum() > 0)


def test_fused_parameter_backward_equals_two_launches():
    """hgs_backward_multi_params + hgs_hair_endpoint_gather (round 5: the segment geometry's backward applied in the rasterizer
    backward's per-Gaussian lanes, endpoint contributions instead of 124 bytes of per-Gaussian gradients) against
    hgs_backward_multi + hgs_hair_params_backward: every parameter gradient, the densification statistics, the RGB-only
    screen-space gradient and the loss terms BIT FOR BIT (the shared parameter arithmetic is evaluated without contraction in
    both translation units, hgs_strand_bwd.h) -- with the deferred head tail riding in the gather launch and without, with and
    without the smoothness term."""
    from arguments import OptimizationParams
    from hgs_runtime.strand_step import FusedStrandStep, ViewTable
    from synthetic import build_workload
    model, cams, _ = build_workload("tiny", device="cuda", with_targets=True)
    params = [model._endpoints, model._width, model._opacity, model._mask, model._features_dc]
    stats = lambda: [model.max_radii2D, model.xyz_gradient_accum, model.denom]
    for lam_smooth, defer in ((None, False), (None, True), (0.0, False)):
        opt = OptimizationParams()
        if lam_smooth is not None:
            opt.lambda_smooth = lam_smooth
        model.training_setup(opt)
        runs = {}
        for fuse in (False, True):
            for t in stats():
                t.zero_()
            views = ViewTable(cams)
            step = FusedStrandStep(model, views, opt, torch.zeros(3, device="cuda"))
            step.fuse_param_backward = fuse
            step.defer_tail = defer
            assert step.ep_segments is not None
            seen = []
            for v in (0, 2, 1, 3):
                for p in params:
                    p.grad = None
                views.prologue(v, ride=True)
                loss, terms = step.loss()
                step.backward(loss)
                assert step.last["stats_done"]
                seen.append([loss.detach().clone(), terms[:14].clone(), step.last["dmean2D"].clone()] +
                            [p.grad.clone() for p in params] + [t.clone() for t in stats()])
            runs[fuse] = seen
        for a, b in zip(runs[False], runs[True]):
            for x, y in zip(a, b):
                assert torch.equal(x, y)
            assert bool(a[3].abs().sum() > 0) and bool(a[-1].sum() > 0)


def test_fused_cloud_parameter_backward_equals_two_launches():
    """hgs_backward_multi_params(HGS_PARAMS_CLOUD) against hgs_backward_multi + hgs_cloud_params_backward (Stage-I cloud): the
    launch of the parameters' backward is gone, its results -- gradients of all seven parameter groups, statistics, loss terms
    (the deferred tail rides in the rasterizer backward's spare workgroup) -- bit for bit."""
    from arguments import OptimizationParams
    from hgs_runtime.strand_step import FusedCloudStep, ViewTable
    from synthetic import attach_targets, cameras_extent, make_cameras, make_cloud_model
    cams = make_cameras(4, 200, 120, device="cuda")
    model = make_cloud_model(3000, device="cuda", spatial_lr_scale=cameras_extent(cams))
    attach_targets(cams, model)
    opt = OptimizationParams()
    model.training_setup(opt)
    params = [model._xyz, model._scaling, model._rotation, model._opacity, model._mask, model._features_dc]
    stats = lambda: [model.max_radii2D, model.xyz_gradient_accum, model.denom]
    for defer in (False, True):
        out = {}
        for fuse in (False, True):
            for t in stats():
                t.zero_()
            views = ViewTable(cams)
            step = FusedCloudStep(model, views, opt, torch.zeros(3, device="cuda"))
            step.fuse_param_backward = fuse
            step.defer_tail = defer
            seen = []
            for v in (0, 2, 1, 3):
                for p in params:
                    p.grad = None
                views.prologue(v, ride=True)
                loss, terms = step.loss()
                step.backward(loss)
                seen.append([loss.detach().clone(), terms[:14].clone(), step.last["dmean2D"].clone()] +
                            [p.grad.clone() for p in params] + [t.clone() for t in stats()])
            out[fuse] = seen
        for a, b in zip(out[False], out[True]):
            for x, y in zip(a, b):
                assert torch.equal(x, y)
            assert bool(a[3].abs().sum() > 0)


def test_densification_inputs_of_the_fused_pass_equal_the_three_pass_form():
    """The one input of densification() that no reference run pins: xyz_gradient_accum / denom /
    max_radii2D as the fused iteration accumulates them -- from the RGB-only moments of the single 7-channel pass, in the lanes of
    the rasterizer backward (hgs_backward_multi_params) -- against the reference's structure: render() (the RGB pass and ITS
    screen-space gradient), loss_function with its two more passes, update_densification_stats (scene/hair_gaussian_model.py:
    1401-1408, train.py:170).  Both forms see the same parameters in every one of 40 training iterations (the fused iteration's
    Adam step moves them); the accumulated statistics agree to 1e-5 of their scale and select the same segments at the clone /
    split threshold.  (tools/three_stage.py runs the same check on a 100 k-segment Stage-II model: profiles/r05_three_stage.json.)"""
    import copy
    from arguments import OptimizationParams
    from gaussian_renderer import render
    from hgs_runtime.strand_step import ViewTable, fused_step_for
    from loss.losses import loss_function
    from synthetic import build_workload
    from train import ViewSampler, training_step
    model, cams, extent = build_workload("tiny", device="cuda", with_targets=True)
    opt = OptimizationParams()
    opt.enable_topology = False
    model.training_setup(opt)
    bg = torch.zeros(3, device="cuda")
    fused = fused_step_for(model, ViewTable(cams), opt, bg)
    stats = (model.max_radii2D, model.xyz_gradient_accum, model.denom)
    acc3 = [torch.zeros_like(t) for t in stats]
    sampler = ViewSampler(cams, seed=7)
    for it in range(1, 41):
        cam = sampler.next()
        model.optimizer.zero_grad(set_to_none=True)
        pkg = render(cam, model, bg)
        loss, _ = loss_function(model, pkg["render"], cam, opt)
        loss.backward()
        with torch.no_grad():
            mine = [t.clone() for t in stats]
            for t, a in zip(stats, acc3):
                t.copy_(a)
            model.update_densification_stats(pkg["viewspace_points"], pkg["radii"], pkg["visibility_filter"])
            for t, a, f in zip(stats, acc3, mine):
                a.copy_(t)
                t.copy_(f)
        model.optimizer.zero_grad(set_to_none=True)
        training_step(model, cam, opt, bg, it, extent=extent, fused=fused)
        assert fused.last["stats_done"]
    for name, a, b in zip(("max_radii2D", "xyz_gradient_accum", "denom"), stats, acc3):
        assert float(b.abs().max()) > 0
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()), name
    seen = acc3[2].reshape(-1) > 0
    gf = (stats[1].reshape(-1)[seen] / stats[2].reshape(-1)[seen])
    g3 = (acc3[1].reshape(-1)[seen] / acc3[2].reshape(-1)[seen])
    thr = float(opt.densify_grad_threshold)
    near = (g3 - thr).abs() <= 1e-5 * thr             # (a segment exactly at the threshold may fall either way)
    assert bool(((gf >= thr) == (g3 >= thr))[~near].all()) and int(seen.sum()) > 100


@pytest.mark.parametrize("kind", ["strands", "cloud"])
def test_adam_in_the_backward_lanes_equals_the_adam_launch(kind):
    """Round 5: a single-rank captured iteration holds no optimizer launch -- the lanes of the backward that finish a parameter
    element's gradient apply its Adam update (include/hgs.h HgsAdamSlot), the prologue advances the step counters and forms
    the bias-correction coefficients.  Against the same iterations with hgs_adam_step as a launch of its own: parameters, both
    moments, step counters and statistics bit for bit -- replayed (one and four steps per graph launch) and eager."""
    from arguments import OptimizationParams
    from synthetic import attach_targets, build_workload, cameras_extent, make_cameras, make_cloud_model
    from train import GraphedStep, ViewSampler, training_step
    from hgs_runtime.strand_step import ViewTable, fused_step_for
    from diff_gaussian_rasterization import _C as raster
    bg = torch.zeros(3, device="cuda")

    def fresh():
        if kind == "strands":
            model, cams, extent = build_workload("tiny", device="cuda", with_targets=True)
        else:
            cams = make_cameras(4, 200, 120, device="cuda")
            extent = cameras_extent(cams)
            model = make_cloud_model(3000, device="cuda", spatial_lr_scale=extent)
            attach_targets(cams, model)
        opt = OptimizationParams()
        opt.enable_topology = False
        model.training_setup(opt)
        return model, cams, extent, opt

    def state(model):
        out = []
        for g_ in model.optimizer.param_groups:
            for p in g_["params"]:
                st = model.optimizer.state.get(p, {})
                out += [p.detach().clone()] + [st[k].clone() for k in ("exp_avg", "exp_avg_sq", "step") if k in st]
        return out + [model.max_radii2D.clone(), model.xyz_gradient_accum.clone(), model.denom.clone()]

    runs = {}
    try:
        for inline in (False, True):
            model, cams, extent, opt = fresh()
            opt.inline_adam = inline
            sampler = ViewSampler(cams, seed=3)
            gs = GraphedStep(model, cams, opt, bg, extent=extent, steps_per_graph=4)
            gs.capture(cams, iteration=1)
            assert gs.inline_adam is inline
            it = 0
            for _ in range(3):
                it += 1
                gs.step(sampler.next(), it)
            gs.step_many([sampler.next() for _ in range(4)], it + 1)
            it += 4
            gs.check()
            raster.set_async(False)
            snap = state(model)
            # eager continuation (topology iterations of training() run this way -- with the launch; here also in-lane)
            views = ViewTable(cams)
            fused = fused_step_for(model, views, opt, bg)
            assert fused.enable_inline_adam(inline) is inline
            for _ in range(3):
                it += 1
                training_step(model, sampler.next(), opt, bg, it, extent=extent, fused=fused)
            torch.cuda.synchronize()
            runs[inline] = (snap, state(model))
    finally:
        raster.set_async(False)
    for a, b in zip(runs[False], runs[True]):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    steps = sorted(float(t) for t in runs[True][1] if t.ndim == 0)      # (an empty tensor -- no higher SH coefficients -- never steps)
    assert steps and steps[-1] == 10.0 and set(steps) <= {0.0, 10.0} and steps.count(10.0) >= 5


def test_device_smoothness_pairs_equal_the_native_filter():
    """smoothness_index_pairs() of a GPU model builds the table of consecutive strand segments on the device, from the strand
    tables compute_strands_info / sort_spatially keep there; c_utils.filter_strand_segments_flat (the native restatement of the
    reference's Cython helper, pinned against its build) on the host tables gives the same pairs in the same order -- on the
    initial strands, after densification + merging (clones, splits, background segments) and after the storage sort."""
    from arguments import OptimizationParams
    from c_utils import filter_strand_segments_flat
    from synthetic import build_workload
    from train import training
    model, cams, extent = build_workload("tiny", device="cuda", with_targets=True)
    opt = OptimizationParams()
    opt.densify_from_iter, opt.densification_interval, opt.merge_interval, opt.densify_grad_threshold = 3, 6, 8, 1e-7
    model.training_setup(opt)

    def check():
        model._smooth_pairs = None
        got = model.smoothness_index_pairs()
        assert got.is_cuda and got.dtype == torch.long
        want = np.asarray(filter_strand_segments_flat(*model.strands_info.flat)).reshape(-1, 2, 2)
        assert np.array_equal(got.cpu().numpy(), want) and want.shape[0] > 0

    check()
    training(model, cams, opt, iterations=26, extent=extent)
    assert model.get_xyz.shape[0] != 1200          # the operators did change the model
    model.compute_strands_info()
    check()
    model.sort_spatially()
    check()


def test_replays_after_a_blocking_pass_on_the_same_views():
    """The captured step's first launch counts into the image buffer's tile counters beside the workgroups that clear the other
    counters, so it needs them at zero -- which capacity-mode passes leave behind and a blocking-mode pass does not.  A blocking
    pass over the same ViewTable between two replays is noticed (ViewTable.counts_clean) and repaired: same parameters as
    without it, bit for bit."""
    from arguments import OptimizationParams
    from diff_gaussian_rasterization import _C as raster
    from hgs_runtime.strand_step import ViewTable, fused_step_for
    from synthetic import build_workload
    from train import GraphedStep
    from utils.general import safe_state
    res = {}
    try:
        for disturb in (False, True):
            safe_state(True)
            model, cams, extent = build_workload("tiny", device="cuda", with_targets=True)
            opt = OptimizationParams()
            opt.enable_topology = False
            model.training_setup(opt)
            bg = torch.zeros(3, device="cuda")
            views = ViewTable(cams)
            gs = GraphedStep(model, cams, opt, bg, extent=extent, views=views)
            gs.capture(cams)
            assert views.counts_clean and gs._binding[2]            # the rider variant was captured
            for it, ci in enumerate([1, 3, 0], 1):
                gs.step(cams[ci], it)
            if disturb:
                torch.cuda.synchronize()
                raster._state["async"] = False                      # a blocking pass through the same table and image buffer
                probe = fused_step_for(model, views, opt, bg)
                views.prologue(2, ride=True)
                with torch.no_grad():
                    probe.loss()
                raster._state["async"] = True
                assert not views.counts_clean
            for it, ci in enumerate([2, 1, 0], 4):
                gs.step(cams[ci], it)
            gs.check()
            assert views.counts_clean
            res[disturb] = [g["params"][0].detach().clone() for g in model.optimizer.param_groups]
            raster.set_async(False)
    finally:
        raster.set_async(False)
    for a, b in zip(res[False], res[True]):
        assert torch.equal(a, b)


def test_training_falls_back_when_the_views_are_not_uniform(capsys):
    """A capture whose views differ in size cannot go into the view table of the fused iteration (nor into a captured graph,
    whose launches bake the image size in): training() says so and runs the op-by-op iteration eagerly on every camera as it is."""
    from arguments import OptimizationParams
    from synthetic import attach_targets, cameras_extent, make_cameras, make_strand_model
    from train import training
    from utils.general import safe_state
    safe_state(True)
    cams = make_cameras(2, 160, 96, device="cuda") + make_cameras(2, 128, 80, device="cuda")
    model = make_strand_model(n_strands=40, n_seg=10, seed=1, device="cuda", spatial_lr_scale=cameras_extent(cams))
    model.compute_strands_info(only_foreground=True)
    attach_targets(cams[:2], model)
    attach_targets(cams[2:], model)
    opt = OptimizationParams()
    opt.enable_topology = False
    model.training_setup(opt)
    before = model._endpoints.detach().clone()
    ema = training(model, cams, opt, iterations=6, extent=cameras_extent(cams))
    assert torch.isfinite(ema) and not torch.equal(before, model._endpoints.detach())
    assert "op-by-op iteration eagerly" in capsys.readouterr().out


@pytest.mark.parametrize("kind", ["cloud", "hair"])
def test_render_with_python_side_sh_and_covariance(kind):
    """render(convert_SHs_python=True, compute_cov3D_python=True) (reference gaussian_renderer/__init__.py:70-108: colours from
    utils.sh.eval_sh, 3D covariances from the model's get_covariance, both handed to the rasterizer precomputed) gives the image of
    the default call, and its gradients reach the same parameters."""
    from gaussian_renderer import render
    from synthetic import build_workload, cameras_extent, make_cameras, make_cloud_model
    if kind == "hair":
        model, cams, _ = build_workload("tiny", device="cuda", with_targets=False)
    else:
        cams = make_cameras(3, 200, 120, device="cuda")
        model = make_cloud_model(2000, device="cuda", spatial_lr_scale=cameras_extent(cams))
    bg = torch.tensor([0.2, 0.1, 0.3], device="cuda")
    cam = cams[1]
    ref = render(cam, model, bg)
    for flags in (dict(convert_SHs_python=True), dict(compute_cov3D_python=True), dict(convert_SHs_python=True, compute_cov3D_python=True)):
        out = render(cam, model, bg, **flags)
        assert torch.equal(out["radii"], ref["radii"]), flags
        assert float((out["render"] - ref["render"]).detach().abs().max()) <= 2e-5, flags
    w = torch.randn_like(ref["render"])
    grads = {}
    leaf = model._endpoints if kind == "hair" else model._xyz
    for name, flags in (("default", {}), ("python", dict(convert_SHs_python=True, compute_cov3D_python=True))):
        for p in (leaf, model._features_dc, model._opacity):
            p.grad = None
        (render(cam, model, bg, **flags)["render"] * w).sum().backward()
        grads[name] = [p.grad.clone() for p in (leaf, model._features_dc, model._opacity)]
    for a, b in zip(grads["default"], grads["python"]):
        scale = float(a.abs().max())
        assert scale > 0 and float((a - b).abs().max()) <= 2e-3 * scale


def test_debug_mode_snapshots_the_inputs_of_a_failing_call(tmp_path, monkeypatch):
    """raster_settings.debug (reference diff_gaussian_rasterization/__init__.py:83-101, 127-140): the call synchronises and gives
    the same image; a failing forward leaves its inputs in snapshot_fw.dump (CPU copies) before the error propagates."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from gaussian_renderer import render
    from synthetic import build_workload
    model, cams, _ = build_workload("tiny", device="cuda", with_targets=False)
    bg = torch.zeros(3, device="cuda")
    with torch.no_grad():
        assert torch.equal(render(cams[0], model, bg, debug=True)["render"], render(cams[0], model, bg)["render"])
    monkeypatch.chdir(tmp_path)
    cam = cams[0]
    import math
    rs = GaussianRasterizationSettings(image_height=int(cam.image_height), image_width=int(cam.image_width),
                                       tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), bg=bg, scale_modifier=1.0,
                                       viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=3,
                                       campos=cam.camera_center, prefiltered=False, debug=True)
    with torch.no_grad(), pytest.raises(Exception):      # degree 3 asked of a model that stores one coefficient
        GaussianRasterizer(rs)(means3D=model.get_xyz, means2D=torch.zeros_like(model.get_xyz), opacities=model.get_opacity,
                               shs=model.get_features, scales=model.get_scaling, rotations=model.get_rotation)
    dump = torch.load(tmp_path / "snapshot_fw.dump", weights_only=False)
    assert len(dump) == 19 and not dump[1].is_cuda and dump[1].shape == model.get_xyz.shape


def test_device_storage_sort_equals_the_host_form():
    """sort_spatially on a GPU model runs on the device (HairGaussianModel._sort_spatially_device); it applies the permutations of
    storage_order() -- the numpy form, which CPU models keep -- and leaves the same renumbered strand tables, parameters, Adam
    moments and statistics.  Checked on a model whose storage the topology operators have scrambled."""
    from arguments import OptimizationParams
    from synthetic import build_workload
    from train import training
    from utils.general import safe_state
    import copy
    safe_state(True)
    model, cams, extent = build_workload("tiny", device="cuda", with_targets=True)
    opt = OptimizationParams()
    opt.spatial_sort = False                          # leave the operators' appended segments where they are
    opt.densify_from_iter, opt.densification_interval, opt.merge_interval, opt.opacity_reset_interval = 3, 6, 8, 12
    model.training_setup(opt)
    training(model, cams, opt, iterations=20, extent=extent)
    model.compute_strands_info()
    ref = copy.deepcopy(model)
    assert model._strands_dev is not None
    dev_perms = model.sort_spatially()                # device form
    ref._strands_dev = None
    host_perms = ref.sort_spatially()                 # numpy form (no device tables: storage_order)
    assert dev_perms is not None and host_perms is not None
    for a, b in zip(dev_perms, host_perms):
        assert np.array_equal(a, b)
    assert torch.equal(model.endpoint_pairs, ref.endpoint_pairs)
    for ga, gb in zip(model.optimizer.param_groups, ref.optimizer.param_groups):
        pa, pb = ga["params"][0], gb["params"][0]
        assert torch.equal(pa, pb)
        if pa.numel():
            assert torch.equal(model.optimizer.state[pa]["exp_avg"], ref.optimizer.state[pb]["exp_avg"])
    for attr in ("xyz_gradient_accum", "denom", "max_radii2D", "strand_root_endpoint_idx"):
        assert torch.equal(getattr(model, attr), getattr(ref, attr)), attr
    for a in ("offsets", "rows", "segment_rows", "id_to_strand_id", "strand_endpoint_id_to_complementary"):
        assert np.array_equal(np.asarray(getattr(model.strands_info, a)), np.asarray(getattr(ref.strands_info, a))), a
    assert model.sort_spatially() is None             # nothing left to move, and the kept device tables say so too
    kept = model.strands_info
    model.compute_strands_info()
    for a in ("offsets", "rows", "id_to_strand_id", "strand_endpoint_id_to_complementary"):
        assert np.array_equal(np.asarray(getattr(kept, a)), np.asarray(getattr(model.strands_info, a))), a


def test_cloud_densification_selects_the_same_rows_as_the_three_pass_form():
    """The Stage-I analogue of test_densification_inputs_of_the_fused_pass_equal_the_three_pass_form:
    through three densification events of a Stage-I run the op-by-op three-pass form -- render() and ITS screen-space gradient,
    loss_function with its two more passes, update_densification_stats (reference train.py:146-171, scene/gaussian_model.py:
    675-682) -- accumulates its statistics beside FusedCloudStep's (RGB-only moments of the single 7-channel pass, in the lanes of
    hgs_backward_multi_params) on the same parameters; in front of every event both select the same rows for densify_and_clone
    and densify_and_split (scene/gaussian_model.py:298-322; rows within 1e-5 of the threshold may fall either way).  The event
    itself then runs from the fused statistics and its counts are logged (training_step(event_log=))."""
    from arguments import OptimizationParams
    from gaussian_renderer import render
    from hgs_runtime.strand_step import ViewTable, fused_step_for
    from loss.losses import loss_function
    from synthetic import build_capture, stage1_cloud
    from train import ViewSampler, training_step
    from utils.general import safe_state
    safe_state(True)
    gt_pts, gt_model, cams, extent = build_capture("tiny_capture", device="cuda")
    model = stage1_cloud(gt_pts, gt_model, extent, device="cuda")
    opt = OptimizationParams()
    opt.densify_from_iter = 100
    model.training_setup(opt)
    bg = torch.zeros(3, device="cuda")
    stats = lambda: (model.max_radii2D, model.xyz_gradient_accum, model.denom)
    sampler = ViewSampler(cams, seed=11)
    events, selected = [], 0
    fused = fused_step_for(model, ViewTable(cams), opt, bg)
    acc3 = [torch.zeros_like(t) for t in stats()]
    for it in range(1, 401):
        cam = sampler.next()
        event = it > opt.densify_from_iter and it % opt.densification_interval == 0
        if event:       # both forms have seen iterations .. it - 1 (this iteration adds one more view to both before it densifies)
            with torch.no_grad():
                thr, dense = float(opt.densify_grad_threshold), float(opt.percent_dense) * extent
                big = torch.max(model.get_scaling, dim=1).values > dense
                sel = []
                for acc, den in ((model.xyz_gradient_accum, model.denom), (acc3[1], acc3[2])):
                    g = acc / den
                    g[g.isnan()] = 0.0
                    hot = torch.norm(g, dim=-1) >= thr
                    sel.append((hot & ~big, hot & big, g.reshape(-1)))
                for name, a, b in zip(("max_radii2D", "xyz_gradient_accum", "denom"), stats(), acc3):
                    assert float((a - b).abs().max()) <= 1e-5 * max(float(b.abs().max()), 1e-30), (it, name)
                near = (sel[1][2] - thr).abs() <= 1e-5 * thr
                differ = ((sel[0][0] != sel[1][0]) | (sel[0][1] != sel[1][1])) & ~near
                assert int(differ.sum()) == 0, (it, int(differ.sum()))
                selected += int(sel[1][0].sum()) + int(sel[1][1].sum())
        model.optimizer.zero_grad(set_to_none=True)
        pkg = render(cam, model, bg)
        loss, _ = loss_function(model, pkg["render"], cam, opt)
        loss.backward()
        with torch.no_grad():
            mine = [t.clone() for t in stats()]
            for t, a in zip(stats(), acc3):
                t.copy_(a)
            model.update_densification_stats(pkg["viewspace_points"], pkg["radii"], pkg["visibility_filter"])
            for t, a, f in zip(stats(), acc3, mine):
                a.copy_(t)
                t.copy_(f)
        model.optimizer.zero_grad(set_to_none=True)
        n_before = model.get_xyz.shape[0]
        training_step(model, cam, opt, bg, it, extent=extent, fused=fused, event_log=events)
        if event:
            assert events and events[-1]["iteration"] == it and events[-1]["primitives_before"] == n_before
            assert events[-1]["primitives_after"] == model.get_xyz.shape[0]
            assert {"clone", "split", "prune_low_opacity", "prune_total"} <= set(events[-1])
            fused = fused_step_for(model, ViewTable(cams), opt, bg)      # (new tensors behind the model)
            acc3 = [torch.zeros_like(t) for t in stats()]
    assert [e["iteration"] for e in events] == [200, 300, 400] and selected > 0


@pytest.mark.parametrize("kind", ["cloud", "strands"])
def test_row_run_counting_in_the_fused_iterations(kind):
    """HGS_COUNT_ROW_RUNS in the captured form of the iteration: the one-launch parameters -> preprocess kernels with the prologue
    riding beside them, on the view table's image buffer whose counters only the kernels themselves keep at zero (the row-run
    marks lie between the tile counters and the range the rider clears: tile_delta_kernel has to leave them at zero).  Six
    iterations over changing views, row runs on against off: loss terms, image planes, radii and every parameter gradient bit
    for bit.  cloud: 3000 Stage-I Gaussians with their scales raised to radii of 50 pixels; strands: the tiny strand model with the extent of its Gaussians
    along the segment raised to ~15 pixels (rectangles of more than 16 tiles)."""
    from arguments import OptimizationParams
    from diff_gaussian_rasterization import _C as raster
    from hgs_runtime.strand_step import FusedCloudStep, FusedStrandStep, ViewTable
    from synthetic import attach_targets, build_workload, cameras_extent, make_cameras, make_cloud_model
    if kind == "cloud":
        cams = make_cameras(4, 200, 120, device="cuda")
        model = make_cloud_model(3000, device="cuda", spatial_lr_scale=cameras_extent(cams))
        with torch.no_grad():
            model._scaling.add_(1.3)
        attach_targets(cams, model)
    else:
        model, cams, _ = build_workload("tiny", device="cuda", with_targets=True)
    opt = OptimizationParams()
    model.training_setup(opt)
    if kind == "strands":
        with torch.no_grad():      # sigma along the segment: 0.06 world units = 15 pixels at these cameras (training_setup sets the factor)
            model.dist_to_scale_factor = float(model.dist_to_scale_factor) * 0.06 / float(model.get_scaling[:, 0].mean())
    params = ([model._xyz, model._scaling, model._rotation, model._opacity, model._mask, model._features_dc] if kind == "cloud"
              else [model._endpoints, model._width, model._opacity, model._mask, model._features_dc])
    out = {}
    was = raster.set_row_runs(False)
    try:
        raster._state["cap"] = 0
        raster.set_async(True, slack=2.0)
        for rows in (False, True):
            raster.set_row_runs(rows)
            views = ViewTable(cams)
            step = (FusedCloudStep if kind == "cloud" else FusedStrandStep)(model, views, opt, torch.zeros(3, device="cuda"))
            seen = []
            for v in (0, 2, 1, 3, 0, 2):
                for p in params:
                    p.grad = None
                views.prologue(v, ride=True)
                fused_now = raster.will_fuse_hair(views.W, views.H)
                loss, terms = step.loss()
                step.backward(loss)
                raster.check_async()
                seen.append([fused_now, views.counts_clean, loss.detach().clone(), terms[:14].clone(), step.last["planes"].clone(),
                             step.last["radii"].clone()] + [p.grad.clone() for p in params])
            out[rows] = seen
    finally:
        raster.set_async(False)
        raster.set_row_runs(was)
    assert all(s[0] and s[1] for s in out[True][1:])        # the one-launch form ran, the tile counters were left at zero
    for a, b in zip(out[False], out[True]):
        for x, y in zip(a[2:], b[2:]):
            assert torch.equal(x, y)
        assert bool(a[6].abs().sum() > 0)
    assert int(out[True][-1][5].max()) > 40                 # radii: rectangles of more than 16 tiles exist


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_device_strand_walk_equals_the_host_walk(seed):
    """hgs_strand_walk_ends / hgs_strand_walk_fill (scene.hair_gaussian_model.walk_chains_device) against walk_chains (numpy: the
    restatement the CPU tests pin to 112 runs of the reference) and walk_chains_torch: the same strands in the same order and
    orientation, the same id -> strand and end -> other-end tables.  Random sets of chains of 1 .. 300 segments with shuffled
    endpoint ids, shuffled rows and randomly reversed rows, closed loops (left out by every form), unused ids; then a table with an
    endpoint of degree 3, which the device form refuses (None: compute_strands_info falls back)."""
    from scene.hair_gaussian_model import walk_chains, walk_chains_device, walk_chains_torch
    rng = np.random.default_rng(seed)
    lens = np.concatenate([rng.integers(1, 300, size=400), np.ones(50, np.int64), rng.integers(2, 12, size=30)])
    n_chain = 450                                            # the last 30 are closed loops
    pairs, nid = [], 0
    for c, L in enumerate(lens):
        ids = np.arange(nid, nid + L + (1 if c < n_chain else 0))
        nid = ids[-1] + 1
        seq = np.stack([ids[:-1], ids[1:]], 1) if c < n_chain else np.stack([ids, np.roll(ids, -1)], 1)
        pairs.append(seq)
    pairs = np.concatenate(pairs)
    n_ep = nid + 17                                          # (ids nobody uses)
    relabel = rng.permutation(n_ep)
    pairs = relabel[pairs]
    swap = rng.random(pairs.shape[0]) < 0.5
    pairs[swap] = pairs[swap][:, ::-1]
    pairs = pairs[rng.permutation(pairs.shape[0])].astype(np.int64)
    pos = rng.normal(size=(n_ep, 3))
    roots = rng.normal(size=(64, 3))

    def end_distance_np(ids):
        return np.sqrt(((pos[ids][:, None, :] - roots[None]) ** 2).sum(-1)).min(1)

    i2s_h, comp_h = -np.ones(n_ep, np.int32), -np.ones(n_ep, np.int32)
    off_h, rows_h, seg_h = walk_chains(pairs, n_ep, i2s_h, comp_h, end_distance_np)
    pt = torch.as_tensor(pairs, device="cuda")
    pos_t, roots_t = torch.as_tensor(pos, device="cuda"), torch.as_tensor(roots, device="cuda")
    end_distance_t = lambda ids: torch.cdist(pos_t[ids], roots_t).min(dim=1).values
    for form in (walk_chains_device, walk_chains_torch):
        off, rows, seg, i2s, comp = form(pt, n_ep, end_distance_t)
        np.testing.assert_array_equal(off.cpu().numpy(), off_h, err_msg=form.__name__)
        np.testing.assert_array_equal(rows.cpu().numpy(), rows_h, err_msg=form.__name__)
        np.testing.assert_array_equal(seg.cpu().numpy(), seg_h, err_msg=form.__name__)
        np.testing.assert_array_equal(i2s.cpu().numpy(), i2s_h, err_msg=form.__name__)
        np.testing.assert_array_equal(comp.cpu().numpy(), comp_h, err_msg=form.__name__)
    assert off_h.shape[0] - 1 == n_chain and int(off_h[-1]) == int(lens[:n_chain].sum())
    bad = np.concatenate([pairs, np.array([[pairs[0, 0], n_ep - 1], [pairs[0, 0], n_ep - 2]], np.int64)])
    assert walk_chains_device(torch.as_tensor(bad, device="cuda"), n_ep, end_distance_t) is None
    assert walk_chains_device(torch.zeros((0, 2), dtype=torch.int64, device="cuda"), n_ep, end_distance_t)[0].tolist() == [0]


@pytest.mark.parametrize("kind", ["cloud", "strands"])
def test_lazy_records_in_the_fused_iterations(kind):
    """hgs_set_lazy_records in the single-pass 7-channel iteration (capacity mode, the one-launch parameter + preprocess kernels, the
    row sums by row_reduce_kernel for the cloud): records built by the blend kernels against records packed by the sort kernel --
    loss terms, image planes, radii and every parameter gradient bit for bit over six iterations."""
    import hgs_runtime as rt
    from arguments import OptimizationParams
    from diff_gaussian_rasterization import _C as raster
    from hgs_runtime.strand_step import FusedCloudStep, FusedStrandStep, ViewTable
    from synthetic import attach_targets, build_workload, cameras_extent, make_cameras, make_cloud_model
    if kind == "cloud":
        cams = make_cameras(4, 200, 120, device="cuda")
        model = make_cloud_model(3000, device="cuda", spatial_lr_scale=cameras_extent(cams))
        with torch.no_grad():
            model._scaling.add_(1.3)
        attach_targets(cams, model)
    else:
        model, cams, _ = build_workload("tiny", device="cuda", with_targets=True)
    opt = OptimizationParams()
    model.training_setup(opt)
    params = ([model._xyz, model._scaling, model._rotation, model._opacity, model._mask, model._features_dc] if kind == "cloud"
              else [model._endpoints, model._width, model._opacity, model._mask, model._features_dc])
    out = {}
    try:
        raster._state["cap"] = 0
        raster.set_async(True, slack=2.0)
        for lazy in (0, 1):
            rt.lib().hgs_set_lazy_records(lazy)
            views = ViewTable(cams)
            step = (FusedCloudStep if kind == "cloud" else FusedStrandStep)(model, views, opt, torch.zeros(3, device="cuda"))
            seen = []
            for v in (0, 2, 1, 3, 0, 2):
                for p in params:
                    p.grad = None
                views.prologue(v, ride=True)
                loss, terms = step.loss()
                step.backward(loss)
                raster.check_async()
                seen.append([loss.detach().clone(), terms[:14].clone(), step.last["planes"].clone(), step.last["radii"].clone()]
                            + [p.grad.clone() for p in params])
            out[lazy] = seen
    finally:
        raster.set_async(False)
        rt.lib().hgs_set_lazy_records(-1)
    for a, b in zip(out[0], out[1]):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
        assert bool(a[4].abs().sum() > 0)
