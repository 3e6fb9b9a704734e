"""gaussian_renderer.frames.FrameRenderer (forward-only renders as a captured graph, the loop of reference render.py:57-62)
against render(): the same image and radii, bit for bit, whatever happens to the model in between."""
import pytest

pytestmark = pytest.mark.gpu


def _setup(kind):
    import torch
    from arguments import OptimizationParams
    from synthetic import build_workload, cameras_extent, make_cameras, make_cloud_model
    if kind == "hair":
        model, cams, _ = build_workload("tiny", device="cuda", with_targets=False)
    else:
        cams = make_cameras(5, 200, 120, device="cuda")
        model = make_cloud_model(3000, device="cuda", spatial_lr_scale=cameras_extent(cams))
    model.training_setup(OptimizationParams())
    return model, cams, torch.tensor([0.1, 0.2, 0.3], device="cuda")


def _same(fr, cams, model, bg, views=None):
    import torch
    from gaussian_renderer import render
    for i in (range(len(cams)) if views is None else views):
        got = fr.render(i)
        with torch.no_grad():
            ref = render(cams[i], model, bg)
        assert torch.equal(got["render"], ref["render"]) and torch.equal(got["radii"], ref["radii"]), i
        assert bool(ref["render"].abs().sum() > 0)


@pytest.mark.parametrize("kind", ["hair", "cloud"])
@pytest.mark.parametrize("use_graph", [True, False])
def test_frame_renderer_equals_render(kind, use_graph):
    import torch
    from gaussian_renderer.frames import FrameRenderer
    model, cams, bg = _setup(kind)
    fr = FrameRenderer(model, cams, bg, use_graph=use_graph)      # cameras without targets: matrices only
    _same(fr, cams, model, bg)
    _same(fr, cams, model, bg, views=[3, 3, 0])                    # replays, repeated views
    assert fr.captures == (1 if use_graph else 0)
    # parameter VALUES change in place (a training step, also through .data): the next frame shows the new state
    with torch.no_grad():
        (model._endpoints if kind == "hair" else model._xyz).data.add_(0.002)
        model._features_dc.mul_(0.9)
    _same(fr, cams, model, bg, views=[1, 2])
    assert fr.captures == (1 if use_graph else 0)
    # the SH degree goes up: a by-value argument of the captured launches -> captured again
    model.max_sh_degree = max(model.max_sh_degree, 1)
    if model._features_rest.shape[1] >= 3:
        model.oneupSHdegree()
        _same(fr, cams, model, bg, views=[0, 4 % len(cams)])
        assert fr.captures == (2 if use_graph else 0)


def test_frame_renderer_follows_topology_changes():
    import torch
    from gaussian_renderer.frames import FrameRenderer
    model, cams, bg = _setup("cloud")
    fr = FrameRenderer(model, cams, bg)
    _same(fr, cams, model, bg, views=[0])
    keep = torch.ones(model.get_xyz.shape[0], dtype=torch.bool, device="cuda")
    keep[::3] = False
    model.prune_points(~keep)                                     # tensors re-created with another size
    _same(fr, cams, model, bg)
    assert fr.captures == 2


def test_frame_renderer_unchecked_frames_and_capacity():
    """check=False only enqueues; validate() reports the frames of a graph whose capacity a view exceeded, and the renderer
    captures again with a larger one."""
    import torch
    from diff_gaussian_rasterization import _C as raster
    from gaussian_renderer import render
    from gaussian_renderer.frames import FrameRenderer
    model, cams, bg = _setup("hair")
    mode_before = raster._state["async"]
    fr = FrameRenderer(model, cams, bg, slack=1.0)
    imgs = [fr.render(i, check=False)["render"].clone() for i in range(len(cams))]
    assert fr.validate() == []
    with torch.no_grad():
        for i, c in enumerate(cams):
            assert torch.equal(imgs[i], render(c, model, bg)["render"])
    # make the captured capacity too small for what the model becomes: every strand ten times as wide
    cap_before = fr._cap
    with torch.no_grad():
        model._width.add_(4.0)
    fr.render(0, check=False)
    bad = fr.validate()
    if bad:                                                        # (the wider strands needed more instances than the capacity)
        assert bad == [0] and fr._graph is None
        _same(fr, cams, model, bg, views=[0, 1])
        assert fr._cap > cap_before and fr.captures == 2
    else:
        _same(fr, cams, model, bg, views=[0, 1])
    out = fr.render(2)                                             # a checked render repairs an overflow by itself
    with torch.no_grad():
        assert torch.equal(out["render"], render(cams[2], model, bg)["render"])
    assert raster._state["async"] is mode_before                  # the module's mode is as it was


def test_view_table_without_targets_is_refused_by_the_training_step():
    import torch
    from arguments import OptimizationParams
    import hgs_runtime as rt
    from hgs_runtime.strand_step import FusedStrandStep, ViewTable
    model, cams, bg = _setup("hair")
    with pytest.raises(rt.HgsError):
        FusedStrandStep(model, ViewTable(cams, targets=False), OptimizationParams(), torch.zeros(3, device="cuda"))


def test_frame_renderer_batches():
    """frames_per_launch = K: K frames by one graph launch, each equal to render() of its view."""
    import torch
    from gaussian_renderer import render
    from gaussian_renderer.frames import FrameRenderer
    model, cams, bg = _setup("hair")
    fr = FrameRenderer(model, cams, bg, frames_per_launch=3)
    for views in ([0, 1, 2], [3, 3, 1], [2, 0]):                  # (a short batch falls back to single frames)
        outs = fr.render_batch(views)
        assert len(outs) == len(views)
        with torch.no_grad():
            for v, o in zip(views, outs):
                ref = render(cams[v], model, bg)
                assert torch.equal(o["render"], ref["render"]) and torch.equal(o["radii"], ref["radii"])
    _same(fr, cams, model, bg, views=[1])                         # single frames still work next to the batch graph
    assert fr.captures == 1


def test_kernel_sigmoid_has_torch_sigmoids_bits():
    """FrameRenderer takes the strands' opacity from hgs_hair_params_forward (1 / (1 + expf(-x))) where render() calls
    torch.sigmoid: the two agree bit for bit (10^6 values over the range of opacity logits, and the special cases)."""
    import ctypes as C
    import numpy as np
    import torch
    import hgs_runtime as rt
    P = 1_000_000
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(P, 1, device="cuda", generator=g) * 6.0
    x[:8, 0] = torch.tensor([0.0, -0.0, 30.0, -30.0, 88.0, -88.0, 1e-8, -104.0], device="cuda")
    ep = torch.rand(P + 1, 3, device="cuda")
    pairs = torch.stack([torch.arange(P, device="cuda"), torch.arange(1, P + 1, device="cuda")], 1).contiguous()
    w = torch.zeros(P, 1, device="cuda")
    f32 = dict(dtype=torch.float32, device="cuda")
    xyz, scale, quat = torch.empty((P, 3), **f32), torch.empty((P, 3), **f32), torch.empty((P, 4), **f32)
    o, e4 = torch.empty((P, 1), **f32), torch.empty((P, 4), **f32)
    rt.check(rt.lib().hgs_hair_params_forward(rt.current_stream(), P, rt.ptr(ep), rt.ptr(pairs), rt.ptr(w), 1.0, rt.ptr(x), rt.ptr(x),
                                              rt.ptr(xyz), rt.ptr(scale), rt.ptr(quat), None, rt.ptr(o), rt.ptr(e4), None))
    ref = torch.sigmoid(x)
    assert torch.equal(o.view(torch.int32), ref.view(torch.int32))
    assert torch.equal(e4[:, 0].contiguous().view(torch.int32), ref[:, 0].contiguous().view(torch.int32))


def test_frame_renderer_on_an_empty_model():
    """No Gaussians: the frame is the background (eager path; nothing to capture), as render() gives it."""
    import torch
    from gaussian_renderer import render
    from gaussian_renderer.frames import FrameRenderer
    model, cams, bg = _setup("cloud")
    model.prune_points(torch.ones(model.get_xyz.shape[0], dtype=torch.bool, device="cuda"))
    assert model.get_xyz.shape[0] == 0
    fr = FrameRenderer(model, cams, bg)
    out = fr.render(1)
    with torch.no_grad():
        ref = render(cams[1], model, bg)["render"]
    assert torch.equal(out["render"], ref) and fr.captures == 0
    assert torch.equal(out["render"], bg.view(3, 1, 1).expand_as(ref))
