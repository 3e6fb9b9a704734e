"""Model containers on CPU tensors (the rasterizer itself needs the GPU; see tests/test_gpu_train.py)."""
import math

import numpy as np
import pytest
import torch

from arguments import OptimizationParams


def _cloud(n=50, seed=0):
    from scene.gaussian_model import GaussianModel
    from torch import nn
    g = torch.Generator().manual_seed(seed)
    m = GaussianModel(sh_degree=1, spatial_lr_scale=2.0, device="cpu")
    m._xyz = nn.Parameter(torch.randn(n, 3, generator=g))
    m._features_dc = nn.Parameter(torch.randn(n, 1, 3, generator=g))
    m._features_rest = nn.Parameter(torch.zeros(n, 3, 3))
    m._scaling = nn.Parameter(torch.log(torch.rand(n, 3, generator=g) * 0.1 + 0.01))
    m._rotation = nn.Parameter(torch.randn(n, 4, generator=g))
    m._opacity = nn.Parameter(torch.randn(n, 1, generator=g))
    m._mask = nn.Parameter(torch.randn(n, 1, generator=g))
    m.max_radii2D = torch.zeros(n)
    return m


def test_gaussian_model_getters_and_covariance():
    m = _cloud()
    assert torch.allclose(m.get_scaling, torch.exp(m._scaling))
    assert torch.allclose(m.get_rotation.norm(dim=1), torch.ones(50), atol=1e-6)
    assert m.get_features.shape == (50, 4, 3)
    cov = m.get_covariance(1.0)
    from utils.transform import build_rotation
    R = build_rotation(m._rotation)
    S = torch.diag_embed(m.get_scaling)
    full = R @ S @ S @ R.transpose(1, 2)
    ref = torch.stack([full[:, 0, 0], full[:, 0, 1], full[:, 0, 2], full[:, 1, 1], full[:, 1, 2], full[:, 2, 2]], 1)
    assert torch.allclose(cov, ref, atol=1e-6)
    ori = m.get_orientation
    idx = m.get_scaling.argmax(1)
    assert torch.allclose(ori, R[torch.arange(50), :, idx], atol=1e-6)


def test_training_setup_groups_and_lr_schedule():
    m = _cloud()
    opt = OptimizationParams()
    m.training_setup(opt)
    names = [g["name"] for g in m.optimizer.param_groups]
    assert names == ["xyz", "f_dc", "f_rest", "opacity", "scaling", "mask", "rotation"]
    lrs = {g["name"]: g["lr"] for g in m.optimizer.param_groups}
    assert lrs["xyz"] == pytest.approx(0.00016 * 2.0) and lrs["f_rest"] == pytest.approx(0.025 / 20)
    assert m.optimizer.defaults["eps"] == 1e-15
    assert m.update_learning_rate(0) == pytest.approx(0.00016 * 2.0)
    assert m.update_learning_rate(30000) == pytest.approx(0.0000016 * 2.0)
    mid = m.update_learning_rate(15000)
    assert mid == pytest.approx(math.sqrt(0.00016 * 0.0000016) * 2.0, rel=1e-6)
    assert float(m.dist_to_scale_factor) == pytest.approx(0.5102133812190369, rel=1e-6)


def test_densify_prune_keeps_optimizer_state_consistent():
    torch.manual_seed(0)
    m = _cloud(n=40)
    opt = OptimizationParams()
    m.training_setup(opt)
    loss = (m._xyz ** 2).sum() + (m._opacity ** 2).sum() + (m._scaling ** 2).sum() + (m._rotation ** 2).sum() + \
        (m._features_dc ** 2).sum() + (m._features_rest ** 2).sum() + (m._mask ** 2).sum()
    loss.backward()
    m.optimizer.step()
    m.xyz_gradient_accum[:] = 1.0
    m.denom[:] = 1.0
    m.densification(extent=1.0, max_screen_size=None)
    n = m.get_xyz.shape[0]
    for g in m.optimizer.param_groups:
        p = g["params"][0]
        assert p.shape[0] == n and p.requires_grad
        st = m.optimizer.state[p]
        assert st["exp_avg"].shape == p.shape and st["exp_avg_sq"].shape == p.shape
    assert m.xyz_gradient_accum.shape == (n, 1) and m.max_radii2D.shape == (n,)
    m.reset_opacity()
    assert float(m.get_opacity.max()) <= 0.01 + 1e-6
    st = m.optimizer.state[m._opacity]
    assert float(st["exp_avg"].abs().max()) == 0.0


def test_sort_spatially_permutes_parameters_moments_and_statistics_alike():
    """GaussianModel.sort_spatially (Morton order of the centres; what a GPU cloud does in training_setup and after every
    densification): one permutation for all parameter groups, their Adam moments and the statistics; the cloud as a set is
    unchanged, neighbours on the curve are neighbours in space, a second sort is the identity."""
    torch.manual_seed(0)
    m = _cloud(n=500)
    opt = OptimizationParams()
    m.training_setup(opt)                      # (CPU model: no automatic sort)
    (m._xyz ** 2).sum().backward()
    ((m._opacity ** 2).sum() + (m._features_dc ** 3).sum() + (m._scaling ** 2).sum() + (m._rotation ** 2).sum()
     + (m._features_rest ** 2).sum() + (m._mask ** 2).sum()).backward()
    m.optimizer.step()
    m.xyz_gradient_accum[:] = torch.arange(500.0)[:, None]
    m.max_radii2D[:] = torch.arange(500.0)
    before = {g["name"]: (g["params"][0].detach().clone(), m.optimizer.state[g["params"][0]]["exp_avg"].clone(),
                          m.optimizer.state[g["params"][0]]["exp_avg_sq"].clone()) for g in m.optimizer.param_groups}
    perm = m.sort_spatially()
    assert sorted(perm.tolist()) == list(range(500))
    for g in m.optimizer.param_groups:
        p = g["params"][0]
        old_p, old_m, old_v = before[g["name"]]
        assert p.requires_grad and torch.equal(p.detach(), old_p[perm])
        st = m.optimizer.state[p]
        assert torch.equal(st["exp_avg"], old_m[perm]) and torch.equal(st["exp_avg_sq"], old_v[perm])
    assert torch.equal(m.xyz_gradient_accum[:, 0], perm.to(torch.float32)) and torch.equal(m.max_radii2D, perm.to(torch.float32))
    assert m._xyz is m.optimizer.param_groups[0]["params"][0]
    # locality: the mean distance between consecutive centres drops well below that of the unordered cloud
    step_sorted = (m._xyz[1:] - m._xyz[:-1]).norm(dim=1).mean()
    step_before = (before["xyz"][0][1:] - before["xyz"][0][:-1]).norm(dim=1).mean()
    assert float(step_sorted) < 0.5 * float(step_before)
    assert m.sort_spatially().tolist() == list(range(500))


def test_capture_restore_round_trip_with_spatial_sort(monkeypatch):
    """restore() after a capture() whose cloud was re-ordered since: Adam moments, statistics and parameters must end up
    attached to the same Gaussians (training_setup sorts a GPU cloud; here the sort is forced on the CPU model, and the
    centres move between capture and restore so that the permutation at restore is not the identity)."""
    from scene.gaussian_model import GaussianModel
    monkeypatch.setattr(GaussianModel, "_maybe_sort_spatially", lambda self: self.sort_spatially())
    torch.manual_seed(1)
    m = _cloud(n=300)
    opt = OptimizationParams()
    m.training_setup(opt)
    for _ in range(3):      # three Adam steps: moments differ per Gaussian, centres move
        m.update_learning_rate(1)
        ((m._xyz ** 2).sum() * 50 + (m._opacity ** 2).sum() + (m._features_dc ** 3).sum() + (m._scaling ** 2).sum()
         + (m._rotation ** 2).sum() + (m._features_rest ** 2).sum() + (m._mask ** 2).sum()).backward()
        m.optimizer.step()
        m.optimizer.zero_grad(set_to_none=True)
    with torch.no_grad():
        m._xyz.add_(torch.randn(300, 3) * 0.5)      # (the curve order of the moved cloud differs)
    m.xyz_gradient_accum[:] = torch.arange(300.0)[:, None]
    m.denom[:] = 2 * torch.arange(300.0)[:, None]
    m.max_radii2D = 3 * torch.arange(300.0)
    tag = m._opacity.detach().clone().reshape(-1)    # identifies a Gaussian whatever its position
    want = {float(tag[i]): (m._xyz[i].detach().clone(), m.optimizer.state[m._xyz]["exp_avg"][i].clone(),
                            m.optimizer.state[m._scaling]["exp_avg_sq"][i].clone(), float(i)) for i in range(300)}
    ckpt = m.capture()
    m2 = GaussianModel(sh_degree=1, spatial_lr_scale=2.0, device="cpu")
    m2.restore(ckpt, opt)
    order = [want[float(t)][3] for t in m2._opacity.detach().reshape(-1)]
    assert order != sorted(order)                    # restore() did re-order the cloud
    for i in range(300):
        xyz, m1, v2, k = want[float(m2._opacity[i])]
        assert torch.equal(m2._xyz[i].detach(), xyz)
        assert torch.equal(m2.optimizer.state[m2._xyz]["exp_avg"][i], m1)
        assert torch.equal(m2.optimizer.state[m2._scaling]["exp_avg_sq"][i], v2)
        assert float(m2.xyz_gradient_accum[i]) == k and float(m2.denom[i]) == 2 * k and float(m2.max_radii2D[i]) == 3 * k


def _strands(S=6, V=9, seed=0):
    from synthetic import strand_polylines
    from scene.hair_gaussian_model import HairGaussianModel
    pts = strand_polylines(S, V - 1, seed=seed)
    return HairGaussianModel.from_strands(pts, device="cpu", sh_degree=0), pts


def test_hair_model_closed_forms():
    m, pts = _strands()
    m.set_pval(0.05)
    e0 = torch.from_numpy(pts[:, :-1].reshape(-1, 3))
    e1 = torch.from_numpy(pts[:, 1:].reshape(-1, 3))
    assert torch.allclose(m.get_xyz, 0.5 * (e0 + e1), atol=1e-7)
    d = e1 - e0
    ln = d.norm(dim=1)
    sc = m.get_scaling
    assert torch.allclose(sc[:, 0], ln / 2 * float(m.dist_to_scale_factor), rtol=1e-6)
    assert torch.allclose(sc[:, 1:], torch.full_like(sc[:, 1:], 1e-4), rtol=1e-5)
    q = m.get_rotation
    from utils.transform import build_rotation, xaxis_to_direction_quaternion
    assert torch.allclose(q, xaxis_to_direction_quaternion(d / ln[:, None]), atol=1e-5)
    assert torch.allclose(build_rotation(q)[:, :, 0], d / ln[:, None], atol=1e-5)
    assert torch.allclose(m.get_orientation, d / ln[:, None], atol=1e-6)
    # collapsed segment -> identity rotation, x_hat orientation, clamped scale
    with torch.no_grad():
        m._endpoints[1] = m._endpoints[0]
    assert torch.equal(m.get_rotation[0], torch.tensor([1.0, 0, 0, 0]))
    assert torch.equal(m.get_orientation[0], torch.tensor([1.0, 0, 0]))
    assert float(m.get_scaling[0, 0]) == pytest.approx(1e-7)


def test_hair_gradients_reach_shared_endpoints():
    m, _ = _strands()
    (m.get_xyz.sum() + m.get_scaling.sum() + (m.get_rotation ** 2).sum()).backward()
    g = m._endpoints.grad
    assert g is not None and torch.isfinite(g).all() and (g.abs().sum(dim=1) > 0).all()


def test_strands_info_and_smoothness_pairs():
    m, pts = _strands(S=5, V=7)
    m.compute_strands_info(only_foreground=True)
    info = m.strands_info
    assert len(info.list_strands) == 5
    for s, strand in enumerate(info.list_strands):
        assert strand.shape == (6, 2)
        assert (strand[1:, 0] == strand[:-1, 1]).all()          # consecutive segments share an endpoint
        assert strand[0, 0] == s * 7                             # oriented root -> tip
        assert info.strand_endpoint_id_to_complementary[s * 7] == s * 7 + 6
    pairs = m.smoothness_index_pairs()
    assert pairs.shape == (5 * 5, 2, 2)
    from loss.losses import angle_smoothness_loss
    v = angle_smoothness_loss(m, threshold=0.0)
    assert torch.is_tensor(v) and float(v) > 0
    # reversed storage order of one strand must still come out root -> tip
    m.endpoint_pairs = m.endpoint_pairs.flip(0)
    m.compute_strands_info(only_foreground=True)
    assert sorted(int(s[0, 0]) for s in m.strands_info.list_strands) == [0, 7, 14, 21, 28]


def test_hair_prune_and_cat_segments():
    m, _ = _strands(S=4, V=6)
    m.training_setup(OptimizationParams())
    P0, E0 = m.endpoint_pairs.shape[0], m._endpoints.shape[0]
    prune = torch.zeros(P0, dtype=torch.bool)
    prune[:5] = True  # the whole first strand
    m.prune_segments(prune)
    assert m.endpoint_pairs.shape[0] == P0 - 5 and m._endpoints.shape[0] == E0 - 6
    assert int(m.endpoint_pairs.max()) == m._endpoints.shape[0] - 1
    n_new = 3
    m.cat_segments(torch.tensor([[0, 1]] * n_new), torch.zeros(0, 3), torch.zeros(n_new, 1, 3), torch.zeros(n_new, 0, 3),
                   torch.zeros(n_new, 1), torch.zeros(n_new, 1), torch.zeros(n_new, 1))
    assert m.endpoint_pairs.shape[0] == P0 - 5 + n_new == m._opacity.shape[0] == m.denom.shape[0]


def test_synthetic_rig_sees_the_anchor():
    from synthetic import make_cameras
    cams = make_cameras(5, 64, 48, device="cpu", dist=0.5)
    for cam in cams:
        c = cam.camera_center
        assert abs(float(c.norm()) - 0.5) < 1e-5
        fwd = cam.world_view_transform[:3, 2]          # world-space viewing direction (third column of W2C^T)
        assert torch.allclose(fwd, -c / c.norm(), atol=1e-5)
    assert float(cams[-1].camera_center[1]) > 0.49      # the extra camera looks down from above (y-up world)


def test_generate_cameras_rig():
    from utils.camera import generate_cameras
    pose = np.eye(4)
    pose[:3, 3] = [0, 0, -0.5]
    cams, Es = generate_cameras(8, 100, 200, cam_pose=pose, offset=0.5, focal_length_px=100)
    assert len(cams) == len(Es) == 8 and cams[1].params == [100, 100.0, 50.0]
    for i in range(1, 8):
        c2w = np.linalg.inv(Es[i])
        assert np.linalg.norm(c2w[:3, 3]) == pytest.approx(0.5)
        fwd = c2w[:3, 2]
        assert np.allclose(fwd, -c2w[:3, 3] / 0.5, atol=1e-6)   # every ring camera looks at the anchor
    top = np.linalg.inv(Es[8])
    assert np.allclose(top[:3, 3], [0, 0.5, 0])


def _chain_invariants(m):
    pairs = m.endpoint_pairs
    assert int(pairs.max()) == m._endpoints.shape[0] - 1 and int(pairs.min()) == 0
    counts = torch.bincount(pairs.flatten(), minlength=m._endpoints.shape[0])
    assert int(counts.min()) >= 1 and int(counts.max()) <= 2          # open polylines only, every endpoint referenced
    P = pairs.shape[0]
    for t in (m._features_dc, m._features_rest, m._opacity, m._mask, m._width, m.denom, m.xyz_gradient_accum):
        assert t.shape[0] == P
    for g in m.optimizer.param_groups:
        p = g["params"][0]
        assert p.shape[0] == (m._endpoints.shape[0] if g["name"] == "endpoints" else P)


def test_hair_merging_joins_facing_strand_ends():
    """Two collinear strands whose tip/root are 1 mm apart get merged into one; a far-away strand is untouched."""
    from scene.hair_gaussian_model import HairGaussianModel
    x = np.linspace(0, 0.02, 6)
    s0 = np.stack([x, np.zeros(6), np.zeros(6)], 1)
    s1 = np.stack([x + 0.021, np.zeros(6), np.zeros(6)], 1)          # starts 1 mm after s0 ends, same direction
    s2 = np.stack([x, np.full(6, 0.05), np.zeros(6)], 1)             # parallel, 5 cm away
    pts = np.stack([s0, s1, s2]).astype(np.float32)
    roots = np.array([[0, 0, 0], [0, 0.05, 0]], np.float32)
    m = HairGaussianModel.from_strands(pts, device="cpu", ref_strand_root=roots)
    m.training_setup(OptimizationParams())
    m.compute_strands_info()
    assert len(m.strands_info.list_strands) == 3
    pairs = m.compute_endpoint_pair_to_merge()
    assert pairs.shape == (1, 2) and sorted(pairs[0].tolist()) == [5, 6]      # tip of s0 with root of s1
    P0 = m.endpoint_pairs.shape[0]
    m.merging()
    assert m.endpoint_pairs.shape[0] == P0                 # two end segments re-created, none lost
    assert m._endpoints.shape[0] == 18 - 1                 # two endpoints fused into one
    assert len(m.strands_info.list_strands) == 2
    lens = sorted(s.shape[0] for s in m.strands_info.list_strands)
    assert lens == [5, 10]
    _chain_invariants(m)
    long = max(m.strands_info.list_strands, key=len)
    assert np.allclose(m._endpoints.detach().numpy()[long[0, 0]], [0, 0, 0], atol=1e-6)   # oriented from the root


def test_hair_densification_split_clone_prune():
    from scene.hair_gaussian_model import HairGaussianModel
    from synthetic import strand_polylines
    torch.manual_seed(0)
    m = HairGaussianModel.from_strands(strand_polylines(6, 8, seed=1), device="cpu")
    opt = OptimizationParams()
    m.training_setup(opt)
    m.compute_strands_info()
    P0, E0 = m.endpoint_pairs.shape[0], m._endpoints.shape[0]
    # split: give the first strand a large gradient + make extent tiny so the "large" branch triggers
    m.denom[:] = 1.0
    m.xyz_gradient_accum[:8] = 1.0
    with torch.no_grad():
        m._opacity[40:44] = -10.0            # transparent tail segments of the last strand -> pruned (strand ends only)
    m.densification(extent=1e-3, max_screen_size=None)
    _chain_invariants(m)
    info_strands = m.strands_info.list_strands
    assert m.endpoint_pairs.shape[0] > P0 - 4            # 8 segments were split in two, at most 4 pruned
    assert sum(s.shape[0] for s in info_strands) <= m.endpoint_pairs.shape[0]
    # clone branch: large gradient, extent huge -> small relative size
    m.denom[:] = 1.0
    m.xyz_gradient_accum[:] = 0.0
    m.xyz_gradient_accum[:3] = 1.0
    n1 = m.endpoint_pairs.shape[0]
    m.max_segment_length = torch.tensor(1e9)
    m.densification(extent=1e3, max_screen_size=None)
    assert m.endpoint_pairs.shape[0] == n1 + 3
    _chain_invariants(m)


def test_hair_sort_spatially_restores_strand_order_without_changing_the_model():
    """HairGaussianModel.sort_spatially: after clones / splits / merges appended their results at the end of the arrays,
    the storage goes back to strand by strand, root -> tip -- parameters, Adam moments, statistics and id tables by ONE
    pair of permutations; the model (segments with their attributes, strands as vertex sequences) is unchanged as a set,
    and a second call finds nothing to move."""
    from scene.hair_gaussian_model import HairGaussianModel
    from synthetic import strand_polylines
    torch.manual_seed(0)
    m = HairGaussianModel.from_strands(strand_polylines(12, 10, seed=5), device="cpu")
    opt = OptimizationParams()
    opt.spatial_sort = False                      # (the operators below must leave their mess behind)
    m.training_setup(opt)
    m.compute_strands_info()
    # a few Adam steps so that moments differ per element, then operators that append / prune
    for _ in range(2):
        ((m._endpoints ** 2).sum() * 30 + (m._opacity ** 2).sum() + (m._features_dc ** 3).sum() + (m._width ** 2).sum()
         + (m._mask ** 2).sum()).backward()
        m.optimizer.step()
        m.optimizer.zero_grad(set_to_none=True)
    m.denom[:] = 1.0
    m.xyz_gradient_accum[::3] = 1.0               # every third segment is split in two
    m.densification(extent=1e-3, max_screen_size=None)
    m.xyz_gradient_accum[:, 0] = torch.arange(m.endpoint_pairs.shape[0], dtype=torch.float32)
    m.max_radii2D = 2 * torch.arange(m.endpoint_pairs.shape[0], dtype=torch.float32)

    def segment_records(model):
        ep = model._endpoints.detach()[model.endpoint_pairs]                 # [P, 2, 3]
        st_w = model.optimizer.state[model._width]
        st_e = model.optimizer.state[model._endpoints]
        mom_e = st_e["exp_avg"][model.endpoint_pairs]                        # endpoint moments seen from the segments
        rec = torch.cat([ep.reshape(-1, 6), model._opacity.detach(), model._width.detach(), model._features_dc.detach().reshape(-1, 3),
                         st_w["exp_avg"], st_w["exp_avg_sq"], mom_e.reshape(-1, 6), model.xyz_gradient_accum,
                         model.max_radii2D[:, None]], dim=1).numpy()
        return rec[np.lexsort(rec.T[::-1])]

    def strands_as_points(model):
        ep = model._endpoints.detach().numpy()
        return sorted(tuple(map(tuple, ep[np.r_[s[:, 0], s[-1, 1]]].tolist())) for s in model.strands_info.list_strands)
    before, strands_before = segment_records(m), strands_as_points(m)
    rows0 = [np.asarray(r) for r in m.strands_info.list_strands_segments_id]
    assert any(np.any(np.diff(r) != 1) for r in rows0)                        # storage IS scrambled
    perms = m.sort_spatially()
    assert perms is not None and sorted(perms[0].tolist()) == list(range(m.endpoint_pairs.shape[0]))
    _chain_invariants(m)
    assert np.array_equal(segment_records(m), before) and strands_as_points(m) == strands_before
    # strand by strand: the segments of a strand are consecutive rows, its vertices consecutive ids, root first
    for rows, seq in zip(m.strands_info.list_strands_segments_id, m.strands_info.list_strands):
        assert np.all(np.diff(rows) == 1) and np.all(np.diff(seq[:, 0]) == 1) and np.all(seq[:, 1] == seq[:, 0] + 1)
    for g in m.optimizer.param_groups:
        assert g["params"][0] is getattr(m, dict(m._PARAM_ATTRS)[g["name"]]) and g["params"][0].requires_grad
    # the sort renumbers the strand bookkeeping instead of walking the chains again: the result IS what a fresh walk gives
    remapped = m.strands_info
    m.compute_strands_info()
    fresh = m.strands_info
    for a in ("offsets", "rows", "segment_rows", "id_to_strand_id", "strand_endpoint_id_to_complementary"):
        assert np.array_equal(np.asarray(getattr(remapped, a)), np.asarray(getattr(fresh, a))), a
    assert m.sort_spatially() is None


def test_merge_collapsed_segment_fuses_its_endpoints():
    from scene.hair_gaussian_model import HairGaussianModel
    from synthetic import strand_polylines
    m = HairGaussianModel.from_strands(strand_polylines(2, 6, seed=2), device="cpu")
    m.training_setup(OptimizationParams())
    with torch.no_grad():
        m._endpoints[3] = m._endpoints[2]     # interior segment (2,3) of strand 0 collapses
    P0, E0 = m.endpoint_pairs.shape[0], m._endpoints.shape[0]
    info = {}
    m.merge_collapsed_segments(info)
    assert info["merge_collapsed"] == 1
    assert m.endpoint_pairs.shape[0] == P0 - 1 and m._endpoints.shape[0] == E0 - 1
    _chain_invariants(m)
    m.compute_strands_info()
    assert sorted(s.shape[0] for s in m.strands_info.list_strands) == [5, 6]


def test_merge_driver_rounds_until_nothing_left():
    """merge.py's loop (reference merge.py:123-190): three collinear strands 1 mm apart end up as one after two rounds."""
    import merge as merge_cli
    from scene.hair_gaussian_model import HairGaussianModel
    x = np.linspace(0, 0.02, 6)
    strands = [np.stack([x + k * 0.021, np.zeros(6), np.zeros(6)], 1) for k in range(3)]
    m = HairGaussianModel.from_strands(np.stack(strands).astype(np.float32), device="cpu",
                                       ref_strand_root=np.array([[0, 0, 0]], np.float32))
    m.training_setup(OptimizationParams())
    m.compute_strands_info()
    log = []
    rounds = merge_cli.merge_rounds(m, 10, log=log.append)
    assert 1 <= rounds <= 2 and len(log) == rounds
    assert len(m.strands_info.list_strands) == 1 and m.strands_info.list_strands[0].shape[0] == 15
    assert merge_cli.merge_rounds(m, 10, log=log.append) == 0      # nothing left to merge
    _chain_invariants(m)


def _c1_merge(device):
    """BASELINE.json C1 at its stated size: 1k-Gaussian cloud -> to_hair_gaussian_model -> merge_rounds (reference
    merge.py:114-190).  Returns (model, rounds, strands as sorted tuples of rounded vertex coordinates)."""
    import merge as merge_cli
    from tests.scenes import c1_cloud
    cloud, pts = c1_cloud(device=device)
    opt = OptimizationParams()
    cloud.training_setup(opt)
    with torch.no_grad():
        hair = cloud.to_hair_gaussian_model()
        assert hair.get_xyz.shape[0] == 1000 and hair._endpoints.shape[0] == 2000
        assert hair.strands_info.n_strands == 1000            # every Gaussian one disconnected segment
        rounds = merge_cli.merge_rounds(hair, opt.iterations, log=lambda *_: None)
    ep = hair._endpoints.detach().cpu().numpy()
    strands = sorted(tuple(map(tuple, np.round(ep[np.r_[s[:, 0], s[-1, 1]]], 5).tolist())) for s in hair.strands_info.list_strands)
    return hair, rounds, strands, pts


def test_c1_merge_at_size_reassembles_the_strands():
    """C1 (merge.py on 1k synthetic Gaussians, CPU model): the 1000 shuffled line-like Gaussians of 50 polylines x 20
    segments are merged back into exactly those 50 strands of 20 segments, vertex for vertex, in ~log2(20) rounds; a
    second call finds nothing left to merge."""
    import merge as merge_cli
    hair, rounds, strands, pts = _c1_merge("cpu")
    _chain_invariants(hair)
    assert 4 <= rounds <= 8
    assert hair.strands_info.n_strands == 50 and hair.get_xyz.shape[0] == 1000 and hair._endpoints.shape[0] == 50 * 21
    assert all(s.shape[0] == 20 for s in hair.strands_info.list_strands)
    assert merge_cli.merge_rounds(hair, 10, log=lambda *_: None) == 0
    # the merged strands ARE the generating polylines (root -> tip: the roots are the reference roots)
    want = sorted(tuple(map(tuple, np.round(p, 5).tolist())) for p in pts)
    got = np.asarray(strands)
    assert np.abs(got - np.asarray(want)).max() <= 2e-5
    # appearance travelled with the segments: the multiset of colours is the cloud's
    from tests.scenes import c1_cloud
    cloud, _ = c1_cloud(device="cpu")
    a = np.sort(hair._features_dc.detach().numpy().reshape(-1, 3), axis=0)
    b = np.sort(cloud._features_dc.detach().numpy().reshape(-1, 3), axis=0)
    assert np.array_equal(a, b)


def _walk_chains_loop(pairs, n_ep, end_distance):
    """Edge-by-edge walk (the form the reference uses, scene/hair_gaussian_model.py:1410-1498): checker for the
    vectorised scene.hair_gaussian_model.walk_chains."""
    id_to_strand = -np.ones(n_ep, np.int32)
    complementary = -np.ones(n_ep, np.int32)
    flat = pairs.reshape(-1)
    order = np.argsort(flat, kind="stable")
    ids_sorted = flat[order]
    first = np.r_[True, ids_sorted[1:] != ids_sorted[:-1]]
    inc = -np.ones((n_ep, 2), np.int64)
    inc[ids_sorted[first], 0] = order[first] // 2
    inc[ids_sorted[~first], 1] = order[~first] // 2
    counts = np.bincount(flat, minlength=n_ep)
    visited = np.zeros(n_ep, bool)
    strands, strands_rows = [], []
    for start in np.nonzero(counts == 1)[0]:
        if visited[start]:
            continue
        cur, row = start, inc[start, 0]
        seq, seq_rows = [], []
        sid = len(strands)
        while row != -1:
            id_to_strand[cur] = sid
            a, b = pairs[row]
            nxt = a if a != cur else b
            seq.append((cur, nxt))
            seq_rows.append(row)
            cur = nxt
            r0, r1 = inc[cur]
            row = r0 if r0 != row else r1
        id_to_strand[cur] = sid
        visited[start] = visited[cur] = True
        complementary[start], complementary[cur] = cur, start
        seq, seq_rows = np.asarray(seq, np.int64), np.asarray(seq_rows, np.int64)
        d = end_distance(np.array([start, cur]))
        if d[0] > d[1]:
            seq, seq_rows = seq[::-1, ::-1].copy(), seq_rows[::-1].copy()
        strands.append(seq)
        strands_rows.append(seq_rows)
    return strands, strands_rows, id_to_strand, complementary


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_walk_chains_matches_edge_by_edge_walk(seed):
    from scene.hair_gaussian_model import walk_chains
    rng = np.random.default_rng(seed)
    n_ep = 700
    ids = rng.permutation(n_ep)[:620]                     # some endpoint ids stay unused
    pairs, pos = [], 0
    while pos + 1 < ids.size:
        ln = int(rng.integers(1, 40))                     # chains of 1 .. 39 edges
        chain = ids[pos:pos + ln + 1]
        pos += ln + 1
        for a, b in zip(chain[:-1], chain[1:]):
            pairs.append((a, b) if rng.random() < 0.5 else (b, a))
    if seed == 2:                                         # a closed loop has no end: left out by both
        loop = np.arange(n_ep, n_ep + 5)
        pairs += [(loop[i], loop[(i + 1) % 5]) for i in range(5)]
        n_ep += 5
    pairs = np.asarray(pairs, np.int64)[rng.permutation(len(pairs))]
    dist = rng.random(n_ep)
    ref = _walk_chains_loop(pairs, n_ep, lambda e: dist[e])
    i2s, comp = -np.ones(n_ep, np.int32), -np.ones(n_ep, np.int32)
    from scene.hair_gaussian_model import StrandsInfo
    info = StrandsInfo(*walk_chains(pairs, n_ep, i2s, comp, lambda e: dist[e]), i2s, comp)
    ls, lr = info.list_strands, info.list_strands_segments_id
    assert len(ls) == len(ref[0]) > 10
    for a, b, ra, rb in zip(ls, lr, ref[0], ref[1]):
        np.testing.assert_array_equal(a, ra)
        np.testing.assert_array_equal(b, rb)
    np.testing.assert_array_equal(i2s, ref[2])
    np.testing.assert_array_equal(comp, ref[3])
    # the torch form (what a model on the GPU runs), here on CPU tensors
    import torch
    from scene.hair_gaussian_model import walk_chains_torch
    dist_t = torch.from_numpy(dist)
    o, r, sr, i2s_t, comp_t = walk_chains_torch(torch.from_numpy(pairs), n_ep, lambda e: dist_t[e])
    info_t = StrandsInfo(o.numpy(), r.numpy(), sr.numpy(), i2s_t.numpy(), comp_t.numpy())
    for a, b, ra, rb in zip(info_t.list_strands, info_t.list_strands_segments_id, ref[0], ref[1]):
        np.testing.assert_array_equal(a, ra)
        np.testing.assert_array_equal(b, rb)
    np.testing.assert_array_equal(i2s_t.numpy(), ref[2])
    np.testing.assert_array_equal(comp_t.numpy(), ref[3])


def test_nearest_distance_matches_kdtree():
    import torch
    from scipy.spatial import cKDTree
    from scene.hair_gaussian_model import nearest_distance
    rng = np.random.default_rng(3)
    pts, refs = rng.normal(size=(5000, 3)).astype(np.float32), rng.normal(size=(300, 3)).astype(np.float32)
    d = nearest_distance(torch.from_numpy(pts), torch.from_numpy(refs), chunk=1024).numpy()
    ref = cKDTree(refs.astype(np.float64)).query(pts.astype(np.float64), k=1)[0]
    np.testing.assert_allclose(d, ref, rtol=1e-14, atol=0)


def _magnet_by_loops(model):
    """strand_joints_magnet_loss of the reference (loss/losses.py:106-172) as plain loops over the strand ends, float64
    distances for the neighbour ORDER only; returns (loss, gradient w.r.t. the endpoints) with knn_points' gradient."""
    ep = model._endpoints.detach().numpy().astype(np.float32)
    pairs = model.endpoint_pairs.numpy()
    ids, counts = np.unique(pairs, return_counts=True)
    ends = [int(i) for i in ids[counts == 1]]
    partner = {}
    for a, b in pairs:
        partner.setdefault(int(a), int(b))
        partner.setdefault(int(b), int(a))
    mapping = np.zeros(ep.shape[0], np.int64)
    for e in ends:
        mapping[e] = partner[e]
    ends = [e for e in ends if np.linalg.norm(ep[e] - ep[partner[e]]) > model.min_val]
    comp = [partner[e] for e in ends]
    pts = ep[ends]
    n = len(ends)
    grad = np.zeros_like(ep, dtype=np.float64)
    vals = []
    for i in range(n):
        d = pts[i][None, :] - pts
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]).astype(np.float32)
        order = np.lexsort((np.arange(n), d2))[:3]
        j = order[1] if (order[1] != i and order[1] != comp[i]) else order[2]     # (local index vs GLOBAL id: as the reference)
        nn_dir = ep[j] - ep[mapping[j]]                                             # (local index into the global table: same)
        if not np.linalg.norm(nn_dir) > model.min_val:
            continue
        vals.append((i, j, float(d2[j])))
    m = len(vals)
    loss = sum(v * v for _, _, v in vals) / m
    for i, j, v in vals:
        g = (2.0 * v / m) * 2.0 * (pts[i] - pts[j]).astype(np.float64)
        grad[ends[i]] += g
        grad[ends[j]] -= g
    return loss, grad


def test_magnet_loss_matches_the_reference_statement_by_statement():
    """loss/losses.py::strand_joints_magnet_loss (reference :106-172) against a loop restatement, value and gradient;
    one collapsed end segment is excluded like in the reference."""
    from loss.losses import strand_joints_magnet_loss
    m, _ = _strands(S=40, V=6, seed=3)
    with torch.no_grad():
        m._endpoints[5] = m._endpoints[4]          # end segment of strand 0 collapsed (vertices 4, 5 of 0..5)
        m._endpoints.mul_(30.0)                    # (distances of order 1: the fourth power stays well inside fp32)
    m.compute_strands_info(only_foreground=False)
    loss = strand_joints_magnet_loss(m)
    loss.backward()
    ref_loss, ref_grad = _magnet_by_loops(m)
    assert abs(float(loss) - ref_loss) <= 1e-5 * abs(ref_loss)
    got = m._endpoints.grad.numpy()
    assert np.abs(got - ref_grad).max() <= 1e-4 * np.abs(ref_grad).max()
    # only strand ends receive a gradient, and the collapsed end does not take part
    ids, counts = np.unique(m.endpoint_pairs.numpy(), return_counts=True)
    inner = np.setdiff1d(np.arange(got.shape[0]), ids[counts == 1])
    assert np.all(got[inner] == 0) and np.all(got[5] == 0)


def test_loss_function_with_magnet_term_runs_on_cpu_model():
    from loss.losses import angle_smoothness_loss, strand_joints_magnet_loss
    m, _ = _strands(S=10, V=8, seed=1)
    m.compute_strands_info(only_foreground=False)
    total = angle_smoothness_loss(m) + 0.1 * strand_joints_magnet_loss(m)
    total.backward()
    assert torch.isfinite(m._endpoints.grad).all()
