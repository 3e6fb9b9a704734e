"""distCUDA2 parity (GPU): exact 3-NN, bit-exact against the oracle (and therefore the brute force)."""
import numpy as np
import pytest

from oracle import hgs_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("P", [1, 3, 4, 5, 100, 1024, 1025, 4096, 4097, 20000, 70001])
def test_dist2_bit_exact(P):
    import torch
    from simple_knn._C import distCUDA2
    rng = np.random.default_rng(P)
    pts = (rng.normal(size=(P, 3)) * 0.2 + np.array([0.3, -0.1, 0.8])).astype(np.float32)
    if P >= 100:
        pts[: P // 10] = pts[P // 10: 2 * (P // 10)]  # duplicates -> zero distances, Morton ties
    got = distCUDA2(torch.from_numpy(pts).cuda()).cpu().numpy()
    ref = O.dist2(pts)
    if P < 4:  # fewer than 3 neighbours: FLT_MAX sums overflow to inf in both
        assert np.array_equal(np.isinf(got), np.isinf(ref))
        return
    np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_dist2_empty_and_errors():
    import torch
    from simple_knn._C import distCUDA2
    assert distCUDA2(torch.empty((0, 3), device="cuda")).shape == (0,)
    with pytest.raises(Exception):
        distCUDA2(torch.zeros((4, 3)))  # CPU tensor: no CPU path


def test_radius_pairs_match_kdtree_query_pairs():
    """hgs_radius_pairs (GPU candidate search of strand merging) against scipy cKDTree.query_pairs + direction test."""
    import torch
    from scipy.spatial import cKDTree
    from scene.hair_topology import HairTopologyMixin
    rng = np.random.default_rng(5)
    for n, r in ((1, 0.1), (257, 0.08), (3000, 0.02)):
        pos = rng.random((n, 3)).astype(np.float32)
        d = rng.normal(size=(n, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        for bidir in (False, True):
            a, b = HairTopologyMixin._radius_pairs_gpu(torch.from_numpy(pos).cuda(), torch.from_numpy(d).cuda(), r, 0.3, bidir) \
                if n > 1 else (np.zeros(0, np.int64), np.zeros(0, np.int64))
            ref = cKDTree(pos.astype(np.float64)).query_pairs(r=r, output_type="ndarray") if n > 1 else np.zeros((0, 2), np.int64)
            dot = -(d[ref[:, 0]] * d[ref[:, 1]]).sum(1)
            dot = np.abs(dot) if bidir else dot
            ref = ref[dot >= 0.3]
            got = set(zip(a.tolist(), b.tolist()))
            want = set(zip(ref[:, 0].tolist(), ref[:, 1].tolist()))
            # pairs whose distance / cosine sits within rounding of the thresholds may differ (fp32 vs fp64)
            dist = lambda p: np.linalg.norm(pos[p[0]].astype(np.float64) - pos[p[1]])
            edge = lambda p: abs(dist(p) - r) < 1e-6 or abs(abs(float(-(d[p[0]] * d[p[1]]).sum())) - 0.3) < 1e-5
            assert all(edge(p) for p in got ^ want), (n, bidir, len(got ^ want))
            assert len(got) > 0 or n < 100


def test_merge_candidates_gpu_model_equals_cpu_model():
    import torch
    from arguments import OptimizationParams
    from scene.hair_gaussian_model import HairGaussianModel
    rng = np.random.default_rng(2)
    x = np.linspace(0, 0.02, 6)
    strands = []
    for k in range(40):       # chains of 3 collinear strands 1 mm apart, scattered in space
        o = rng.random(3) * 0.5
        strands += [np.stack([x + j * 0.021, np.zeros(6), np.zeros(6)], 1) + o for j in range(3)]
    pts = np.stack(strands).astype(np.float32)
    roots = pts[::3, 0]
    out = {}
    for dev in ("cpu", "cuda"):
        m = HairGaussianModel.from_strands(pts, device=dev, ref_strand_root=roots)
        m.training_setup(OptimizationParams())
        m.compute_strands_info()
        out[dev] = m.compute_endpoint_pair_to_merge().cpu().numpy()
    assert out["cpu"].shape[0] >= 40
    assert sorted(map(tuple, np.sort(out["cpu"], 1).tolist())) == sorted(map(tuple, np.sort(out["cuda"], 1).tolist()))


@pytest.mark.parametrize("n", [1, 2, 3, 255, 256, 257, 5000])
def test_knn3_matches_brute_force_exactly(n):
    """hgs_knn3 (the neighbour search of the magnet loss, pytorch3d knn_points(p, p, K=3) in the reference,
    loss/losses.py:139-144): indices and squared distances bit for bit against a float32 brute force with the same
    expression and (distance, index) order; duplicates included."""
    import torch
    from loss.losses import knn3_self
    rng = np.random.default_rng(n)
    pts = rng.normal(size=(n, 3)).astype(np.float32)
    if n >= 255:
        pts[:20] = pts[20:40]                     # exact duplicates: zero distances, index order decides
    d2g, idg = knn3_self(torch.from_numpy(pts).cuda())
    d2c, idc = knn3_self(torch.from_numpy(pts))   # chunked distance matrix on the CPU: same expression, stable sort
    np.testing.assert_array_equal(idg.cpu().numpy(), idc.numpy())
    np.testing.assert_array_equal(d2g.cpu().numpy().view(np.uint32), d2c.numpy().view(np.uint32))
    if n >= 3:
        assert np.all(d2c.numpy()[:, 0] == 0)   # the point itself (or a duplicate with a smaller index) comes first


def test_magnet_loss_gpu_model_equals_cpu_model():
    import torch
    from loss.losses import strand_joints_magnet_loss
    from scene.hair_gaussian_model import HairGaussianModel
    from synthetic import strand_polylines
    pts = strand_polylines(300, 12, seed=4) * 30.0
    out = {}
    for dev in ("cpu", "cuda"):
        m = HairGaussianModel.from_strands(pts, device=dev)
        m.compute_strands_info(only_foreground=False)
        loss = strand_joints_magnet_loss(m)
        loss.backward()
        out[dev] = (float(loss), m._endpoints.grad.cpu().numpy())
    assert abs(out["cpu"][0] - out["cuda"][0]) <= 1e-5 * abs(out["cpu"][0])
    assert np.abs(out["cpu"][1] - out["cuda"][1]).max() <= 1e-5 * np.abs(out["cpu"][1]).max()


def test_training_step_with_magnet_term():
    """--lambda_magnet 0.1 works on the reference (loss/losses.py:352-354): here it selects the op-by-op single-pass
    iteration (the fused iteration covers the default terms) and trains."""
    import torch
    from arguments import OptimizationParams
    from synthetic import build_workload
    from train import fused_step_applicable, training_step
    model, cams, extent = build_workload("tiny", device="cuda", with_targets=True)
    opt = OptimizationParams()
    opt.enable_topology = False
    opt.lambda_magnet = 0.1
    model.training_setup(opt)
    assert not fused_step_applicable(model, opt)
    bg = torch.zeros(3, device="cuda")
    for it in range(1, 4):
        loss, terms, _ = training_step(model, cams[it % len(cams)], opt, bg, it, extent=extent)
        assert "magnet" in terms and torch.isfinite(loss) and torch.isfinite(terms["magnet"])


def test_c1_merge_at_size_gpu_model_equals_cpu_model():
    """BASELINE.json C1 at its stated size on a GPU model (candidate search: hgs_radius_pairs; strand walk: pointer
    doubling on the device) against the CPU model (cKDTree; numpy walk): the same strands, vertex for vertex, in the same
    number of rounds (reference merge.py:114-190)."""
    from tests.test_models_cpu import _c1_merge
    h_cpu, r_cpu, s_cpu, _ = _c1_merge("cpu")
    h_gpu, r_gpu, s_gpu, _ = _c1_merge("cuda")
    assert r_cpu == r_gpu and h_gpu.strands_info.n_strands == h_cpu.strands_info.n_strands == 50
    assert np.abs(np.asarray(s_cpu) - np.asarray(s_gpu)).max() <= 2e-5
    assert h_gpu._endpoints.is_cuda and h_gpu.get_xyz.shape[0] == 1000


@pytest.mark.parametrize("N,M", [(1, 1), (1000, 7), (70001, 1000), (300, 2500)])
def test_nearest_distance_f64_matches_the_tree(N, M):
    """hgs_nearest_distance_f64 (what orients every strand root -> tip, scene/hair_gaussian_model.py:1466-1470) against
    scipy cKDTree(refs).query(points)[0] and the float64 brute force: the same distances to the last bit of the brute force,
    1 ulp of the tree's."""
    import torch
    from scipy.spatial import cKDTree
    from scene.hair_gaussian_model import nearest_distance
    rng = np.random.default_rng(N + M)
    pts = rng.normal(size=(N, 3)).astype(np.float32)
    refs = rng.normal(size=(M, 3))
    if N >= 1000:
        pts[:50] = refs[:50 % M or 1][0].astype(np.float32)       # points (nearly) on a reference
    got = nearest_distance(torch.from_numpy(pts).cuda(), torch.from_numpy(refs).cuda()).cpu().numpy()
    d = pts.astype(np.float64)[:, None, :] - refs[None, :, :] if N * M <= 4_000_000 else None
    tree = cKDTree(refs).query(pts.astype(np.float64), k=1)[0]
    assert np.abs(got - tree).max() <= 4e-16 * max(1.0, np.abs(tree).max())
    if d is not None:
        brute = np.sqrt((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]).min(axis=1)
        np.testing.assert_array_equal(got, brute)

