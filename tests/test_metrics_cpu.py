"""Strand metrics restatement vs a brute-force evaluation of the same definition (loss/metrics.py:12-85)."""
import numpy as np


def _brute(p1, p2, dist_th, angle_th, bidir, s1=None, s2=None):
    cos_th = np.cos(np.deg2rad(angle_th))
    count, stats = 0, {}
    for i in range(len(p1[0])):
        d = np.linalg.norm(p2[0] - p1[0][i], axis=1)
        near = np.nonzero(d <= dist_th)[0]
        if s1 is not None:
            st = stats.setdefault(int(s1[i]), {"m": [], "n": 0})
            st["n"] += 1
        if len(near):
            dot = p2[1][near] @ p1[1][i]
            if bidir:
                dot = np.abs(dot)
            ok = dot >= cos_th
            if ok.any():
                count += 1
                if s1 is not None:
                    st["m"].extend(np.unique(s2[near[ok]]).tolist())
    ratio = count / len(p1[0])
    cons = None
    if s1 is not None:
        tot = 0.0
        for v in stats.values():
            if v["m"]:
                _, c = np.unique(np.array(v["m"]), return_counts=True)
                tot += c.max() / v["n"]
        cons = tot / len(stats)
    return ratio, cons


def test_metrics_match_bruteforce():
    from loss.metrics import HairEvalData, compute_metrics
    rng = np.random.default_rng(0)
    n = 400
    gt_p = rng.uniform(0, 0.03, (n, 3))
    gt_d = rng.normal(size=(n, 3)); gt_d /= np.linalg.norm(gt_d, axis=1, keepdims=True)
    pr_p = gt_p + rng.normal(size=(n, 3)) * 0.0015
    pr_d = gt_d + rng.normal(size=(n, 3)) * 0.4; pr_d /= np.linalg.norm(pr_d, axis=1, keepdims=True)
    sg, sp = np.repeat(np.arange(n // 20), 20), np.repeat(np.arange(n // 10), 10)
    gt, pred = HairEvalData(gt_p, gt_d, sg), HairEvalData(pr_p, pr_d, sp)
    for bidir in (False, True):
        m, labels = compute_metrics(pred, gt, bidirectional=bidir)
        sfx = "(b)" if bidir else ""
        assert labels[0] == "0.002m&20°" and len(labels) == 4
        for i, (d, a) in enumerate(zip((2e-3, 3e-3, 4e-3, 4e-3), (20, 30, 40, 90))):
            p, _ = _brute((pr_p, pr_d), (gt_p, gt_d), d, a, bidir)
            r, c = _brute((gt_p, gt_d), (pr_p, pr_d), d, a, bidir, sg, sp)
            assert abs(m["precision" + sfx][i] - p) < 1e-12 and abs(m["recall" + sfx][i] - r) < 1e-12
            assert abs(m["strand_consistency" + sfx][i] - c) < 1e-12
            f1 = 2 * p * r / (p + r) if p + r > 0 else 0
            assert abs(m["f1" + sfx][i] - f1) < 1e-12


# ---- the reference's own evaluation code, executed on CPU (tests/golden/make_ref_metrics_pins.py) -------------------------------
import os

import pytest

_PINS = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_metrics_pins.npz"))
_NAMES = ("precision", "recall", "f1", "strand_consistency")


@pytest.mark.parametrize("ci", [int(c) for c in _PINS["meta_cases"]])
def test_compute_metrics_equals_the_reference_run(ci):
    """compute_metrics (reference loss/metrics.py:88-173) on random oriented points in strands: precision / recall / F1 /
    strand consistency at the four default threshold pairs, both settings of `bidirectional` (ratios of integer counts:
    equal to the last bit up to the order of a float sum)."""
    from loss.metrics import HairEvalData, compute_metrics
    k = f"rand{ci}_"
    gt = HairEvalData(_PINS[k + "gt_points"], _PINS[k + "gt_dirs"], _PINS[k + "gt_strand"])
    pred = HairEvalData(_PINS[k + "pred_points"], _PINS[k + "pred_dirs"], _PINS[k + "pred_strand"])
    for bidir in (False, True):
        m, labels = compute_metrics(pred, gt, bidirectional=bidir)
        assert labels == [str(t) for t in _PINS["meta_thresholds"]]
        sfx = "(b)" if bidir else ""
        want = _PINS[k + f"metrics_b{int(bidir)}"]
        for j, nm in enumerate(_NAMES):
            assert np.allclose(m[nm + sfx], want[j], rtol=1e-12, atol=1e-15), (nm, bidir)


def _strand_model(tag):
    import torch
    from arguments import OptimizationParams
    from scene.hair_gaussian_model import HairGaussianModel
    topo = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_topology_pins.npz"))
    k = f"{tag}_"
    m = HairGaussianModel(sh_degree=3, device="cpu")
    m.ref_strand_root = topo[k + "ref_strand_root"]
    m.strand_root_endpoint_idx = torch.from_numpy(topo[k + "root_idx"])
    m.endpoint_pairs = torch.from_numpy(topo[k + "pairs"])
    P = lambda a: torch.nn.Parameter(torch.from_numpy(a.copy()).requires_grad_(True))
    m._endpoints, m._features_dc, m._features_rest = P(topo[k + "endpoints"]), P(topo[k + "f_dc"]), P(topo[k + "f_rest"])
    m._opacity, m._mask, m._width = P(topo[k + "opacity"]), P(topo[k + "mask"]), P(topo[k + "width"])
    m.training_setup(OptimizationParams())
    m.compute_strands_info()
    return m


def test_eval_data_of_a_strand_model_equals_the_reference_run():
    """compute_eval_data_from_hair_gs (reference data/eval_data.py:133-171): the joints in strand order, their unit directions,
    strand ids and edges, with and without the foreground filter -- and the metrics of one model against another."""
    from loss.metrics import compute_eval_data_from_hair_gs, compute_metrics
    evs = {}
    for tag in [str(t) for t in _PINS["meta_eval_tags"]]:
        m = _strand_model(tag)
        for edges, fg in ((False, False), (True, False), (True, True)):
            ev = compute_eval_data_from_hair_gs(m, compute_edges=edges, only_foreground=fg)
            k = f"eval_{tag}_e{int(edges)}f{int(fg)}_"
            assert np.array_equal(ev.points, _PINS[k + "points"]) and np.array_equal(ev.directions, _PINS[k + "dirs"]), (tag, edges, fg)
            assert np.array_equal(np.asarray(ev.points_id_to_strand_id), _PINS[k + "strand"])
            if edges:
                assert np.array_equal(np.asarray(ev.edges), _PINS[k + "edges"])
            else:
                assert ev.edges is None
        evs[tag] = compute_eval_data_from_hair_gs(m)
    for bidir in (False, True):
        got, _ = compute_metrics(pred=evs["c0"], gt=evs["c2"], bidirectional=bidir, dist_ths=[2e-3, 4e-3, 8e-3, 2e-2], angle_ths=[20, 30, 40, 90])
        want = _PINS[f"eval_c0_vs_c2_metrics_b{int(bidir)}"]
        sfx = "(b)" if bidir else ""
        for j, nm in enumerate(_NAMES):
            assert np.allclose(got[nm + sfx], want[j], rtol=1e-12, atol=1e-15), (nm, bidir)
