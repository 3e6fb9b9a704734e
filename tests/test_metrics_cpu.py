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
