"""Strand-reconstruction metrics (counterpart of the reference's loss/metrics.py:12-173): precision / recall / F1 of
oriented points under (distance, angle) thresholds and strand consistency.  CPU code (scipy cKDTree), a reported
baseline of the path, not an optimisation target.  Instead of the reference's Python loop over
every point, the radius-query result is flattened to CSR arrays and reduced with numpy; the threshold pairs run on a
thread pool (cKDTree releases the GIL) instead of 8 forked processes with a Manager dict."""
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass
from typing import Optional

import numpy as np
from scipy.spatial import cKDTree


@dataclass
class HairEvalData:
    """Oriented point cloud: points [N,3], unit directions [N,3], optional strand id per point, optional edges between points
    (reference data/eval_data.py:16-20)."""
    points: np.ndarray
    directions: np.ndarray
    points_id_to_strand_id: Optional[np.ndarray] = None
    edges: Optional[np.ndarray] = None


def _csr_matches(p1, p2, dist_th, cos_th, bidirectional):
    """For every point of p1: p2 indices within dist_th whose direction agrees.  Returns (row_of_match, p2_index)."""
    lists = cKDTree(p2.points).query_ball_point(p1.points, r=dist_th, workers=-1)
    lens = np.fromiter((len(x) for x in lists), dtype=np.int64, count=len(lists))
    if lens.sum() == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    rows = np.repeat(np.arange(len(lists)), lens)
    cols = np.fromiter((j for x in lists for j in x), dtype=np.int64, count=int(lens.sum()))
    dot = np.einsum("ij,ij->i", p1.directions[rows], p2.directions[cols])
    if bidirectional:
        dot = np.abs(dot)
    keep = dot >= cos_th
    return rows[keep], cols[keep]


def pct_matched_points(p1, p2, dist_th, angle_th, bidirectional=False, compute_strand_consistency=False):
    """Fraction of p1's points that have a p2 point within dist_th with direction within angle_th; optionally the
    strand consistency: per p1 strand, the largest share of its points matched to one p2 strand, averaged
    (reference :12-85).  Returns (ratio, strand_consistency or None)."""
    cos_th = np.cos(np.deg2rad(angle_th))
    n = p1.points.shape[0]
    rows, cols = _csr_matches(p1, p2, dist_th, cos_th, bidirectional)
    matched = np.zeros(n, bool)
    matched[rows] = True
    ratio = matched.sum() / n
    consistency = None
    if compute_strand_consistency:
        s1 = p1.points_id_to_strand_id
        s2 = p2.points_id_to_strand_id
        strands, n_pts = np.unique(s1, return_counts=True)
        # each p1 point votes once for every distinct p2 strand it matched
        votes = np.unique(np.stack([rows, s2[cols]], 1), axis=0) if rows.size else np.zeros((0, 2), np.int64)
        total = 0.0
        if votes.shape[0]:
            key = np.stack([s1[votes[:, 0]], votes[:, 1]], 1)          # (p1 strand, p2 strand)
            uk, cnt = np.unique(key, axis=0, return_counts=True)
            best = {}
            for (a, _), c in zip(uk, cnt):
                best[a] = max(best.get(a, 0), c)
            size = dict(zip(strands.tolist(), n_pts.tolist()))
            total = sum(c / size[a] for a, c in best.items())
        consistency = total / len(strands)
    return ratio, consistency


def compute_metrics(pred, gt, dist_ths=(2e-3, 3e-3, 4e-3, 4e-3), angle_ths=(20, 30, 40, 90),
                    metrics=("precision", "recall", "f1", "strand_consistency"), bidirectional=False, processes=None):
    """Returns ({metric[(b)]: array over thresholds}, [threshold labels]) like the reference (:88-173)."""
    consistency = ("strand_consistency" in metrics and pred.points_id_to_strand_id is not None
                   and gt.points_id_to_strand_id is not None)
    labels = [f"{d}m&{a}°" for d, a in zip(dist_ths, angle_ths)]
    jobs = []
    if "precision" in metrics:
        jobs += [("precision", i, pred, gt, d, a, False) for i, (d, a) in enumerate(zip(dist_ths, angle_ths))]
    if "recall" in metrics:
        jobs += [("recall", i, gt, pred, d, a, consistency) for i, (d, a) in enumerate(zip(dist_ths, angle_ths))]
    out = {m: {} for m in metrics}
    with ThreadPoolExecutor(max_workers=8 if processes is None else processes) as pool:
        futs = [(name, i, pool.submit(pct_matched_points, a, b, d, ang, bidirectional, cs))
                for name, i, a, b, d, ang, cs in jobs]
        for name, i, f in futs:
            ratio, cons = f.result()
            out[name][i] = ratio
            if cons is not None:
                out["strand_consistency"][i] = cons
    if "f1" in out and "precision" in out and "recall" in out:
        for i in range(len(labels)):
            p, r = out["precision"].get(i), out["recall"].get(i)
            if p is not None and r is not None:
                out["f1"][i] = 2 * p * r / (p + r) if p + r > 0 else 0
    suffix = "(b)" if bidirectional else ""
    return ({k + suffix: np.array([v[i] for i in range(len(labels)) if i in v]) for k, v in out.items()}, labels)


def compute_eval_data_from_hair_gs(hair_gs, compute_edges=False, only_foreground=False):
    """Oriented points of a strand model as the reference defines them (data/eval_data.py:133-171): one point per segment of
    `strands_info.list_strands` -- its FIRST joint in strand order, root to tip --, the unit direction towards the next joint,
    and the joint's strand id.  (`strands_info` computed with only_foreground=True already holds foreground segments only;
    only_foreground=True here filters again by the current foreground mask, like the reference.)"""
    endpoints = hair_gs._endpoints.detach().cpu().numpy()
    if hair_gs.strands_info is None:          # (the constructor's default: the reference would fail here; train.evaluate() may come first)
        hair_gs.compute_strands_info(only_foreground=True)
    strands = list(hair_gs.strands_info.list_strands)
    if len(strands) == 0:                     # nothing to evaluate: an empty point set instead of np.concatenate's error
        return HairEvalData(points=np.zeros((0, 3), endpoints.dtype), directions=np.zeros((0, 3), endpoints.dtype),
                            points_id_to_strand_id=np.zeros((0,), np.int64), edges=np.zeros((0, 2), np.int32) if compute_edges else None)
    segments_id = np.concatenate(strands, axis=0)
    if only_foreground:
        mask = hair_gs.compute_foreground_mask().cpu().numpy()
        line_points = hair_gs.endpoint_pairs.cpu().numpy()[mask].flatten()
        segments_id = segments_id[np.any(np.isin(segments_id, line_points), axis=1)]
    segments = endpoints[segments_id]
    directions = segments[:, 1] - segments[:, 0]
    directions /= np.linalg.norm(directions, axis=1, keepdims=True)
    points_id = segments_id[:, 0]
    edges = None
    if compute_edges:   # indices into the new point set; single-segment strands have no edge (:160-166)
        mapping = np.zeros(segments_id.max() + 1, dtype=np.int32)
        mapping[segments_id[:, 0]] = np.arange(segments_id.shape[0])
        u, c = np.unique(segments_id, return_counts=True)
        edges = mapping[segments_id[np.isin(segments_id[:, 1], u[c > 1])]]
    return HairEvalData(points=endpoints[points_id], directions=directions,
                        points_id_to_strand_id=np.asarray(hair_gs.strands_info.id_to_strand_id)[points_id], edges=edges)


def compute_eval_data_from_gs(gaussians):
    """Oriented points of a Gaussian cloud: foreground means + the direction of the longest axis (reference
    data/eval_data.py:121-130)."""
    import torch
    with torch.no_grad():
        fg = gaussians.compute_foreground_mask()
        return HairEvalData(points=gaussians.get_xyz[fg].cpu().numpy(), directions=gaussians.get_orientation[fg].cpu().numpy())

