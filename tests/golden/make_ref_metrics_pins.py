"""Generates tests/golden/ref_metrics_pins.npz by RUNNING the reference's evaluation code on the CPU of the authoring container:
  * loss/metrics.py::compute_metrics (:88-173, through pct_matched_points :12-85; its own Pool of 8 processes) on random
    oriented point clouds arranged in strands, both settings of `bidirectional`;
  * data/eval_data.py::compute_eval_data_from_hair_gs (:133-171; with and without compute_edges / only_foreground) on the
    reference's HairGaussianModel (device="cpu") holding the cut-up-curve states of ref_topology_pins.npz, and the metrics of
    one such model against another.
Only numeric inputs and outputs are stored.  Absent third-party imports: tests/golden/_ref_harness.py; the reference's `loss`,
`scene` and `data` packages are entered without running their __init__.py (dataset readers), and `data.HairEvalData` is bound
to data/eval_data.py's class, as data/__init__.py's star import would.
"""
import argparse
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "ref_metrics_pins.npz")
CASES = [0, 1, 2]


def main():
    sys.path.insert(0, HERE)
    from _ref_harness import REF, enter_reference
    enter_reference()
    pkg = types.ModuleType("data")
    pkg.__path__ = [os.path.join(REF, "data")]
    sys.modules["data"] = pkg
    import data.eval_data as ED
    pkg.HairEvalData = ED.HairEvalData
    import torch
    from arguments import OptimizationParams
    from loss import metrics as RM
    from scene.hair_gaussian_model import HairGaussianModel
    opt = OptimizationParams(argparse.ArgumentParser())
    out = {"meta_cases": np.array(CASES)}
    rng = np.random.default_rng(7)
    names = ("precision", "recall", "f1", "strand_consistency")
    # ---- random oriented points in strands
    for ci in CASES:
        n = 300 + 60 * ci
        gt_p = rng.uniform(0, 0.03, (n, 3))
        gt_d = rng.normal(size=(n, 3)); gt_d /= np.linalg.norm(gt_d, axis=1, keepdims=True)
        pr_p = gt_p + rng.normal(size=(n, 3)) * 0.0015
        pr_d = gt_d + rng.normal(size=(n, 3)) * 0.4; pr_d /= np.linalg.norm(pr_d, axis=1, keepdims=True)
        sg, sp = np.repeat(np.arange(n // 20), 20), np.repeat(np.arange(n // 10), 10)
        k = f"rand{ci}_"
        out[k + "gt_points"], out[k + "gt_dirs"], out[k + "gt_strand"] = gt_p, gt_d, sg
        out[k + "pred_points"], out[k + "pred_dirs"], out[k + "pred_strand"] = pr_p, pr_d, sp
        gt, pred = ED.HairEvalData(gt_p, gt_d, sg, None), ED.HairEvalData(pr_p, pr_d, sp, None)
        for bidir in (False, True):
            m, th = RM.compute_metrics(pred=pred, gt=gt, bidirectional=bidir)
            sfx = "(b)" if bidir else ""
            out[k + f"metrics_b{int(bidir)}"] = np.stack([np.asarray(m[nm + sfx], dtype=np.float64) for nm in names])
            out["meta_thresholds"] = np.array(th)
    # ---- eval data of strand models
    topo = np.load(os.path.join(HERE, "ref_topology_pins.npz"))

    def model(tag):
        k = f"{tag}_"
        m = HairGaussianModel(sh_degree=3, device="cpu")
        m.ref_strand_root = topo[k + "ref_strand_root"]
        m.strand_root_endpoint_idx = torch.from_numpy(topo[k + "root_idx"])
        m.endpoint_pairs = torch.from_numpy(topo[k + "pairs"])
        P = lambda a: torch.nn.Parameter(torch.from_numpy(a.copy()).requires_grad_(True))
        m._endpoints, m._features_dc, m._features_rest = P(topo[k + "endpoints"]), P(topo[k + "f_dc"]), P(topo[k + "f_rest"])
        m._opacity, m._mask, m._width = P(topo[k + "opacity"]), P(topo[k + "mask"]), P(topo[k + "width"])
        m.training_setup(opt)
        m.compute_strands_info()
        return m

    evs = {}
    for tag in ("c0", "c2", "c4"):
        m = model(tag)
        for edges, fg in ((False, False), (True, False), (True, True)):
            ev = ED.compute_eval_data_from_hair_gs(m, compute_edges=edges, only_foreground=fg)
            k = f"eval_{tag}_e{int(edges)}f{int(fg)}_"
            out[k + "points"], out[k + "dirs"], out[k + "strand"] = ev.points, ev.directions, np.asarray(ev.points_id_to_strand_id)
            out[k + "edges"] = np.asarray(ev.edges) if ev.edges is not None else np.zeros((0, 2), np.int32)
        evs[tag] = ED.compute_eval_data_from_hair_gs(m)
    out["meta_eval_tags"] = np.array(["c0", "c2", "c4"])
    for bidir in (False, True):
        m, _ = RM.compute_metrics(pred=evs["c0"], gt=evs["c2"], bidirectional=bidir, dist_ths=[2e-3, 4e-3, 8e-3, 2e-2], angle_ths=[20, 30, 40, 90])
        sfx = "(b)" if bidir else ""
        out[f"eval_c0_vs_c2_metrics_b{int(bidir)}"] = np.stack([np.asarray(m[nm + sfx], dtype=np.float64) for nm in names])
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT} ({os.path.getsize(OUT) / 1024:.0f} KB, {len(out)} arrays)")


if __name__ == "__main__":
    main()
