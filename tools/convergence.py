#!/usr/bin/env python3
"""Convergence evidence (SURVEY.md 8d 'PSNR': training-curve PSNR against the synthetic ground truth at fixed iteration
counts; loop: reference train.py:133-204).

A strand workload with CONSISTENT targets -- ground-truth image, mask and orientation field all rendered from ONE
perturbed copy of the strands (synthetic.attach_targets(consistent=True): the orientation target is the projected
direction of the ground-truth strands, confidence 1) -- is trained from a state that is off in geometry (the unperturbed
strands) AND in appearance (colours and opacities shifted), once per structure of the iteration:

  fused_graph   the default: fused strand iteration, single 7-channel raster pass, HIP-graph replay, 8 steps per launch
  op_by_op      getters / render_multi / loss_function_single_pass as separate autograd ops, graph replay
  three_pass    the reference's structure: three render() calls per iteration, eager, blocking forward

and the mean PSNR of ALL views against their targets is recorded at iterations 0 / 500 / 1000 / 3000 (topology operators
off: the three trajectories optimise the same parameters and can be compared number by number).  Two more runs repeat
fused_graph and op_by_op WITH the topology operators (densification, merging, opacity reset at iteration 3000).  Prints one JSON object
(-> profiles/r03_convergence.json).

  python tools/convergence.py [workload=north_star] [views=8] [checkpoints=0,500,1000,3000]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import torch  # noqa: E402

import hgs_runtime as rt  # noqa: E402
from arguments import OptimizationParams  # noqa: E402
from gaussian_renderer import render  # noqa: E402
from synthetic import build_workload  # noqa: E402
from train import training  # noqa: E402
from utils.general import safe_state  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "north_star"
n_views = int(sys.argv[2]) if len(sys.argv) > 2 else 8
marks = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "0,500,1000,3000").split(",")]
rt.lib()
safe_state(True)
dev = torch.device("cuda")
bg = torch.zeros(3, device=dev)

PATHS = {
    "fused_graph": dict(fused_step=True, single_pass=True, use_graph=True, topology=False),
    "op_by_op": dict(fused_step=False, single_pass=True, use_graph=True, topology=False),
    "three_pass": dict(fused_step=False, single_pass=False, use_graph=False, topology=False),
    # with the operators: iteration 3000 is an opacity reset (opacity_reset_interval), so 2900 is recorded next to it
    "fused_graph_with_topology": dict(fused_step=True, single_pass=True, use_graph=True, topology=True,
                                      marks=[0, 500, 1000, 2000, 2900, 3000]),
    "op_by_op_with_topology": dict(fused_step=False, single_pass=True, use_graph=True, topology=True, marks=[0, 500, 1000]),
}


def psnr_all(model, cams):
    with torch.no_grad():
        v = []
        for c in cams:
            img = render(c, model, bg)["render"].clamp(0, 1)
            v.append(float(-10 * torch.log10(((img - c.original_image) ** 2).mean())))
    return sum(v) / len(v)


def endpoint_rmse_mm(model, gt_endpoints):
    if gt_endpoints is None or model._endpoints.shape != gt_endpoints.shape:
        return None
    return float(((model._endpoints.detach() - gt_endpoints) ** 2).sum(dim=1).mean().sqrt()) * 1e3


out = {"workload": wl, "views": n_views, "checkpoints": marks, "targets": "consistent (GT image, mask and orientation "
       "field from one copy of the strands with endpoints + N(0, 1 mm); confidence 1)",
       "start_state": "unperturbed geometry, SH DC + 0.3, opacity logit - 1.0", "paths": {}}
for name, cfg in PATHS.items():
    torch.manual_seed(0)
    model, cams, extent = build_workload(wl, device=dev, seed=0, n_views=n_views, consistent=True)
    from synthetic import perturbed_copy
    gt_ep = perturbed_copy(model, seed=0)[0]._endpoints.detach().clone() if hasattr(model, "_endpoints") else None
    with torch.no_grad():
        model._features_dc.add_(0.3)
        model._opacity.sub_(1.0)
    opt = OptimizationParams()
    opt.fused_step, opt.single_pass, opt.enable_topology = cfg["fused_step"], cfg["single_pass"], cfg["topology"]
    model.training_setup(opt)
    traj, t_train, done = [], 0.0, 0
    for m in cfg.get("marks", marks):
        if m > done:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ema = training(model, cams, opt, iterations=m - done, extent=extent, start_iteration=done,
                           use_graph=cfg["use_graph"])
            torch.cuda.synchronize()
            t_train += time.perf_counter() - t0
            done = m
        traj.append({"iteration": m, "psnr_db": psnr_all(model, cams), "endpoint_rmse_mm": endpoint_rmse_mm(model, gt_ep),
                     "segments": int(model.get_xyz.shape[0])})
        print(name, traj[-1], flush=True, file=sys.stderr)
    finite = all(bool(torch.isfinite(p).all()) for p in (model._endpoints, model._opacity, model._features_dc))
    out["paths"][name] = {"config": cfg, "trajectory": traj, "train_seconds": t_train, "its_per_sec": done / max(t_train, 1e-9),
                          "rollbacks": getattr(training, "last_rollbacks", None), "parameters_finite": finite}
    del model, cams
    torch.cuda.empty_cache()

base = out["paths"]["fused_graph"]["trajectory"]
out["checks"] = {
    "start_psnr_finite": all(abs(p["trajectory"][0]["psnr_db"]) < 1e3 for p in out["paths"].values()),
    "psnr_rises": all(p["trajectory"][-1]["psnr_db"] > p["trajectory"][0]["psnr_db"] + 1.0
                      for k, p in out["paths"].items() if not k.endswith("_with_topology")),
    "max_psnr_gap_db_between_structures": max(abs(a["psnr_db"] - b["psnr_db"]) for k in ("op_by_op", "three_pass")
                                              for a, b in zip(base, out["paths"][k]["trajectory"])),
}
print(json.dumps(out, indent=1))
