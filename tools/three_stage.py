#!/usr/bin/env python3
"""The reference's three-stage workflow (README.md:136-153: train.py -> merge.py -> train.py) on ONE synthetic capture at a
BASELINE.json size, timed and scored per stage (BASELINE config 4 "full 3-stage" in miniature):

  ground truth  S strands x 100 segments (synthetic.strand_polylines), rendered to image / mask / orientation targets for 16
                views at 800 x 800 (BASELINE config 2's frame) from the ground-truth strand model itself (consistent targets)
  Stage I       a Gaussian cloud of as many points as the ground truth has segments -- the segments' midpoints + N(0, 2 mm), what
                a sparse reconstruction hands train.py -- optimised with the reference's Stage-I loop: densify_and_clone /
                split / prune every 100 iterations from 500, opacity reset every 3000 (train.training, FusedCloudStep)
  Stage II      to_hair_gaussian_model + merge rounds until nothing is left to merge (merge.merge_rounds)
  Stage III     the strand model optimised WITH the topology operators (densification, merging, opacity reset) -- and, from the
                same Stage-II model (a deep copy), WITHOUT them, so that what the operators do to the
                image error on this capture can be read off: the two trajectories, and for the first events of the run with
                operators the PSNR right before and right after each densification (`events`)

PSNR (mean over all views against the targets), primitive count and wall time are recorded every 500 iterations; strand
metrics (loss/metrics.py compute_metrics against the ground-truth strands) at the end of stages II and III.
In the same run, `densify_inputs`: the ONE input of densification() that no reference run pins -- xyz_gradient_accum / denom as
accumulated from the RGB-only moments of the fused 7-channel pass -- is accumulated for 100 iterations of Stage III beside the
three-pass op-by-op form (the reference's structure: RGB render + its own screen-space gradient) evaluated on the SAME
parameters every iteration, and compared.

Round 6: `events` of every stage = the operators' own counts per event (clone / split / merge_collapsed / prune_* / merge, the
reference's TrainingInfo.densification_info; train.training(event_log=)), and for three Stage-I events `stage_I_event_checks`:
the op-by-op three-pass form accumulates its statistics beside the fused iteration over the 100 iterations in front of the event
and selects the same rows for clone and split.

  python tools/three_stage.py [strands=500] [iters_stage1=5000] [iters_stage3=5000] [views=16] [W=800] [H=800] [curly=0] [lean=0] > profiles/r05_three_stage.json
  (profiles/r05_three_stage_1080p.json: 2000 5000 5000 32 1920 1080 -- 200 k ground-truth segments, BASELINE config 3's frame and views)
  (profiles/r06_three_stage_c4.json: 10000 5000 5000 48 1920 1080 1 1 -- BASELINE config 4 as written: curly strands, 1 M ground-truth
   segments, 48 views at 1080p; lean=1 leaves out the Stage-III run without operators and the one-by-one events)
"""
import faulthandler
import json
import os
import sys
import time

faulthandler.enable()

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import hgs_runtime as rt  # noqa: E402
from arguments import OptimizationParams  # noqa: E402
from gaussian_renderer import render  # noqa: E402
from synthetic import build_capture, stage1_cloud  # noqa: E402
from train import training  # noqa: E402
from utils.general import safe_state  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 500
N1 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
N3 = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
N_SEG = 100
VIEWS = int(sys.argv[4]) if len(sys.argv) > 4 else 16
W = int(sys.argv[5]) if len(sys.argv) > 5 else 800
H = int(sys.argv[6]) if len(sys.argv) > 6 else 800
CURLY = bool(int(sys.argv[7])) if len(sys.argv) > 7 else False
LEAN = bool(int(sys.argv[8])) if len(sys.argv) > 8 else False
rt.lib()
_real_stdout, sys.stdout = sys.stdout, sys.stderr      # (the model classes print progress lines: stdout is the JSON alone)
safe_state(True)
dev = torch.device("cuda")
bg = torch.zeros(3, device=dev)
log = lambda *a: print(*a, file=sys.stderr, flush=True)


def psnr_all(model, cams):
    with torch.no_grad():
        v = [float(-10 * torch.log10(((render(c, model, bg)["render"].clamp(0, 1) - c.original_image) ** 2).mean())) for c in cams]
    return sum(v) / len(v)


def strand_metrics(model, gt_pts):
    """precision / recall / F1 of the model's oriented points against the ground-truth strands (loss/metrics.py)."""
    from loss.metrics import HairEvalData, compute_eval_data_from_hair_gs, compute_metrics
    mid = 0.5 * (gt_pts[:, 1:] + gt_pts[:, :-1]).reshape(-1, 3).astype(np.float64)
    d = (gt_pts[:, 1:] - gt_pts[:, :-1]).reshape(-1, 3).astype(np.float64)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    gt = HairEvalData(mid, d, np.repeat(np.arange(gt_pts.shape[0]), gt_pts.shape[1] - 1))
    res, labels = compute_metrics(compute_eval_data_from_hair_gs(model), gt, bidirectional=True)
    return {k: [float(x) for x in v] for k, v in res.items()}, [str(l) for l in labels]


def run_stage(model, cams, opt, extent, n_iters, name, events=None, pause_at=()):
    """`events`: list for the operators' counts (training(event_log=)); `pause_at`: iterations (multiples of 100, inside the
    densification schedule) whose densification inputs are checked against the three-pass form: the stage stops 100 iterations in
    front of each, cloud_event_check() runs those 100 itself."""
    traj, done, t_total, t_checks = [], 0, 0.0, 0.0
    checks = []
    traj.append({"iteration": 0, "psnr_db": psnr_all(model, cams), "primitives": int(model.get_xyz.shape[0]), "seconds": 0.0})
    log(name, traj[-1])
    while done < n_iters:
        nxt = (done // 500 + 1) * 500
        stop = min([nxt, n_iters] + [p - 100 for p in pause_at if p - 100 > done])
        n = stop - done
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        training(model, cams, opt, iterations=n, extent=extent, start_iteration=done, event_log=events)
        torch.cuda.synchronize()
        t_total += time.perf_counter() - t0
        done += n
        if done + 100 in pause_at:
            t0 = time.perf_counter()
            checks.append(cloud_event_check(model, cams, opt, extent, done, events))
            torch.cuda.synchronize()
            t_total += time.perf_counter() - t0       # (the check's 100 iterations are iterations of the stage, run eagerly twice over)
            t_checks += time.perf_counter() - t0
            done += 100
            log(name, "event check", checks[-1])
        if done % 500 == 0 or done == n_iters:
            traj.append({"iteration": done, "psnr_db": psnr_all(model, cams), "primitives": int(model.get_xyz.shape[0]),
                         "seconds": t_total})
            log(name, traj[-1])
    if pause_at:
        return traj, t_total, checks, t_checks
    return traj, t_total


def cloud_event_check(model, cams, opt, extent, done, events):
    """Iterations done + 1 .. done + 100 of Stage I, the last of which densifies: every iteration the op-by-op three-pass form
    (render() + loss_function + update_densification_stats: the reference's structure, train.py:146-171) is evaluated on the SAME
    parameters beside the fused iteration (whose Adam step alone is applied) and accumulates its own statistics; in front of the
    event both statistics select rows for clone and split exactly as densify_and_clone / densify_and_split do
    (scene/gaussian_model.py:298-322): the selections are compared row by row."""
    from hgs_runtime.strand_step import ViewTable, fused_step_for
    from loss.losses import loss_function
    from train import ViewSampler, training_step
    views = ViewTable(cams)
    fused = fused_step_for(model, views, opt, bg)
    stats = lambda: (model.max_radii2D, model.xyz_gradient_accum, model.denom)
    acc3 = [t.clone() for t in stats()]                # (both forms start from what the stage has accumulated since the last event)
    sampler = ViewSampler(cams, seed=1000 + done)
    out = None
    for it in range(done + 1, done + 101):
        cam = sampler.next()
        if it == done + 100:      # the event's inputs: both forms have seen iterations .. it - 1 (this one adds one more view to both)
            with torch.no_grad():
                thr, dense = float(opt.densify_grad_threshold), float(opt.percent_dense) * extent
                big = torch.max(model.get_scaling, dim=1).values > dense
                sel = {}
                for tag, (acc, den) in (("fused", (model.xyz_gradient_accum, model.denom)), ("three_pass", (acc3[1], acc3[2]))):
                    g = acc / den
                    g[g.isnan()] = 0.0
                    hot = torch.norm(g, dim=-1) >= thr
                    sel[tag] = (hot & ~big, hot & big, g.reshape(-1))
                near = (sel["three_pass"][2] - thr).abs() <= 1e-5 * thr
                out = {"event_iteration": done + 100, "gaussians": int(model.get_xyz.shape[0]),
                       "clone_rows": {k: int(v[0].sum()) for k, v in sel.items()}, "split_rows": {k: int(v[1].sum()) for k, v in sel.items()},
                       "rows_selected_differently": int(((sel["fused"][0] != sel["three_pass"][0]) | (sel["fused"][1] != sel["three_pass"][1])).sum()),
                       "rows_selected_differently_outside_1e-5_of_the_threshold":
                           int((((sel["fused"][0] != sel["three_pass"][0]) | (sel["fused"][1] != sel["three_pass"][1])) & ~near).sum()),
                       "statistics_identical": bool(all(torch.equal(a, b) for a, b in zip(stats(), acc3))),
                       "statistics_max_rel_diff": max(float((a - b).abs().max() / b.abs().max().clamp(min=1e-30)) for a, b in zip(stats(), acc3))}
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
        training_step(model, cam, opt, bg, it, extent=extent, fused=fused, event_log=events)
    return out


def densify_inputs_check(model, cams, opt, extent, iters=100):
    """xyz_gradient_accum / denom / max_radii2D over `iters` iterations: the fused 7-channel iteration (RGB-only moments of the
    single pass, statistics folded into the backward's lanes) against the three-pass op-by-op form -- render() + loss_function +
    update_densification_stats, the reference's structure -- evaluated on the same parameters each iteration (only the fused
    iteration's Adam step is applied)."""
    import copy
    from hgs_runtime.strand_step import ViewTable, fused_step_for
    from loss.losses import loss_function
    from train import ViewSampler, training_step
    o = copy.copy(opt)
    o.enable_topology = False
    views = ViewTable(cams)
    fused = fused_step_for(model, views, o, bg)
    keep = [t.clone() for t in (model.max_radii2D, model.xyz_gradient_accum, model.denom)]
    acc3 = [torch.zeros_like(t) for t in keep]
    for t in (model.max_radii2D, model.xyz_gradient_accum, model.denom):
        t.zero_()
    sampler = ViewSampler(cams, seed=7)
    for it in range(1, iters + 1):
        cam = sampler.next()
        # three passes on the CURRENT parameters, statistics into acc3, gradients discarded
        model.optimizer.zero_grad(set_to_none=True)
        pkg = render(cam, model, bg)
        loss, _ = loss_function(model, pkg["render"], cam, o)
        loss.backward()
        with torch.no_grad():
            fused_stats = [t.clone() for t in (model.max_radii2D, model.xyz_gradient_accum, model.denom)]
            for t, a in zip((model.max_radii2D, model.xyz_gradient_accum, model.denom), acc3):
                t.copy_(a)
            model.update_densification_stats(pkg["viewspace_points"], pkg["radii"], pkg["visibility_filter"])
            for t, a, f in zip((model.max_radii2D, model.xyz_gradient_accum, model.denom), acc3, fused_stats):
                a.copy_(t)
                t.copy_(f)
        model.optimizer.zero_grad(set_to_none=True)
        training_step(model, cam, o, bg, it, extent=extent, fused=fused)          # fused statistics + the Adam step
    torch.cuda.synchronize()
    fz = [t.clone() for t in (model.max_radii2D, model.xyz_gradient_accum, model.denom)]
    for t, k in zip((model.max_radii2D, model.xyz_gradient_accum, model.denom), keep):
        t.copy_(k)
    out = {"iterations": iters, "segments": int(fz[0].shape[0])}
    for name, a, b in zip(("max_radii2D", "xyz_gradient_accum", "denom"), fz, acc3):
        a, b = a.reshape(-1).double(), b.reshape(-1).double()
        out[name] = {"max_abs_diff": float((a - b).abs().max()), "max_rel_diff_of_scale": float((a - b).abs().max() / b.abs().max().clamp(min=1e-30)),
                     "identical": bool(torch.equal(a, b))}
    vis = acc3[2].reshape(-1) > 0
    ratio_f = (fz[1].reshape(-1)[vis] / fz[2].reshape(-1)[vis]).double()
    ratio_3 = (acc3[1].reshape(-1)[vis] / acc3[2].reshape(-1)[vis]).double()
    out["grads_ratio_max_rel_diff_of_scale"] = float((ratio_f - ratio_3).abs().max() / ratio_3.abs().max())
    thr = float(opt.densify_grad_threshold)
    out["selected_by_threshold"] = {"fused": int((ratio_f >= thr).sum()), "three_pass": int((ratio_3 >= thr).sum()),
                                    "differ": int(((ratio_f >= thr) != (ratio_3 >= thr)).sum())}
    return out


# ---- ground truth and targets
gt_pts, gt_model, cams, extent = build_capture((S, CURLY, VIEWS, W, H), device=dev, seed=0, n_seg=N_SEG)
out = {"ground_truth": {"strands": S, "segments": S * N_SEG, "views": VIEWS, "width": W, "height": H, "curly": CURLY,
                        "psnr_db_of_the_ground_truth_model": psnr_all(gt_model, cams)}}

from merge import merge_rounds  # noqa: E402


def stages_one_and_two(record):
    """Stage I + II."""
    safe_state(True)                      # (torch / numpy / random seeds: densify_and_split samples new centres)
    cloud = stage1_cloud(gt_pts, gt_model, extent, device=dev)
    opt1 = OptimizationParams()
    opt1.iterations = N1
    opt1._finalise()
    cloud.training_setup(opt1)
    if record:
        ev1 = []
        # events 6, 10 and 20 of the schedule (densification from iteration 500, every 100: iterations 1100, 1500, 2500)
        first = (int(opt1.densify_from_iter) // 100 + 1) * 100
        pause = tuple(p for p in (first + 500, first + 900, first + 1900) if p <= min(N1, int(opt1.densify_until_iter) - 1))
        traj1, t1, checks, tc = run_stage(cloud, cams, opt1, extent, N1, "stage I", events=ev1, pause_at=pause) if pause else \
            (run_stage(cloud, cams, opt1, extent, N1, "stage I", events=ev1) + ([], 0.0))
        # (an event check runs its 100 iterations eagerly, twice over, in the three-pass form: seconds of the tool, not of the loop)
        out["stage_I"] = {"iterations": N1, "seconds": t1, "its_per_sec": N1 / t1, "seconds_in_event_checks": tc,
                          "its_per_sec_outside_the_checks": (N1 - 100 * len(checks)) / max(t1 - tc, 1e-9), "trajectory": traj1,
                          "events": ev1, "stage_I_event_checks": checks}
    else:
        training(cloud, cams, opt1, iterations=N1, extent=extent, start_iteration=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hair = cloud.to_hair_gaussian_model()
    n_before = int(hair.strands_info.n_strands)
    rounds = merge_rounds(hair, 100, log=log if record else (lambda *a: None))
    torch.cuda.synchronize()
    t2 = time.perf_counter() - t0
    if record:
        m2, labels = strand_metrics(hair, gt_pts)
        out["stage_II"] = {"seconds": t2, "merge_rounds": rounds, "segments": int(hair.get_xyz.shape[0]), "strands_before": n_before,
                           "strands_after": int(hair.strands_info.n_strands), "psnr_db": psnr_all(hair, cams), "metrics": m2,
                           "metric_thresholds": labels}
        log("stage II", {k: v for k, v in out["stage_II"].items() if k != "metrics"})
    return hair


def stage_three_options(topology):
    opt3 = OptimizationParams()
    opt3.iterations = N3
    opt3._finalise()
    opt3.enable_topology = topology
    return opt3


# ---- Stage I, II, then III with the operators (every Stage-III variant below starts from a deep copy of the Stage-II model)
import copy  # noqa: E402
stage2_model = stages_one_and_two(True)
fresh_copy = lambda: copy.deepcopy(stage2_model)
hair = fresh_copy()
opt3 = stage_three_options(True)
hair.training_setup(opt3)
ev3 = []
traj3, t3 = run_stage(hair, cams, opt3, extent, N3, "stage III", events=ev3)
m3, _ = strand_metrics(hair, gt_pts)
out["stage_III"] = {"iterations": N3, "seconds": t3, "its_per_sec": N3 / t3, "trajectory": traj3, "metrics": m3, "events": ev3,
                    "strands": int(hair.strands_info.n_strands) if hair.strands_info is not None else None,
                    "rollbacks": getattr(training, "last_rollbacks", None)}
del hair
torch.cuda.empty_cache()

if LEAN:
    out["summary"] = {"psnr_db": {"stage_I_start": out["stage_I"]["trajectory"][0]["psnr_db"], "stage_I_end": out["stage_I"]["trajectory"][-1]["psnr_db"],
                                  "after_merge": out["stage_II"]["psnr_db"], "stage_III_start": traj3[0]["psnr_db"],
                                  "stage_III_best": max(p["psnr_db"] for p in traj3), "stage_III_end": traj3[-1]["psnr_db"]},
                      "wall_seconds": {"stage_I": out["stage_I"]["seconds"], "stage_II": out["stage_II"]["seconds"], "stage_III": t3}}
    sys.stdout = _real_stdout
    print(json.dumps(out, indent=1))
    sys.exit(0)

# ---- the same Stage-II model again: the densification inputs, Stage III WITHOUT the operators, and the first events one by one
hair = fresh_copy()
hair.training_setup(stage_three_options(True))
out["densify_inputs"] = densify_inputs_check(hair, cams, stage_three_options(True), extent)
log("densify inputs", out["densify_inputs"])
del hair
hair = fresh_copy()
opt3n = stage_three_options(False)
hair.training_setup(opt3n)
traj3n, t3n = run_stage(hair, cams, opt3n, extent, N3, "stage III without operators")
m3n, _ = strand_metrics(hair, gt_pts)
out["stage_III_without_operators"] = {"iterations": N3, "seconds": t3n, "its_per_sec": N3 / t3n, "trajectory": traj3n, "metrics": m3n}
del hair
hair = fresh_copy()
opt3 = stage_three_options(True)
hair.training_setup(opt3)
events, done = [], 0
first_event = (int(opt3.densify_from_iter) // 100 + 1) * 100
training(hair, cams, opt3, iterations=first_event - 1, extent=extent, start_iteration=0)
done = first_event - 1
for k in range(8):
    before = {"iteration": done, "psnr_db": psnr_all(hair, cams), "segments": int(hair.get_xyz.shape[0]),
              "opacity_sum": float(hair.get_opacity.sum())}
    training(hair, cams, opt3, iterations=1, extent=extent, start_iteration=done)          # the iteration with the operators
    done += 1
    after = {"iteration": done, "psnr_db": psnr_all(hair, cams), "segments": int(hair.get_xyz.shape[0]),
             "opacity_sum": float(hair.get_opacity.sum())}
    training(hair, cams, opt3, iterations=99, extent=extent, start_iteration=done)
    done += 99
    events.append({"before": before, "after": after, "psnr_change_by_the_event_db": after["psnr_db"] - before["psnr_db"],
                   "psnr_99_iterations_later_db": psnr_all(hair, cams)})
    log("event", events[-1])
out["events"] = events
out["summary"] = {"psnr_db": {"stage_I_start": out["stage_I"]["trajectory"][0]["psnr_db"], "stage_I_end": out["stage_I"]["trajectory"][-1]["psnr_db"],
                              "after_merge": out["stage_II"]["psnr_db"], "stage_III_start": traj3[0]["psnr_db"],
                              "stage_III_best": max(p["psnr_db"] for p in traj3), "stage_III_end": traj3[-1]["psnr_db"],
                              "stage_III_without_operators_end": traj3n[-1]["psnr_db"]},
                  "stage_III_ends_better_than_it_starts": traj3[-1]["psnr_db"] > traj3[0]["psnr_db"],
                  "stage_III_without_operators_ends_better_than_it_starts": traj3n[-1]["psnr_db"] > traj3n[0]["psnr_db"],
                  "mean_psnr_change_by_a_densification_event_db": sum(e["psnr_change_by_the_event_db"] for e in events) / len(events),
                  "wall_seconds": {"stage_I": out["stage_I"]["seconds"], "stage_II": out["stage_II"]["seconds"], "stage_III": t3}}
sys.stdout = _real_stdout
print(json.dumps(out, indent=1))
