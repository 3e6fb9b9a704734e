#!/usr/bin/env python3
"""bench.py -- headline benchmark: strand-Gaussian training iterations per second (+ forward render ms/view).

  python bench.py --gpus N --steps K --warmup W [--workload north_star|c2|c3|c4|c5|tiny] [--scaling weak|strong]

One *train iteration* is the reference's (train.py:133-204): lr update, ONE camera view, render + 2 more raster
passes inside the losses (mask, orientation), backward through all three, densification statistics, Adam step.
With N GPUs (one process per GPU, launched by torch.distributed.run, backend nccl = RCCL over xGMI) the views of an
optimizer step are shared between the ranks and the gradients all-reduced before Adam (hair-gs_amd/train.py):
  --scaling weak   (default) every rank trains ONE view per step: N views per optimizer step, per-GPU work fixed;
  --scaling strong SURVEY.md 8e's protocol: a FIXED global batch of --global-views V (default 8) views per optimizer
                   step, rank r renders views r, r+N, ... and sums their gradients inside its captured graph; one GPU
                   renders all V sequentially.  Total work is fixed, so the N-GPU speed-up is read off `value` directly.
`value` = whole-job train iterations (views) per second in both modes.

Default workload = BASELINE.json north_star: synthetic 100k strand-Gaussians (1000 strands x 100 segments), 1080p,
32 views.  Data is synthetic (SURVEY.md 8d generators), parameters random-init: there is no dataset offline.

Timing: W untimed warm-up steps, then the region of EXACTLY K steps (barrier + synchronize on both sides, maximum over
the ranks) is timed --repeats times back to back; `value` / `ms_per_step` are the MEDIAN region, `repeats` holds
min / median / max.  On one GPU the steps go out --steps-per-graph (default 8) at a time: each is a full optimizer step on
its own random view, eight of them are captured in one HIP graph and replayed with ONE launch (GraphedStep.step_many: a
graph launch costs ~8 us of idle GPU whatever it holds; bit-identical to single-step replays); steps that do not fill a
launch -- and every step with several ranks -- replay the single-step graph.  A `sustained` leg then replays steps for >= --sustained-seconds (default 2 s) in one region, long
enough for an SMI sampler to see the GPU busy, and reports its own rate.

The JSON line also carries
  psnr_vs_oracle_db  PSNR of the HIP render of view 0 against the CPU oracle's image of the same parameters (cpu_baseline leg)
  trained_state the same protocol after --trained-iters N (default 1000) iterations of the FULL loop (densification, merging,
                opacity reset, graph re-captures), beside the headline: the headline times the sparsest state of the workload
                (five steps after initialisation); `trained_state` carries value, segments, sum of tile-list lengths, per-kernel
                times and its own `roofline`.  (--trained-iters 0 where one state per kernel is wanted: the rocprofv3 scripts)
  roofline      blend_bwd_kernel (the dominant kernel), see roofline_block(): `frac` on the ALGORITHMIC bytes of SURVEY.md 8d
                (reference tile lists; 7-channel pass: (64 + 60) B x sum_tiles L_t + 36 B x W*H + 8 B x T per launch) over the
                mean launch duration -- measured in THIS run with HIP events on the launch stream around every launch of the
                kernel, in an EAGER continuation of the same steps right after the timed regions (events cannot bracket the
                nodes of a replayed graph); peak = 8 TB/s HBM3E -- and, beside it, `frac_moved` on the bytes the implementation
                physically moves (pixel planes of the tiles with a contributor only, culled lists).  `traffic` is not
                measurable from inside the process: it is the HBM byte count of the committed rocprofv3 --pmc passes of the
                same workload (`traffic_source` names the file), per launch, with FETCH_SIZE corrected per access shape.
  roofline_c3   the same block for BASELINE.json config 3 (200k strand-Gaussians), timed by a child process that runs before this
                one touches the GPU (--no-c3-leg skips it)
  pipeline_states  the states the three-stage workflow lives in (synthetic.PIPELINE_STATES, built the way tools/three_stage.py
                builds them): `stage1_1080p` -- the Stage-I cloud of a 200 k-segment capture at BASELINE config 3's frame after 1000
                iterations of the Stage-I loop -- and `stage3_merged` -- the Stage-II product of the same capture (5000 Stage-I
                iterations, merge rounds to the fixed point), the model Stage III starts from; each timed by a child process with
                the headline's protocol (--workload stage1_1080p / stage3_merged runs one directly): value, kernel_us_per_launch,
                roofline.  --no-pipeline-legs skips them
  cpu_baseline  the CPU oracle (oracle/, OpenMP C restatement of the reference rasterizer) doing the 3 raster fwd+bwd
                passes of one iteration on the host cores, plus -- `cpu_only_paths` -- the reference's CPU-only paths
                timed in the same run on the same cores: c_utils.filter_strand_list_segments (this package's native
                module; the reference's own Cython build is timed in the authoring container only, BASELINE.md
                B2: nothing compiled from the reference travels) and the strand metrics of eval.py
                (loss/metrics.py compute_metrics).  Rank 0, N=1 only, bounded samples.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "hair-gs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md:36)
HBM_ACHIEVABLE_GBS = 6300.0  # what a float4 copy reaches on this part (MI355X_MICROARCH.md:36,296: 6.29 TB/s, 79 %)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1,
                    help="ranks of the run, one per GPU.  N > 1 outside a torch.distributed.run environment: this process starts "
                         "`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>` as a child "
                         "BEFORE anything touches the GPU, relays its output and exits with its code (launch_ranks)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="print the launcher command of --gpus N as one JSON line and exit (nothing is started; CPU test)")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="north_star")
    ap.add_argument("--views", type=int, default=None, help="override the number of camera views")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--global-views", type=int, default=8, help="views per optimizer step of the strong-scaling protocol")
    ap.add_argument("--repeats", type=int, default=5, help="how many times the K-step region is timed")
    ap.add_argument("--steps-per-graph", type=int, default=8,
                    help="optimizer steps captured per graph launch (GraphedStep.step_many; 1 GPU, one view per step; the "
                         "steps that do not fill a launch replay the single-step graph)")
    ap.add_argument("--sustained-seconds", type=float, default=2.0, help="length of the sustained leg (0: skip)")
    ap.add_argument("--trained-iters", type=int, default=1000,
                    help="second leg (1 GPU): train this many iterations WITH the topology operators (densification, merging, "
                         "opacity reset), then time the same protocol on that state; reported beside the headline as "
                         "`trained_state` (0: skip -- what the rocprofv3 scripts pass, so that their kernel statistics average ONE "
                         "state per kernel)")
    ap.add_argument("--no-c3-leg", action="store_true",
                    help="skip `roofline_c3`: BASELINE.json config 3 (200k strand-Gaussians) timed by a child process of this run "
                         "(rank 0, 1 GPU, north_star workload only)")
    ap.add_argument("--no-pipeline-legs", action="store_true",
                    help="skip `pipeline_states`: the Stage-I cloud and the merged Stage-III start model of a 200 k-segment capture, "
                         "each timed by a child process of this run (rank 0, 1 GPU, north_star workload only)")
    ap.add_argument("--stage1-iters", type=int, default=None,
                    help="--workload stage1_1080p / stage3_merged: iterations of the Stage-I loop before the state is taken")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="disable per-kernel HIP-event timing")
    ap.add_argument("--three-pass", action="store_true",
                    help="three separate raster passes per iteration like the reference instead of the single 7-channel pass")
    ap.add_argument("--op-by-op", action="store_true", help="the iteration as the reference structures it (getters, render, loss_function as separate autograd ops) instead of the fused strand iteration")
    ap.add_argument("--head-tail-launch", action="store_true",
                    help="A/B: the loss head's last sums as a launch of their own instead of a spare workgroup of the backward's parameter launch")
    ap.add_argument("--prologue-launch", action="store_true",
                    help="A/B: the iteration prologue (view select + counter clearing) as a launch of its own instead of a rider of the parameter forward launch")
    ap.add_argument("--eager", action="store_true", help="eager dispatch of every kernel instead of replaying the captured HIP graph")
    ap.add_argument("--blocking", action="store_true",
                    help="reference-style forward (host reads num_rendered in every pass) instead of the async capacity mode")
    return ap.parse_args()


def cpu_baseline(model, cam, threads, bg):
    """Oracle timing of the raster work of ONE iteration (3 fwd+bwd passes) on the host; also the PSNR of the HIP
    render() of the same view and parameters against the oracle's RGB pass (the checker, not the product)."""
    import numpy as np
    import torch
    from oracle import hgs_oracle as O
    O.set_threads(threads)
    with torch.no_grad():
        base = dict(means3D=model.get_xyz.cpu().numpy(), opacities=model.get_opacity.cpu().numpy().reshape(-1),
                    scales=model.get_scaling.cpu().numpy(), rotations=model.get_rotation.cpu().numpy(),
                    cov3D_precomp=None, viewmatrix=cam.world_view_transform.cpu().numpy(),
                    projmatrix=cam.full_proj_transform.cpu().numpy(), campos=cam.camera_center.cpu().numpy(),
                    bg=np.zeros(3, np.float32), tanfovx=float(np.tan(cam.FoVx * 0.5)), tanfovy=float(np.tan(cam.FoVy * 0.5)),
                    W=cam.image_width, H=cam.image_height, sh_degree=model.active_sh_degree, scale_modifier=1.0)
        passes = [dict(base, shs=model.get_features.cpu().numpy(), colors_precomp=None),
                  dict(base, shs=None, colors_precomp=model.get_mask.repeat(1, 3).cpu().numpy()),
                  dict(base, shs=None, colors_precomp=model.get_orientation.cpu().numpy())]
    dpix = np.ones((3, cam.image_height, cam.image_width), np.float32)
    t0 = time.perf_counter()
    for s in passes:
        f = O.forward(s)
        O.backward(s, f, dpix)
    dt = time.perf_counter() - t0
    from gaussian_renderer import render
    with torch.no_grad():
        got = render(cam, model, bg)["render"].cpu().numpy()
    ref = O.forward(passes[0])["out_color"]
    mse = float(np.mean((got.astype(np.float64) - ref.astype(np.float64)) ** 2))
    psnr = None if mse == 0 else 10.0 * float(np.log10(1.0 / mse))      # (None: bit-identical images)
    return dt, {"value": psnr, "identical": mse == 0, "max_abs_diff": float(np.abs(got - ref).max()), "view": 0}


def cpu_only_paths():
    """The reference's CPU-only paths on bounded samples: the strand-segment filter of the smoothness
    loss (c_utils/c_utils.pyx:83-127) and the strand metrics of eval.py (loss/metrics.py:88-173)."""
    import numpy as np
    import c_utils
    from loss.metrics import HairEvalData, compute_metrics
    from synthetic import strand_polylines
    out = {}
    rng = np.random.default_rng(0)
    S = 2000
    strands = np.empty(S, object)
    for j in range(S):
        strands[j] = rng.integers(0, 10 ** 6, (101, 2)).astype(np.int64)

    def med(f, n=9):
        ts = []
        for _ in range(n):
            t = time.perf_counter()
            f()
            ts.append(time.perf_counter() - t)
        return sorted(ts)[n // 2]
    out["filter_strand_list_segments_ms"] = med(lambda: c_utils.filter_strand_list_segments(strands)) * 1e3
    out["filter_strand_list_segments_sample"] = f"{S} strands x 100 segments, 1 core (native CPython extension)"
    n_str = 500
    sp = strand_polylines(n_str, 100, seed=0)
    mid = 0.5 * (sp[:, 1:] + sp[:, :-1]).reshape(-1, 3)
    d = (sp[:, 1:] - sp[:, :-1]).reshape(-1, 3)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    sid = np.repeat(np.arange(n_str), 100)
    gt = HairEvalData(mid.astype(np.float64), d.astype(np.float64), sid)
    pred = HairEvalData(mid + rng.normal(size=mid.shape) * 1e-3, d.astype(np.float64), sid)
    t = time.perf_counter()
    res, _ = compute_metrics(pred, gt, bidirectional=True)
    out["compute_metrics_s"] = time.perf_counter() - t
    out["compute_metrics_sample"] = f"{mid.shape[0]} predicted vs {mid.shape[0]} ground-truth oriented points, 4 threshold pairs"
    out["compute_metrics_f1"] = [float(x) for x in res["f1(b)"]]
    return out

def roofline_block(kern, fs, W, H, ch, wl_tag):
    """`roofline` of the dominant kernel (blend_bwd_kernel) from this run's per-launch HIP-event time `kern` and the list
    statistics `fs` (frame_stats) of the state that was timed.
      algorithmic bytes  SURVEY.md 8d, generalised to the C-channel pass, counted on the REFERENCE's tile lists:
                         (16 C + 16 + 4 NPART) sum_t L_t + (4 C + 8) W H + 8 T      (C = 7: 124 sum L_t + 36 W H + 8 T)
      moved bytes model  what this implementation physically moves: per-pixel planes only on tiles where a pixel blended an entry
                         (the backward returns before touching pixel data elsewhere), records and rows of the CULLED lists:
                         (4 C + 8) 256 #tiles(maxc > 0) + (16 C + 16 + 4 NPART) sum_t L_t(culled) + 8 T
      traffic            HBM bytes of the committed rocprofv3 --pmc passes of the same workload, per launch.  FETCH_SIZE counts a
                         64-byte request at 64 bytes and a 128-byte request at 64 bytes too (profiles/r04_fetch_shape_probe.txt,
                         tools/probes/fetch_shape_probe.hip: ratio 1.000 for the blend's dword-per-lane quadrant reads of planar
                         images and for 64-byte gathers, 0.500 for 256 contiguous bytes per wave instruction and for 16 B/lane
                         streams): the per-pixel planes are counted in full, the coalesced record batches at half, so
                         traffic = FETCH_SIZE + 1/2 record bytes (culled lists) + WRITE_SIZE  (rounds 1-3 doubled all of FETCH_SIZE)."""
    bwd_ms, bwd_n = kern["blend_bwd_kernel"]
    fwd_ms, fwd_n = kern["blend_fwd_kernel"]
    T = ((W + 15) // 16) * ((H + 15) // 16)
    rec_b, part_b = (64.0, 60.0) if ch == 7 else (48.0, 36.0)
    pix_b = 4.0 * ch + 8.0
    bytes_bwd = (rec_b + part_b) * fs["meanL"] + pix_b * W * H + 8.0 * T
    bytes_fwd = rec_b * fs["meanL"] + pix_b * W * H + 8.0 * T
    moved_bwd = (rec_b + part_b) * fs["meanL_culled"] + pix_b * 256.0 * fs["tiles_used"] + 8.0 * T
    dur_s = bwd_ms / max(bwd_n, 1) * 1e-3
    ach = bytes_bwd / dur_s / 1e9 if bwd_ms > 0 else 0.0
    moved = moved_bwd / dur_s / 1e9 if bwd_ms > 0 else 0.0
    traffic = traffic_source = valu_util = lanes_busy = None
    for name in (f"r06_pmc_raster_{wl_tag}.json", f"r05_pmc_raster_{wl_tag}.json", f"r04_pmc_raster_{wl_tag}.json", f"r03_pmc_raster_{wl_tag}.json", f"r02_pmc_raster_{wl_tag}.json", f"pmc_raster_{wl_tag}.json"):
        pmc_path = os.path.join(ROOT, "profiles", name)
        if ch == 7 and os.path.exists(pmc_path):
            pm = json.load(open(pmc_path)).get("blend_bwd_kernel<7>", {})
            if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
                traffic = (pm["FETCH_SIZE"] + pm["WRITE_SIZE"]) * 1024.0 + 0.5 * rec_b * fs["meanL_culled"]
                traffic_source = (f"profiles/{name}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of the same workload (profiled, static), "
                                  "FETCH_SIZE corrected per access shape (profiles/r04_fetch_shape_probe.txt)")
                if pm.get("GRBM_GUI_ACTIVE") and pm.get("SQ_INSTS_VALU"):
                    # vector-pipe utilisation: every VALU instruction occupies its SIMD for 4 cycles; 1024 SIMDs;
                    # GRBM_GUI_ACTIVE counts the kernel's cycles on each of the 8 XCDs
                    valu_util = pm["SQ_INSTS_VALU"] * 4.0 / (pm["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
                if pm.get("SQ_THREAD_CYCLES_VALU") and pm.get("SQ_ACTIVE_INST_VALU"):
                    lanes_busy = pm["SQ_THREAD_CYCLES_VALU"] / (pm["SQ_ACTIVE_INST_VALU"] * 64.0)
                break
    # `bound`: the roofline the kernel is priced against -- HBM, as SURVEY.md 8d prescribes for this byte-moving path (`frac` is
    # the fraction of the HBM specification).  `limited_by`: what the counters of the committed --pmc passes say actually holds
    # it -- vector-instruction ISSUE (pipe utilisation 0.7 with every lane enabled while ~10 of 64 lanes blend a kept entry),
    # not HBM; `valu_util` is the fraction of that limit in use.  null where no --pmc pass of THIS state is committed.
    roof = {"kernel": "blend_bwd_kernel", "bound": "hbm", "limited_by": None if valu_util is None else ("valu" if valu_util >= 0.5 else "hbm"),
            "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            # the fraction of the resource that actually binds this kernel (vector-instruction issue: `limited_by`), beside `frac`
            "valu_frac": valu_util,
            "peak_achievable": HBM_ACHIEVABLE_GBS, "frac_of_achievable": ach / HBM_ACHIEVABLE_GBS,
            "moved_bytes_model": moved_bwd, "moved_gbs": moved, "frac_moved": moved / HBM_PEAK_GBS,
            "tiles_read": fs["tiles_used"], "tiles": T,
            "valu_util": valu_util, "valu_exec_lane_util": lanes_busy, "traffic": traffic, "traffic_source": traffic_source,
            "duration_source": "HIP events around every launch, eager continuation of this run's steps",
            "algorithmic_bytes_per_launch": bytes_bwd, "mean_launch_us": bwd_ms / max(bwd_n, 1) * 1e3, "launches": bwd_n,
            "sum_tile_list_len": fs["meanL"], "sum_tile_list_len_after_tile_cull": fs["meanL_culled"]}
    roof_fwd = {"achieved": bytes_fwd / (fwd_ms / max(fwd_n, 1) * 1e-3) / 1e9 if fwd_ms else 0.0,
                "unit": "GB/s", "mean_launch_us": fwd_ms / max(fwd_n, 1) * 1e3}
    return roof, roof_fwd


def c3_roofline_leg():
    """BASELINE.json config 3 (200k strand-Gaussians, 32 views @ 1080p: "rocprof HBM GB/s on blend kernel"): the blend kernels'
    roofline on that workload, timed by a CHILD process (same library, 4 of the views, eager per-kernel HIP events) that runs
    to completion BEFORE this process touches the GPU (a process that has initialised the GPU must not start another program
    on this pool)."""
    import subprocess
    try:
        cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", "c3", "--views", "4", "--steps", "40",
                             "--warmup", "5", "--repeats", "1", "--sustained-seconds", "0", "--trained-iters", "0",
                             "--no-cpu-baseline", "--no-c3-leg"], capture_output=True, text=True, timeout=300)
        c3 = json.loads(cp.stdout.strip().splitlines()[-1])
        return dict(c3["roofline"], workload=c3["config"]["workload"], iters_per_sec=c3["value"],
                    blend_fwd=c3.get("roofline_blend_fwd"), kernel_us_per_launch=c3.get("kernel_us_per_launch"))
    except Exception as e:
        return {"error": str(e)}


def pipeline_state_leg(name):
    """One entry of `pipeline_states`: this script on --workload <name> as a CHILD process that runs to completion before this
    process touches the GPU (as c3_roofline_leg)."""
    import subprocess
    try:
        cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", name, "--steps", "40", "--warmup", "5",
                             "--repeats", "3", "--sustained-seconds", "0", "--trained-iters", "0", "--no-cpu-baseline",
                             "--no-c3-leg", "--no-pipeline-legs"], capture_output=True, text=True, timeout=600)
        r = json.loads(cp.stdout.strip().splitlines()[-1])
        return {"value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"], "workload": r["config"]["workload"],
                "state": r["config"].get("pipeline_state"), "gaussians": r["config"]["gaussians"],
                "mean_num_rendered_after_tile_cull": r["config"]["mean_num_rendered_after_tile_cull"],
                "mean_sum_tile_list_len": r["config"]["mean_sum_tile_list_len"], "render_ms_per_view": r["render_ms_per_view"],
                "kernel_us_per_launch": r.get("kernel_us_per_launch"), "roofline": r.get("roofline"),
                "roofline_blend_fwd": r.get("roofline_blend_fwd")}
    except Exception as e:
        return {"error": str(e)}


def launcher_command(args, argv, port=None):
    """The command `bench.py --gpus N` (N > 1, no torch.distributed.run environment) starts: one rank per GPU of THIS node,
    rendezvous on 127.0.0.1 (the container's hostname may not resolve), the caller's own arguments handed on unchanged."""
    if port is None:
        import socket
        with socket.socket() as s_:            # a free port, so that two runs on one box do not meet
            s_.bind(("127.0.0.1", 0))
            port = s_.getsockname()[1]
    rest = [a for a in argv if a != "--dry-launch"]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + rest


def visible_gpus():
    """GPUs this process would see, counted WITHOUT a HIP call (the launcher must not initialise the runtime: it only starts the
    ranks): the KFD topology's nodes with SIMDs, cut down by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES.  None: unknown (no
    topology to read although the driver is there) -- the ranks themselves refuse a world larger than their device count."""
    import glob
    n = 0
    if not os.path.isdir("/sys/class/kfd"):
        return 0                                         # no KFD driver: no GPU
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    for f in nodes:
        try:
            props = dict(line.split()[:2] for line in open(f) if len(line.split()) >= 2)
            n += int(props.get("simd_count", "0")) > 0
        except OSError:
            return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(args, argv):
    """`python bench.py --gpus N` by itself: N RCCL ranks as a CHILD process (never exec; this process makes no HIP call at all:
    the devices are counted from the KFD topology, visible_gpus), the child's stdout (rank
    0's JSON line) and stderr inherited, its return code ours.  Fewer visible devices than ranks is an ERROR, not a smaller
    run: a 1-rank line must never be recorded as an N-GPU point (HGS_BENCH_SHARE_GPU=1, the logic check that puts every rank
    on cuda:0 over gloo, is the one exception and says so in its line)."""
    import subprocess
    cmd = launcher_command(args, argv)
    if args.dry_launch:
        print(json.dumps({"launcher": cmd, "gpus": args.gpus}))
        return 0
    have = visible_gpus()
    if have is not None and have < args.gpus and os.environ.get("HGS_BENCH_SHARE_GPU") != "1":
        print(f"bench.py: --gpus {args.gpus} needs {args.gpus} visible GPUs, this node shows {have} "
              "(no line printed: a smaller run is not an N-GPU measurement)", file=sys.stderr)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.dry_launch):
        sys.exit(launch_ranks(args, sys.argv[1:]))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        # (a torchrun world that is not the one asked for: refuse, or `n_gpus` of the line would contradict the command)
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE')}", file=sys.stderr)
        sys.exit(2)
    c3_leg = pipeline_legs = None
    side_legs_ok = (int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.workload == "north_star"
                    and not args.no_kernel_timing and not args.eager and not args.blocking
                    and "rocprof" not in os.environ.get("LD_PRELOAD", "").lower())   # (a profiler's preload has initialised the GPU)
    if side_legs_ok and not args.no_c3_leg:
        c3_leg = c3_roofline_leg()
    if side_legs_ok and not args.no_pipeline_legs:
        pipeline_legs = {name: pipeline_state_leg(name) for name in ("stage1_1080p", "stage3_merged")}
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # HGS_BENCH_SHARE_GPU=1: every rank on cuda:0 with the gloo backend -- a LOGIC check of the multi-rank path on a
    # one-GPU box (the numbers it prints are meaningless); the real path is one rank per GPU over RCCL
    share = os.environ.get("HGS_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if not share and torch.cuda.device_count() < world:
            print(f"bench.py: {world} ranks but {torch.cuda.device_count()} visible GPUs", file=sys.stderr)
            sys.exit(2)
        torch.cuda.set_device(local_rank)
        if share:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        if dist.get_world_size() != args.gpus or (not share and dist.get_backend() != "nccl"):
            print(f"bench.py: process group has {dist.get_world_size()} ranks over {dist.get_backend()}, "
                  f"--gpus {args.gpus} over nccl (RCCL) was asked for", file=sys.stderr)
            sys.exit(2)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import hgs_runtime as rt
    from arguments import OptimizationParams
    from gaussian_renderer import render
    from synthetic import PIPELINE_STATES, WORKLOADS, build_pipeline_state, build_workload
    from train import ViewParallel, ViewSampler, training_step
    from utils.general import safe_state

    rt.lib()  # fail loudly if the HIP library is missing
    if os.environ.get("HGS_SEG_POLICY"):   # measurement aid: "min,max,target" of hgs_set_segment_policy
        pol = [int(x) for x in os.environ["HGS_SEG_POLICY"].split(",")]
        rt.check(rt.lib().hgs_set_segment_policy(*pol[:3]))
    from diff_gaussian_rasterization import _C as raster
    raster.set_async(not args.blocking)
    safe_state(True)
    pipeline_info = None
    if args.workload in PIPELINE_STATES:
        # a state of the three-stage workflow (its own process: the Stage-I loop that leads to it runs here, untimed)
        if world != 1:
            raise SystemExit("the pipeline-state workloads are single-GPU legs")
        args.trained_iters = 0
        model, cams, extent, pipeline_info = build_pipeline_state(args.workload, device=dev, seed=0, stage1_iters=args.stage1_iters,
                                                                   n_views=args.views)
        raster.set_async(not args.blocking)
    else:
        model, cams, extent = build_workload(args.workload, device=dev, seed=0, n_views=args.views)
    opt = OptimizationParams()
    opt.single_pass = not args.three_pass
    opt.fused_step = not args.op_by_op
    opt.defer_head_tail = not args.head_tail_launch
    opt.ride_prologue = not args.prologue_launch
    if os.environ.get("HGS_SKIP_UNREAD") == "0":   # A/B: the SSIM backward filters every block the 3x3 zero rule does not spare
        opt.skip_unread_blocks = False
    opt.enable_topology = False  # densify/merge intervals (every 100 it) are reported separately, not in the timed loop
    model.training_setup(opt)
    bg = torch.zeros(3, dtype=torch.float32, device=dev)
    vp = ViewParallel()
    sampler = ViewSampler(cams, seed=0, rank=vp.rank, world=vp.world)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from train import GraphedStep, fused_step_applicable
    use_graph = not (args.eager or args.blocking)
    strong = args.scaling == "strong"
    views_per_rank = 1
    if strong:
        if not use_graph:
            raise SystemExit("--scaling strong needs the captured step (no --eager / --blocking)")
        if args.global_views % world:
            raise SystemExit(f"--global-views {args.global_views} does not divide over {world} ranks")
        views_per_rank = args.global_views // world
    views_per_step = views_per_rank * world if strong else world      # views of ONE optimizer step, whole job
    def measure(it_start, do_sustained, do_kernels):
        """The measurement protocol on the model's CURRENT state: capture, W warm-up steps, `repeats` regions of EXACTLY K
        steps each restarted from the post-warm-up state, the sustained leg, per-kernel HIP-event timing.  Leaves the
        model in the post-warm-up state."""
        it = it_start
        views = fused = None
        if fused_step_applicable(model, opt):
            from hgs_runtime.strand_step import ViewTable, fused_step_for
            views = ViewTable(cams)
            fused = fused_step_for(model, views, opt, bg)   # eager launches of the same fused iteration (timing pass, --eager)
            fused.defer_tail = opt.defer_head_tail           # (training_step always runs forward and backward together)
        if use_graph:
            # the whole iteration is captured once into a HIP graph and replayed (train.GraphedStep); the W warm-up steps
            # and the K timed steps are real optimizer steps on successive random views, exactly like the eager loop
            spg = args.steps_per_graph if (fused is not None and views_per_rank == 1 and (world == 1 or vp.graph_collective_ok())) else 1
            gs = GraphedStep(model, cams, opt, bg, extent=extent, vp=vp, views=views, views_per_step=views_per_rank,
                             steps_per_graph=max(1, spg))
            gs.capture(cams, iteration=it_start + 1)

            def one_step():
                nonlocal it
                it += 1
                gs.step(sampler.next_batch(args.global_views) if views_per_rank > 1 else sampler.next(), it)
        else:
            def one_step():
                nonlocal it
                it += 1
                training_step(model, sampler.next(), opt, bg, it, extent=extent, vp=vp, fused=fused)

        def run_steps(n_steps):
            """EXACTLY n_steps optimizer steps: whole launches of the several-steps graph, the rest one step per launch."""
            nonlocal it
            K = gs.steps_per_graph if use_graph else 1
            while K > 1 and n_steps >= K:
                gs.step_many([sampler.next() for _ in range(K)], it + 1)
                it += K
                n_steps -= K
            for _ in range(n_steps):
                one_step()

        def timed_region(n_steps):
            """EXACTLY n_steps optimizer steps between two barrier + synchronize pairs; seconds, maximum over the ranks."""
            sync_all()
            t0 = time.perf_counter()
            run_steps(n_steps)
            sync_all()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        run_steps(args.warmup)
        # Training changes the workload (the Gaussians grow: +30 % instances per view after 500 steps).  Every timed region,
        # the kernel-timing pass and the workload statistics therefore start from the SAME state -- the parameters, Adam
        # moments and step counters as they are after the warm-up -- restored outside the timed regions; the regions time
        # real optimizer steps of that trajectory.
        torch.cuda.synchronize()
        state_tensors = [p.data for g_ in model.optimizer.param_groups for p in g_["params"]]
        for g_ in model.optimizer.param_groups:
            for p in g_["params"]:
                state_tensors += [v for v in model.optimizer.state.get(p, {}).values() if torch.is_tensor(v)]
        state_tensors += [model.max_radii2D, model.xyz_gradient_accum, model.denom]
        snapshot = [t.clone() for t in state_tensors]
        it0 = it

        def restore():
            nonlocal it
            with torch.no_grad():
                for t, s_ in zip(state_tensors, snapshot):
                    t.copy_(s_)
            it = it0
            sampler.rng.seed(12345)
            sampler.stack = []
            torch.cuda.synchronize()

        regions = []
        for _ in range(max(1, args.repeats)):
            restore()
            regions.append(timed_region(args.steps))
        regions.sort()
        dt = regions[len(regions) // 2]
        sustained = None
        if do_sustained:
            # >= sustained_seconds of back-to-back steps in ONE region (for SMI samplers); the state is put back every K steps
            # (12 small device copies per K steps, inside the region) so that the workload stays the one the headline times
            chunks = max(1, int(args.sustained_seconds / dt) + 1)
            if world > 1:   # (every rank must run the same number of steps)
                n = torch.tensor([chunks], dtype=torch.int64, device=dev)
                dist.all_reduce(n, op=dist.ReduceOp.MAX)
                chunks = int(n.item())
            restore()
            sync_all()
            t0 = time.perf_counter()
            for _ in range(chunks):
                with torch.no_grad():
                    for t, s_ in zip(state_tensors, snapshot):
                        t.copy_(s_, non_blocking=True)
                it = it0
                run_steps(args.steps)
            sync_all()
            t_sus = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(t_sus, op=dist.ReduceOp.MAX)
            t_sus = float(t_sus.item())
            n_sus = chunks * args.steps
            sustained = {"steps": n_sus, "seconds": t_sus, "iters_per_sec": views_per_step * n_sus / t_sus,
                         "state_restored_every_steps": args.steps}
        restore()
        if use_graph:
            gs.check()  # instance counts of the captured passes stayed within capacity
        # per-kernel device time: HIP events cannot bracket individual nodes of a replayed graph, so the same steps are
        # CONTINUED with eager dispatch after the timed regions and every library launch is bracketed by events on its
        # stream (hgs_prof_*): the figures are per launch, one view per eager step
        kern = {}
        if do_kernels:
            sync_all()
            rt.prof_collect()
            rt.prof_enable(True)
            kern_steps = min(args.steps, 50)
            # (the same form of the step as the timed graph holds: Adam in the backward's lanes where that is what was captured)
            inline = fused is not None and use_graph and gs.inline_adam and fused.enable_inline_adam(True)
            for _ in range(kern_steps):
                it += 1
                training_step(model, sampler.next(), opt, bg, it, extent=extent, vp=vp, fused=fused)
            if inline:
                fused.enable_inline_adam(False)
            sync_all()
            kern = rt.prof_collect()
            rt.prof_enable(False)

        return dict(dt=dt, regions=regions, sustained=sustained, kern=kern, kern_steps=kern_steps if kern else 0,
                    steps_per_graph=(gs.steps_per_graph if use_graph else None), fused=fused,
                    inline_adam=bool(use_graph and gs.inline_adam),
                    collective_captured=(gs.collective_captured if use_graph else False))

    head = measure(0, args.sustained_seconds > 0, not args.no_kernel_timing)
    dt, regions, sustained, kern, kern_steps, fused = (head[k] for k in ("dt", "regions", "sustained", "kern", "kern_steps", "fused"))

    def frame_stats():
        """Forward-only render ms/view and the per-view list statistics of the model's current state."""
        # ---- forward-only render ms/view, all views, after 3 warm-ups
        raster.check_async()
        with torch.no_grad():
            for c in cams[:3]:
                render(c, model, bg)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            sumL = 0
            for c in cams:
                render(c, model, bg)
            torch.cuda.synchronize()
            render_ms = (time.perf_counter() - t1) * 1e3 / len(cams)
            raster.check_async()
            raster.set_async(False)
            # the same frames through gaussian_renderer.frames.FrameRenderer: the whole forward of a view as one captured
            # graph, re-pointed per view (same protocol: 3 warm-ups, all views, one validation at the end)
            from gaussian_renderer.frames import FrameRenderer
            KF = 8 if len(cams) % 8 == 0 else 1
            fr = FrameRenderer(model, cams, bg, frames_per_launch=KF)

            def all_views():
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                if KF > 1:
                    for i in range(0, len(cams), KF):
                        fr.render_batch(list(range(i, i + KF)), check=False)
                else:
                    for i in range(len(cams)):
                        fr.render(i, check=False)
                redo = fr.validate()
                return (time.perf_counter() - t1) * 1e3 / len(cams), redo

            for i in range(3):
                fr.render(i % len(cams))
            render_graph_ms, redo = all_views()
            if redo:            # a view exceeded the captured capacity: measure again on the recaptured graph
                fr.render(redo[0])
                render_graph_ms, redo = all_views()
                assert redo == []
            with torch.no_grad():
                same = bool(torch.equal(fr.render(1 % len(cams))["render"], render(cams[1 % len(cams)], model, bg)["render"]))
            del fr
            # sum over tiles of L_t (= tile_maxc) per view, for the algorithmic-byte model
            from diff_gaussian_rasterization import _C as C_
            W, H = cams[0].image_width, cams[0].image_height
            T = ((W + 15) // 16) * ((H + 15) // 16)
            lay = rt.layout("image", W, H)
            # Two passes per view: with the reference's tile lists (culling off: L_t and num_rendered as SURVEY.md 8d defines
            # them, the unit the algorithmic bytes are counted in) and with the lists the product runs on (culling on).
            import math
            stats = {}
            for cull in (False, True):
                was = C_.set_tile_cull(cull)
                Ls, Rs, Us = [], [], []
                for c in cams:
                    out = C_.rasterize_gaussians(bg, model.get_xyz, torch.empty(0, device=dev), model.get_opacity,
                                                 model.get_scaling, model.get_rotation, 1.0, torch.empty(0, device=dev),
                                                 c.world_view_transform, c.full_proj_transform, math.tan(c.FoVx * 0.5),
                                                 math.tan(c.FoVy * 0.5), H, W, model.get_features, model.active_sh_degree,
                                                 c.camera_center, False, False)
                    img = out[5]
                    maxc = img[lay["tile_maxc"]:lay["tile_maxc"] + 4 * T].view(torch.int32)
                    Ls.append(int(maxc.sum().item()))
                    Us.append(int((maxc > 0).sum().item()))
                    Rs.append(out[0])
                C_.set_tile_cull(was)
                stats[cull] = (sum(Ls) / len(Ls), sum(Rs) / len(Rs), sum(Us) / len(Us))
            (meanL, meanR, _), (meanL_culled, meanR_culled, tiles_used) = stats[False], stats[True]

        return dict(render_ms=render_ms, render_graph_ms=render_graph_ms, render_graph_equal=same, render_graph_frames=KF, meanL=meanL, meanR=meanR, meanL_culled=meanL_culled, meanR_culled=meanR_culled, tiles_used=tiles_used)

    fs = frame_stats()
    render_ms, meanL, meanR, meanL_culled, meanR_culled = (fs[k] for k in ("render_ms", "meanL", "meanR", "meanL_culled", "meanR_culled"))
    P_head = model.get_xyz.shape[0]

    cpu_result = {}
    if world == 1 and not args.no_cpu_baseline:
        threads = min(os.cpu_count() or 1, 64)
        try:
            sec, psnr = cpu_baseline(model, cams[0], threads, bg)
            # render PSNR of the HIP path against the oracle's image of the same parameters and view (SURVEY.md 8d 'PSNR')
            cpu_result["psnr_vs_oracle_db"] = psnr
            cpu_result["cpu_baseline"] = {"value": 1.0 / sec, "unit": "iters/s", "cores": threads, "kind": "port",
                                      "sample": "1 view of the same workload: the 3 raster fwd+bwd passes of one "
                                                "iteration through oracle/ (OpenMP C restatement of the reference "
                                                "rasterizer); losses and Adam excluded"}
            try:
                cpu_result["cpu_baseline"]["cpu_only_paths"] = dict(cpu_only_paths(), host_cores=os.cpu_count())
            except Exception as e:
                cpu_result["cpu_baseline"]["cpu_only_paths"] = {"error": str(e)}
        except Exception as e:  # the baseline must never break the headline number
            cpu_result["cpu_baseline"] = {"value": None, "error": str(e)}

    # ---- second leg: the same protocol on a TRAINED state (the headline times the easiest state of the workload: five steps
    # after initialisation).  N iterations of the full loop -- densification, merging, opacity reset at the reference's
    # intervals, graph re-captures -- then capture / warm-up / regions exactly as above.  Reported beside the headline.
    trained = None
    if args.trained_iters > 0 and world == 1 and use_graph and not strong:
        from train import training
        raster.set_async(not args.blocking)
        opt.enable_topology = True
        t_train = time.perf_counter()
        training(model, cams, opt, iterations=args.trained_iters, extent=extent, start_iteration=args.warmup, seed=1,
                 steps_per_graph=max(1, args.steps_per_graph))
        torch.cuda.synchronize()
        t_train = time.perf_counter() - t_train
        opt.enable_topology = False
        raster.set_async(not args.blocking)
        tr = measure(args.warmup + args.trained_iters, False, not args.no_kernel_timing)
        tfs = frame_stats()
        trained = {"iterations_trained": args.trained_iters, "training_seconds_incl_topology_and_recaptures": t_train,
                   # (the WHOLE loop over those iterations -- graph replays, densification / merging / opacity resets, re-captures;
                   # tools/soak.py runs 3000 of them: profiles/rNN_soak.txt)
                   "whole_loop_iters_per_sec_incl_topology_and_recaptures": args.trained_iters / max(t_train, 1e-9),
                   "gaussians": int(model.get_xyz.shape[0]),
                   "value": args.steps / tr["dt"], "unit": "iters/s", "ms_per_step": tr["dt"] * 1e3 / args.steps,
                   "repeats": {"n": len(tr["regions"]), "min_iters_per_sec": args.steps / tr["regions"][-1],
                               "max_iters_per_sec": args.steps / tr["regions"][0]},
                   "render_ms_per_view": tfs["render_ms"], "render_ms_per_view_frame_renderer": tfs["render_graph_ms"],
                   "mean_num_rendered": tfs["meanR"],
                   "mean_num_rendered_after_tile_cull": tfs["meanR_culled"], "mean_sum_tile_list_len": tfs["meanL"],
                   "capacity_rollbacks_while_training": getattr(training, "last_rollbacks", None)}
        if tr["kern"]:
            trained["kernel_us_per_launch"] = {k: (v[0] / v[1] * 1e3 if v[1] else 0.0) for k, v in tr["kern"].items()}
            trained["roofline"] = roofline_block(tr["kern"], tfs, cams[0].image_width, cams[0].image_height,
                                                 7 if getattr(opt, "single_pass", True) else 3, args.workload + "_trained")[0]

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    W, H = cams[0].image_width, cams[0].image_height
    T = ((W + 15) // 16) * ((H + 15) // 16)
    kind_name = "strand-Gaussians" if hasattr(model, "endpoint_pairs") else "Gaussians (Stage-I cloud)"
    P = P_head
    result = {
        "metric": "train_iters_per_sec", "value": views_per_step * args.steps / dt, "unit": "iters/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "repeats": {"n": len(regions), "min_iters_per_sec": views_per_step * args.steps / regions[-1],
                    "median_iters_per_sec": views_per_step * args.steps / dt,
                    "max_iters_per_sec": views_per_step * args.steps / regions[0]},
        "sustained": sustained,
        "sustained_iters_per_sec": None if sustained is None else sustained["iters_per_sec"],
        "n_ranks_seen": dist.get_world_size() if world > 1 else 1,
        "ranks_share_one_gpu": share if world > 1 else None,    # True: HGS_BENCH_SHARE_GPU=1, a logic check -- not a measurement
        "collective_backend": (dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else "")) if world > 1 else None,
        # True: the gradient all-reduce and Adam are nodes of the step's HIP graph (several optimizer steps per launch work
        # across ranks); False with several ranks: the eager exchange behind the graph (gloo, or the capture probe failed)
        "collective_in_graph": head["collective_captured"] if world > 1 else None,
        "config": {"workload": f"{args.workload}: {P} {kind_name}, {len(cams)} views @ {W}x{H}, "
                               f"{views_per_rank} view(s)/GPU/optimizer step ({views_per_step} views/step), "
                               "RGB+mask+orientation raster fwd+bwd + L1/DSSIM/mask/orientation/smoothness losses + Adam",
                   "gaussians": P, "views": len(cams), "width": W, "height": H, "parallelism": f"view-parallel x{world}",
                   "pipeline_state": pipeline_info,
                   "views_per_optimizer_step": views_per_step,
                   "mean_num_rendered": meanR, "mean_sum_tile_list_len": meanL,
                   "mean_num_rendered_after_tile_cull": meanR_culled, "mean_sum_tile_list_len_after_tile_cull": meanL_culled,
                   "forward_mode": "blocking" if args.blocking else "async-capacity",
                   "dispatch": "hip-graph replay" if use_graph else "eager",
                   "optimizer_steps_per_graph_launch": head["steps_per_graph"],
                   "iteration": "fused iteration" if fused is not None else "op-by-op",
                   # True: no optimizer launch -- the backward's lanes apply Adam to the elements whose gradient they finish
                   "adam_in_backward_lanes": head["inline_adam"],
                   "raster_passes_per_iter": 1 if getattr(opt, "single_pass", True) else 3},
        "render_ms_per_view": render_ms,                       # render(): the drop-in call, eager, ~20 host-side tensor ops per view
        "render_ms_per_view_frame_renderer": fs["render_graph_ms"],   # gaussian_renderer.frames: the same frames as one graph per view
        "frame_renderer_equals_render": fs["render_graph_equal"], "frame_renderer_frames_per_launch": fs["render_graph_frames"],
        "trained_state": trained,
    }
    if kern:
        ch = 7 if getattr(opt, "single_pass", True) else 3
        result["roofline"], result["roofline_blend_fwd"] = roofline_block(kern, fs, W, H, ch, args.workload)
        result["kernel_timing"] = {"method": "HIP event pairs on the launch stream, calibrated bracket cost subtracted",
                                   "bracket_cost_us": rt.lib().hgs_prof_bracket_overhead_ms() * 1e3}
        result["kernel_us_per_launch"] = {k: (v[0] / v[1] * 1e3 if v[1] else 0.0) for k, v in kern.items()}
        result["kernel_ms_per_iter"] = {k: v[0] / kern_steps for k, v in kern.items()}
    result.update(cpu_result)
    if c3_leg is not None:
        result["roofline_c3"] = c3_leg
    if pipeline_legs is not None:
        result["pipeline_states"] = pipeline_legs
    print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
