#!/usr/bin/env python3
"""Round measurements beyond the headline bench line: measured copy bandwidth, render PSNR against the CPU
oracle, distCUDA2 throughput, the other BASELINE.json configurations, CPU baselines B2-B4.  Prints one JSON object."""
import json, math, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import numpy as np
import torch


def copy_bandwidth():
    n = 1 << 29                                   # 2 GiB fp32 source + 2 GiB destination
    a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        b.copy_(a)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 10
    return 2 * 4 * n / dt / 1e9                    # read + write GB/s


def psnr_vs_oracle(workload):
    from oracle import hgs_oracle as O
    from gaussian_renderer import render
    from synthetic import build_workload
    model, cams, _ = build_workload(workload, device="cuda", seed=0, with_targets=False, n_views=2)
    cam, bg = cams[0], torch.zeros(3, device="cuda")
    with torch.no_grad():
        img = render(cam, model, bg)["render"].cpu().numpy()
        s = dict(means3D=model.get_xyz.cpu().numpy(), opacities=model.get_opacity.cpu().numpy().reshape(-1),
                 scales=model.get_scaling.cpu().numpy(), rotations=model.get_rotation.cpu().numpy(), cov3D_precomp=None,
                 viewmatrix=cam.world_view_transform.cpu().numpy(), projmatrix=cam.full_proj_transform.cpu().numpy(),
                 campos=cam.camera_center.cpu().numpy(), bg=np.zeros(3, np.float32), tanfovx=float(np.tan(cam.FoVx * 0.5)),
                 tanfovy=float(np.tan(cam.FoVy * 0.5)), W=cam.image_width, H=cam.image_height,
                 sh_degree=model.active_sh_degree, scale_modifier=1.0, shs=model.get_features.cpu().numpy(), colors_precomp=None)
    t = time.perf_counter()
    ref = O.forward(s)["out_color"]
    t_cpu = time.perf_counter() - t
    mse = float(np.mean((img.astype(np.float64) - ref.astype(np.float64)) ** 2))
    return {"psnr_db": 10 * math.log10(1.0 / mse) if mse > 0 else float("inf"), "max_abs_diff": float(np.abs(img - ref).max()),
            "oracle_forward_s": t_cpu}


def knn_throughput():
    from simple_knn._C import distCUDA2
    out = {}
    g = torch.Generator(device="cuda").manual_seed(0)
    for P in (50000, 200000, 1000000):
        x = torch.rand(P, 3, device="cuda", generator=g)
        distCUDA2(x); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(5):
            distCUDA2(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 5
        out[f"P{P}"] = {"ms": dt * 1e3, "Mpoints_per_s": P / dt / 1e6, "GBps_vs_80B_per_point": 80 * P / dt / 1e9}
    return out


def bench_line(workload, extra=()):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "100", "--warmup", "10",
           "--repeats", "3", "--sustained-seconds", "0", "--no-cpu-baseline", "--trained-iters", "0", "--no-c3-leg", *extra]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=1500)
    line = [l for l in p.stdout.strip().split("\n") if l.startswith("{")]
    if not line:
        return {"error": (p.stderr or p.stdout)[-400:]}
    d = json.loads(line[-1])
    k = d.get("kernel_us_per_launch", {})
    return {"iters_per_s": d["value"], "ms_per_step": d["ms_per_step"], "render_ms_per_view": d["render_ms_per_view"],
            "render_ms_per_view_frame_renderer": d.get("render_ms_per_view_frame_renderer"),
            "gaussians": d["config"]["gaussians"], "mean_num_rendered": d["config"]["mean_num_rendered"],
            "mean_num_rendered_after_tile_cull": d["config"].get("mean_num_rendered_after_tile_cull"),
            "iteration": d["config"].get("iteration"), "scaling": d.get("scaling"), "repeats": d.get("repeats"),
            "blend_fwd_us": k.get("blend_fwd_kernel"), "blend_bwd_us": k.get("blend_bwd_kernel"),
            "sort_tiles_us": k.get("sort_tiles_kernel"), "scatter_us": k.get("scatter_kernel"),
            "preprocess_fwd_us": k.get("preprocess_fwd_kernel"), "preprocess_bwd_us": k.get("preprocess_bwd_kernel"),
            "ssim_fwd_us": k.get("ssim_l1_fwd_kernel"), "ssim_bwd_us": k.get("ssim_l1_bwd_kernel"),
            "roofline_frac": d.get("roofline", {}).get("frac"),
            "algorithmic_bytes_per_launch": d.get("roofline", {}).get("algorithmic_bytes_per_launch")}


def main():
    out = {"device": torch.cuda.get_device_name(0), "host_cores": os.cpu_count()}
    out["copy_bandwidth_GBps"] = copy_bandwidth()
    out["psnr_vs_oracle"] = {w: psnr_vs_oracle(w) for w in ("c2", "north_star")}
    out["distCUDA2"] = knn_throughput()
    torch.cuda.empty_cache()
    out["bench"] = {w: bench_line(w) for w in ("north_star", "c2", "c3", "c5", "c4")}
    out["bench"]["north_star_strong_8_views_per_step"] = bench_line("north_star", ("--scaling", "strong", "--no-kernel-timing"))
    out["bench"]["north_star_op_by_op"] = bench_line("north_star", ("--op-by-op",))
    out["bench"]["north_star_three_pass_eager_blocking"] = bench_line("north_star", ("--op-by-op", "--three-pass", "--blocking"))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cpu_baselines.py")], capture_output=True, text=True, timeout=1500)
    try:
        out["cpu_baselines"] = json.loads([l for l in p.stdout.strip().split("\n") if l.startswith("{")][-1])
    except Exception:
        out["cpu_baselines"] = {"error": (p.stderr or p.stdout)[-400:]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
