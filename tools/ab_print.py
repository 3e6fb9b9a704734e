"""Prints the A/B results of tools/ab_run.sh: tools/ab_print.py "<variants>" "<workloads>" [kernel ...]"""
import json, sys
variants, workloads = sys.argv[1].split(), sys.argv[2].split()
kernels = sys.argv[3:] or ["blend_fwd_kernel", "blend_bwd_kernel", "sort_tiles_kernel", "scatter_kernel"]
for w in workloads:
    for v in variants:
        for rep in (1, 2):
            try:
                d = json.loads(open(f"gpurun_out/ab_{v.replace(chr(64), chr(95)).replace(chr(44), chr(95))}_{w}_{rep}.json").read())
                k = d["kernel_us_per_launch"]
                print(f"{v:10s} {w:10s} {d['value']:8.1f} it/s  render {d['render_ms_per_view']:.3f} ms  " + "  ".join(f"{n.replace('_kernel', '')} {k.get(n, 0):6.1f}" for n in kernels))
            except Exception as e:
                print(v, w, rep, "ERR", e)
