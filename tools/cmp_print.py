"""Summary of tools/cmp_r1.sh: tools/cmp_print.py [tree] [out.json]"""
import json, sys
tree = sys.argv[1] if len(sys.argv) > 1 else "_r1"
out = {}
for w in ("north_star", "c2", "c3", "c4"):
    for v in (tree, "cur"):
        vals, kern = [], {}
        for rep in (1, 2):
            try:
                d = json.loads(open(f"gpurun_out/cmp_{v}_{w}_{rep}.json").read())
                vals.append(d["value"])
                for k, t in d["kernel_us_per_launch"].items():
                    kern.setdefault(k.replace("_kernel", ""), []).append(t)
            except Exception:
                pass
        if vals:
            out.setdefault(w, {})[v] = {"iters_per_s": [round(x, 1) for x in vals],
                                        "kernel_us": {k: round(sum(t) / len(t), 1) for k, t in kern.items() if sum(t) > 0}}
            print(f"{w:10s} {v:5s} {out[w][v]['iters_per_s']}  " + "  ".join(f"{k} {t}" for k, t in out[w][v]["kernel_us"].items()
                                                                               if k in ("scatter", "sort_tiles", "blend_fwd", "blend_bwd", "preprocess_fwd", "preprocess_bwd")))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
