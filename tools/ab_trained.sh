#!/bin/bash
# same-box A/B of library variants on the north_star line INCLUDING its trained-state leg: tools/ab_trained.sh "<variants>" [kernel ...]
#   variant = default | <suffix of libhgs_<suffix>.so>; prints it/s and the named kernels' us per launch, headline / trained state
mkdir -p gpurun_out
V=$1; shift
for rep in 1 2; do
 for v in $V; do
  [ "$v" = "default" ] && sfx="" || sfx="_$v"
  HGS_LIB=$PWD/hair-gs_amd/libhgs$sfx.so timeout 600 python bench.py --steps 200 --warmup 10 --repeats 3 --sustained-seconds 0 --no-cpu-baseline --no-c3-leg 2>/dev/null | tail -1 > gpurun_out/abt_${v}_$rep.json
 done
done
python - "$V" "$@" <<'PY'
import json, sys
ks = sys.argv[2:] or ["scatter_kernel", "preprocess_fwd_kernel", "sort_tiles_kernel"]
for v in sys.argv[1].split():
    for rep in (1, 2):
        try:
            d = json.load(open(f"gpurun_out/abt_{v}_{rep}.json"))
            t = d["trained_state"]
            print(f"{v:8s} head {d['value']:7.1f} it/s " + " ".join(f"{k.replace('_kernel','')} {d['kernel_us_per_launch'][k]:5.1f}" for k in ks)
                  + f" | trained {t['gaussians']} seg {t['value']:7.1f} it/s " + " ".join(f"{k.replace('_kernel','')} {t['kernel_us_per_launch'][k]:5.1f}" for k in ks))
        except Exception as e:
            print(v, rep, "ERR", e)
PY
