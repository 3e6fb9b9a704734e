#!/bin/bash
# Same-box A/B of an environment switch (run on the GPU box):  tools/ab_env.sh VAR "workloads" [reps]
# e.g. tools/ab_env.sh HGS_FUSE_PARAM_BACKWARD "north_star c3 c2 c4" 2   -> gpurun_out/abenv_<VAR>_<0|1>_<workload>_<rep>.json
VAR=$1; WLS=${2:-north_star}; REPS=${3:-2}
mkdir -p gpurun_out
for rep in $(seq 1 $REPS); do for w in $WLS; do for v in 0 1; do
  env $VAR=$v timeout 400 python bench.py --workload $w --steps 200 --warmup 10 --repeats 3 --sustained-seconds 0 --no-cpu-baseline --trained-iters 0 --no-c3-leg --no-pipeline-legs 2>/dev/null | tail -1 > gpurun_out/abenv_${VAR}_${v}_${w}_$rep.json
done; done; done
python3 - "$VAR" <<'PY'
import glob, json, sys, os
var = sys.argv[1]
for f in sorted(glob.glob(f"gpurun_out/abenv_{var}_*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    k = r.get("kernel_us_per_launch", {})
    print(f"{os.path.basename(f):60s} {r['value']:8.1f} it/s  " + " ".join(f"{n.replace('_kernel','')[:14]} {v:5.1f}" for n, v in k.items() if v))
PY
