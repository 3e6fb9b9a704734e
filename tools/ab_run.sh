#!/bin/bash
# same-box A/B of library variants: tools/ab_run.sh "<variants>" "<workloads>"
#   variant = default | <suffix of libhgs_<suffix>.so>, optionally followed by @min,max,target (segment policy of that run);
#   results: gpurun_out/ab_<variant with , and @ as _>_<workload>_<rep>.json (tools/ab_print.py)
mkdir -p gpurun_out
for rep in 1 2; do
for v in $1; do
 lib=${v%%@*}; pol=""; [ "$lib" != "$v" ] && pol=${v#*@}
 [ "$lib" = "default" ] && sfx="" || sfx="_$lib"
 tag=${v//[@,]/_}
 for w in $2; do
  HGS_SEG_POLICY=$pol HGS_LIB=$PWD/hair-gs_amd/libhgs$sfx.so timeout 300 python bench.py --workload $w --steps 200 --warmup 10 --repeats 3 --sustained-seconds 0 --no-cpu-baseline --trained-iters 0 --no-c3-leg --no-pipeline-legs 2>/dev/null | tail -1 > gpurun_out/ab_${tag}_${w}_$rep.json
 done
done
done
