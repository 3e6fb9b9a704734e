#!/bin/bash
# same-box A/B of library variants: tools/ab_run.sh "<variant suffixes>" "<workloads>"   ("" = the default build)
mkdir -p gpurun_out
for rep in 1 2; do
for v in $1; do
 [ "$v" = "default" ] && sfx="" || sfx="_$v"
 for w in $2; do
  HGS_LIB=$PWD/hair-gs_amd/libhgs$sfx.so timeout 300 python bench.py --workload $w --steps 200 --warmup 10 --repeats 3 --sustained-seconds 0 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ab_${v}_${w}_$rep.json
 done
done
done
