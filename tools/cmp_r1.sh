#!/bin/bash
# Same-box comparison of this tree with another checkout of the repository (default: _r1, a worktree of the round-1 head):
#   tools/cmp_r1.sh [tree] -> gpurun_out/cmp_<tree|cur>_<workload>_<rep>.json   (print: tools/cmp_print.py)
# Boxes of the pool differ by a few per cent from call to call (and the workload drifts while it trains): only numbers of ONE
# gpurun call, taken with the same bench protocol, compare.
tree=${1:-_r1}
mkdir -p gpurun_out
for rep in 1 2; do
 for w in ${WORKLOADS:-north_star c2 c3 c4}; do
  (cd $tree && timeout 400 python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > ../gpurun_out/cmp_${tree}_${w}_$rep.json)
  timeout 400 python bench.py --workload $w --steps 100 --warmup 10 --repeats 3 --sustained-seconds 0 --no-cpu-baseline --trained-iters 0 --no-c3-leg 2>/dev/null | tail -1 > gpurun_out/cmp_cur_${w}_$rep.json
 done
done
