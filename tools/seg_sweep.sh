#!/bin/bash
# sweep of the segment policy (hgs_set_segment_policy[, max pieces]) over workloads: gpurun_out/seg_<policy>_<workload>.json
mkdir -p gpurun_out
for pol in ${POLICIES:-128,1024,2048,16 128,1024,2048,1}; do
  for w in ${WORKLOADS:-north_star c2 c3 c4}; do
    HGS_SEG_POLICY=$pol timeout 300 python bench.py --workload $w --steps 100 --warmup 10 --repeats 3 --sustained-seconds 0 --no-cpu-baseline --trained-iters 0 --no-c3-leg 2>/dev/null | tail -1 > gpurun_out/seg_${pol//,/_}_$w.json
  done
done
