#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel trace of the default bench command + the PMC passes over the rasterizer.
#   tools/profile_round.sh <tag>     -> gpurun_out/<tag>_kernel_stats.csv, gpurun_out/<tag>_bench.json, gpurun_out/pmc_summary.json
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --sustained-seconds 0 --trained-iters 0 --no-c3-leg --no-pipeline-legs > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
f=$(ls gpurun_out/prof_$tag/*/*_kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && head -45 "$f" > gpurun_out/${tag}_kernel_stats_top45.csv
rm -rf gpurun_out/prof_$tag
rm -rf gpurun_out/pmc_*
bash tools/pmc_raster.sh north_star
cp gpurun_out/pmc_summary.json gpurun_out/${tag}_pmc_raster_north_star.json
rm -rf gpurun_out/pmc_*/
