#!/bin/bash
# Round-3 evidence in ONE gpurun call (outputs under gpurun_out/r03_*; copy what is to be judged into profiles/):
#   kernel statistics of the default bench command, --pmc passes over the rasterizer at north_star AND C3 (BASELINE config 3),
#   the counter table of all kernels of the iteration, the convergence run, the bench line with the trained-state leg.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
bash tools/profile_round.sh r03                       # -> r03_kernel_stats_top45.csv, r03_bench.json, r03_pmc_raster_north_star.json
rm -rf gpurun_out/pmc_*
bash tools/pmc_raster.sh c3 > gpurun_out/r03_pmc_raster_c3.txt 2>&1
cp gpurun_out/pmc_summary.json gpurun_out/r03_pmc_raster_c3.json
rm -rf gpurun_out/pmc_*/
bash tools/pmc_generic.sh pmcit tools/eager_steps.py north_star 12
python3 tools/pmc_table.py pmcit hair_preprocess_fwd_kernel scatter_kernel sort_tiles_kernel blend_fwd_kernel ssim_l1_fwd_kernel pix_fwd_kernel ssim_l1_bwd_kernel blend_bwd_kernel preprocess_bwd_kernel strand_bwd_kernel adam_kernel > gpurun_out/r03_pmc_iteration_north_star.txt 2>&1
rm -rf gpurun_out/pmcit_*/
python3 tools/convergence.py north_star 8 > gpurun_out/r03_convergence.json 2> gpurun_out/r03_convergence.err
python3 bench.py --trained-iters 1000 > gpurun_out/r03_bench_trained.json 2> gpurun_out/r03_bench_trained.err
python3 bench.py > gpurun_out/r03_bench_line.json 2> gpurun_out/r03_bench_line.err
