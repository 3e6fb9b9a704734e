#!/bin/bash
# Round-5 evidence in ONE gpurun call (outputs under gpurun_out/r05_*; copy what is to be judged into profiles/):
#   kernel statistics of the default bench command, --pmc passes over the rasterizer at north_star, C3 (BASELINE config 3) AND the
#   trained state (1000 iterations of the full loop), the counter table of all kernels of the iteration, the default bench line
#   (headline + trained state + roofline_c3), the soak runs, every BASELINE workload (tools/measure_all.py).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
bash tools/profile_round.sh r05                       # -> r05_kernel_stats_top45.csv, r05_bench.json, r05_pmc_raster_north_star.json
rm -rf gpurun_out/pmc_*
bash tools/pmc_raster.sh c3 > gpurun_out/r05_pmc_raster_c3.txt 2>&1
cp gpurun_out/pmc_summary.json gpurun_out/r05_pmc_raster_c3.json
rm -rf gpurun_out/pmc_*/
bash tools/pmc_raster.sh north_star train=1000 > gpurun_out/r05_pmc_raster_north_star_trained.txt 2>&1
cp gpurun_out/pmc_summary.json gpurun_out/r05_pmc_raster_north_star_trained.json
rm -rf gpurun_out/pmc_*/
bash tools/pmc_generic.sh pmcit tools/eager_steps.py north_star 12
python3 tools/pmc_table.py pmcit hair_preprocess_fwd_kernel scatter_kernel sort_tiles_kernel blend_fwd_kernel ssim_l1_fwd_kernel pix_fwd_kernel ssim_l1_bwd_kernel blend_bwd_kernel preprocess_bwd_kernel strand_gather_kernel strand_bwd_kernel adam_kernel > gpurun_out/r05_pmc_iteration_north_star.txt 2>&1
rm -rf gpurun_out/pmcit_*/
python3 tools/soak.py north_star 3000 > gpurun_out/r05_soak_north_star.txt 2>&1
python3 tools/soak.py c3 1500 > gpurun_out/r05_soak_c3.txt 2>&1
