#!/bin/bash
# Round-6 evidence in ONE gpurun call (outputs under gpurun_out/r06_*; copy what is to be judged into profiles/):
#   kernel statistics of the default bench command (north_star) AND of the two pipeline states (bench.py --workload stage1_1080p /
#   stage3_merged: the states the three-stage workflow lives in), --pmc passes over the rasterizer at north_star, C3, the trained
#   state and both pipeline states, the counter table of all kernels of the iteration, the default bench line (headline +
#   trained_state + roofline_c3 + pipeline_states), the soak runs, every BASELINE workload (tools/measure_all.py).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
bash tools/profile_round.sh r06                       # -> r06_kernel_stats_top45.csv, r06_bench.json, r06_pmc_raster_north_star.json
for st in stage1_1080p stage3_merged; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r06_$st -- python3 bench.py --workload $st --steps 100 --warmup 10 --repeats 3 --sustained-seconds 0 --trained-iters 0 --no-cpu-baseline --no-c3-leg --no-pipeline-legs > gpurun_out/r06_bench_$st.json 2> gpurun_out/r06_bench_$st.err
  f=$(ls gpurun_out/prof_r06_$st/*/*_kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && head -45 "$f" > gpurun_out/r06_kernel_stats_${st}_top45.csv
  rm -rf gpurun_out/prof_r06_$st
  rm -rf gpurun_out/pmc_*/
  bash tools/pmc_raster.sh $st > gpurun_out/r06_pmc_raster_$st.txt 2>&1
  cp gpurun_out/pmc_summary.json gpurun_out/r06_pmc_raster_$st.json
done
rm -rf gpurun_out/pmc_*/
bash tools/pmc_raster.sh c3 > gpurun_out/r06_pmc_raster_c3.txt 2>&1
cp gpurun_out/pmc_summary.json gpurun_out/r06_pmc_raster_c3.json
rm -rf gpurun_out/pmc_*/
bash tools/pmc_raster.sh north_star train=1000 > gpurun_out/r06_pmc_raster_north_star_trained.txt 2>&1
cp gpurun_out/pmc_summary.json gpurun_out/r06_pmc_raster_north_star_trained.json
rm -rf gpurun_out/pmc_*/
bash tools/pmc_generic.sh pmcit tools/eager_steps.py north_star 12
python3 tools/pmc_table.py pmcit hair_preprocess_fwd_kernel scatter_kernel sort_tiles_kernel blend_fwd_kernel ssim_l1_fwd_kernel pix_fwd_kernel ssim_l1_bwd_kernel blend_bwd_kernel preprocess_bwd_kernel strand_gather_kernel row_reduce_kernel > gpurun_out/r06_pmc_iteration_north_star.txt 2>&1
rm -rf gpurun_out/pmcit_*/
python3 tools/soak.py north_star 3000 > gpurun_out/r06_soak_north_star.txt 2>&1
python3 tools/soak.py north_star 3600 > gpurun_out/r06_soak_north_star_3600.txt 2>&1
python3 tools/soak.py c3 1500 > gpurun_out/r06_soak_c3.txt 2>&1
python3 bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench_line.err
python3 tools/measure_all.py > gpurun_out/r06_measurements.json 2> gpurun_out/r06_measurements.err
