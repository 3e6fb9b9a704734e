cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_frames.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r4_t7.log
bash tools/ab_run.sh "r3 default" "north_star c3 c2 c4"
python tools/ab_print.py "r3 default" "north_star c3 c2 c4" blend_fwd_kernel blend_bwd_kernel > gpurun_out/r4_ab7.txt
