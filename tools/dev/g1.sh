cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_raster.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r4_t1.log
bash tools/ab_run.sh "r3 default" "north_star c3" 
python tools/ab_print.py "r3 default" "north_star c3" blend_fwd_kernel blend_bwd_kernel preprocess_bwd_kernel sort_tiles_kernel > gpurun_out/r4_ab1.txt 2>&1
