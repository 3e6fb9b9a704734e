mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_raster.py -m gpu -q -x --tb=short -k "backward or full_size or cull" 2>&1 | tail -8 > gpurun_out/t1.log
for v in "" _single _p4; do
 for w in north_star c2 c4; do
  HGS_LIB=$PWD/hair-gs_amd/libhgs$v.so timeout 300 python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/ab${v}_$w.json
 done
done
