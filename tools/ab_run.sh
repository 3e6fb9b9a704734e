mkdir -p gpurun_out
for rep in 1 2; do
for v in "" _nogrec; do
 for w in north_star c2 c4; do
  HGS_LIB=$PWD/hair-gs_amd/libhgs$v.so timeout 300 python bench.py --workload $w --steps 200 --warmup 10 --repeats 3 --sustained-seconds 0 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ab2${v}_${w}_$rep.json
 done
done
done
