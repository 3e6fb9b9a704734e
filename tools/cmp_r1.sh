mkdir -p gpurun_out
for rep in 1 2; do
(cd _r1 && timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > ../gpurun_out/cmp_r1_$rep.json)
timeout 300 python bench.py --no-cpu-baseline --sustained-seconds 0 2>/dev/null | tail -1 > gpurun_out/cmp_cur_$rep.json
HGS_LIB=$PWD/hair-gs_amd/libhgs_noemit.so timeout 300 python bench.py --no-cpu-baseline --sustained-seconds 0 2>/dev/null | tail -1 > gpurun_out/cmp_noemit_$rep.json
done
