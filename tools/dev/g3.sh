cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_raster.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r4_t3.log
run() { # tag lib variant workload
  HGS_BWD_VARIANT=$3 HGS_LIB=$PWD/hair-gs_amd/$2 timeout 300 python bench.py --workload $4 --steps 200 --warmup 10 --repeats 3 --sustained-seconds 0 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/x_$1_$4.json
}
for w in north_star c3 c2 c4; do
  run old libhgs.so 0 $w
  run rows libhgs.so 2 $w
  run rowsnored libhgs_rowsnored.so 2 $w
done
python - <<'PY' > gpurun_out/r4_x3.txt
import json,glob
for w in ("north_star","c3","c2","c4"):
    for t in ("old","rows","rowsnored"):
        try:
            d=json.load(open(f"gpurun_out/x_{t}_{w}.json")); k=d["kernel_us_per_launch"]
            print(f"{t:9s} {w:10s} {d['value']:8.1f} it/s bwd {k.get('blend_bwd_kernel',0):6.1f} ppb {k.get('preprocess_bwd_kernel',0):6.1f} fwd {k.get('blend_fwd_kernel',0):6.1f} sort {k.get('sort_tiles_kernel',0):6.1f}")
        except Exception as e: print(t,w,"ERR",e)
PY
