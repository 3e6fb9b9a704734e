#!/bin/bash
# FETCH_SIZE calibration on the blend kernels' access shapes (run on the GPU box): tools/probes/fetch_shape_probe.sh
# -> gpurun_out/fetch_shape_probe.txt (counter / known bytes per kernel; copied to profiles/ by hand)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
BIN=tools/probes/fetch_shape_probe
[ -x $BIN ] || hipcc --offload-arch=gfx950 -O3 tools/probes/fetch_shape_probe.hip -o $BIN
mkdir -p gpurun_out
./$BIN > gpurun_out/fetch_shape_bytes.txt
for set in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  t=$(echo $set | cut -c1-12 | tr " " "_")
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/fsp_$t -- ./$BIN > /dev/null 2>&1
done
python3 - <<'PY' > gpurun_out/fetch_shape_probe.txt
import csv, glob, collections
known = {}
for line in open("gpurun_out/fetch_shape_bytes.txt"):
    p = line.split()
    known[p[0]] = int(p[1])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/fsp_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        if name == "k_quadrant_dword" and int(r["Grid_Size"]) < 8160 * 256:
            name = "k_quadrant_sparse"
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# counter values per launch (mean of the launches after the first) against the bytes the kernel reads")
for name, by in sorted(known.items()):
    c = {k: (sum(v[1:]) / max(1, len(v) - 1) if len(v) > 1 else v[0]) for k, v in agg[name].items()}
    fs = c.get("FETCH_SIZE")
    line = f"{name:18s} bytes {by/1e6:9.2f} MB"
    if fs is not None:
        line += f"  FETCH_SIZE {fs*1024/1e6:9.2f} MB (KB units)  ratio {fs*1024/by:5.3f}"
    if "TCC_EA0_RDREQ_sum" in c:
        rd, rd32 = c["TCC_EA0_RDREQ_sum"], c.get("TCC_EA0_RDREQ_32B_sum", 0.0)
        line += f"  RDREQ {rd:12.0f} (32B: {rd32:10.0f})  bytes/RDREQ {by/max(rd,1):6.1f}"
    if "TCC_HIT_sum" in c:
        line += f"  L2 hit {c['TCC_HIT_sum']/max(1.0, c['TCC_HIT_sum']+c.get('TCC_MISS_sum',0.0)):5.3f}"
    print(line)
PY
cat gpurun_out/fetch_shape_probe.txt
