"""Prints the kernel sequence of ONE captured iteration from a rocprofv3 --kernel-trace of bench.py (gpurun_out/<dir>)."""
import csv, glob, re, sys
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_f"
f = glob.glob(d + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# an iteration = from the end of one launch of its LAST kernel to the end of the next (round 5: the endpoint gather, which also
# applies Adam; before: adam_kernel)
last = sys.argv[3] if len(sys.argv) > 3 else "strand_gather_kernel"
idx = [i for i, n in enumerate(names) if last in n]
k = int(sys.argv[2]) if len(sys.argv) > 2 else -60      # which iteration (counted from the end: skip the timing pass)
a, b = idx[k - 1], idx[k]
t0 = prev = int(rows[a]["End_Timestamp"])
tot = 0
for r in rows[a + 1:b + 1]:
    n = r["Kernel_Name"]
    n = n.split("(anonymous namespace)::")[-1] if "anonymous" in n else n.replace("void at::native::", "")
    n = re.sub(r"\s+", " ", n)[:100]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:8.1f} gap {(s - prev) / 1e3:5.1f} dur {(e - s) / 1e3:6.1f}  {n}")
    prev = e
    tot += e - s
print("kernels", b - a, "sum dur us", tot / 1e3, "span us", (int(rows[b]["End_Timestamp"]) - t0) / 1e3)
