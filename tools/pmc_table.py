"""Per-kernel table of the counters collected by tools/pmc_generic.sh: python tools/pmc_table.py <tag> <kernel substring>...

HBM bytes: FETCH_SIZE tallies every memory-side read request at 64 bytes -- a 64-byte request (dword-per-lane reads of 8x8
quadrants of planar images, 64-byte gathers) in full, a 128-byte request (>= 256 contiguous bytes per wave instruction, 16 B per
lane streams) at HALF (profiles/r04_fetch_shape_probe.txt).  The table cannot know a kernel's mix of the two, so it prints the
bracket [FETCH + WRITE, 2 FETCH + WRITE]; bench.py's `traffic` resolves the blend backward's mix from its access shapes
(planes 1:1, record batches doubled).  Rounds 1-4 printed the upper end only, which overstates every kernel with 64-byte gathers."""
import csv, glob, collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, keys = sys.argv[1], sys.argv[2:]
agg = {k: collections.defaultdict(list) for k in keys}
dur = {k: [] for k in keys}
for f in glob.glob(os.path.join(ROOT, f"gpurun_out/{tag}_*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        for k in keys:
            if k in r["Kernel_Name"]:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                if r["Counter_Name"] in ("SQ_WAVES", "FETCH_SIZE"):
                    dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in keys:
    d = {c: sum(v[1:]) / max(1, len(v[1:])) if len(v) > 2 else sum(v) / max(1, len(v)) for c, v in agg[k].items()}
    if "SQ_WAVES" not in d:
        continue
    g = lambda n: d.get(n, 0.0)
    wc = max(g("SQ_WAVE_CYCLES"), 1.0)
    dd = dur[k][2:] or dur[k]
    print(k, "dur_us %.1f" % (sum(dd) / len(dd)), "waves", int(g("SQ_WAVES")))
    print("   per wave: VALU %.0f SALU %.0f SMEM %.0f LDS %.0f VMEM_RD %.1f VMEM_WR %.1f" % tuple(
        g(n) / g("SQ_WAVES") for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")))
    print("   wave-cycle fractions: wait_any %.2f wait_inst_any %.2f active_any %.2f wait_inst_lds %.2f" % (
        g("SQ_WAIT_ANY") / wc, g("SQ_WAIT_INST_ANY") / wc, g("SQ_ACTIVE_INST_ANY") / wc, g("SQ_WAIT_INST_LDS") / wc))
    print("   lane util %.2f  lds bank conflict cycles %.0f / lds active %.0f  HBM MB %.1f .. %.1f (all 64-B .. all 128-B read requests)" % (
        g("SQ_THREAD_CYCLES_VALU") / max(1.0, g("SQ_ACTIVE_INST_VALU") * 64), g("SQ_LDS_BANK_CONFLICT"), g("SQ_ACTIVE_INST_LDS"),
        (g("FETCH_SIZE") + g("WRITE_SIZE")) / 1024, (2 * g("FETCH_SIZE") + g("WRITE_SIZE")) / 1024))
    print("   mean resident waves/CU %.1f   VALU pipe util %.2f" % (
        g("SQ_WAVE_CYCLES") * 4 / max(1.0, g("GRBM_GUI_ACTIVE") / 8 * 256),
        g("SQ_INSTS_VALU") * 4 / max(1.0, g("GRBM_GUI_ACTIVE") / 8 * 1024)))
