"""Aggregates the rocprofv3 --pmc passes of tools/pmc_raster.sh per kernel and prints derived ratios.
  python tools/pmc_aggregate.py <out.json> [last_n]
last_n: average only the LAST n dispatches of every kernel (rows ordered by dispatch id) -- for `raster_only.py ... train=N`, whose
process also holds the N training iterations in front of the passes that are to be measured."""
import csv, glob, collections, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
keys = ['blend_fwd_kernel<7>', 'blend_bwd_kernel<7>', 'sort_tiles_kernel', 'preprocess_fwd_kernel', 'scatter_kernel', 'row_reduce_kernel',
        'preprocess_bwd_kernel']
agg = {k: collections.defaultdict(list) for k in keys}
dur = {k: [] for k in keys}
last_n = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for f in glob.glob(os.path.join(ROOT, 'gpurun_out/pmc_*/*/*_counter_collection.csv')):
    rows = list(csv.DictReader(open(f)))
    if last_n:
        rows.sort(key=lambda r: int(r.get('Dispatch_Id', 0)))
    for r in rows:
        for k in keys:
            if k in r['Kernel_Name'] or k.replace('>', ',') in r['Kernel_Name']:   # (blend_bwd_kernel<7> also matches <7, true>)
                agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
                if r['Counter_Name'] in ('SQ_WAVES', 'FETCH_SIZE'):
                    dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
if last_n:
    for k in keys:
        dur[k] = dur[k][-last_n:]
        for c in list(agg[k]):
            agg[k][c] = agg[k][c][-last_n:]
out = {}
for k in keys:
    d = (dur[k] if last_n else dur[k][2:]) or dur[k] or [0]
    out[k] = {'dur_us': sum(d) / len(d)}
    for c, v in agg[k].items():
        vv = v if last_n else (v[1:] if len(v) > 2 else v)
        out[k][c] = sum(vv) / len(vv)
for k in keys[:2]:
    d = out[k]
    if 'SQ_WAVES' not in d: continue
    g = lambda n: d.get(n, 0.0)
    wc = max(g('SQ_WAVE_CYCLES'), 1.0)
    print(k, 'dur_us %.1f' % d['dur_us'], 'waves', int(g('SQ_WAVES')))
    print('   per wave: VALU %.0f SALU %.0f SMEM %.0f LDS %.0f VMEM_RD %.1f VMEM_WR %.1f' % tuple(
        g(n) / g('SQ_WAVES') for n in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_SMEM', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR')))
    print('   wave-cycle fractions: wait_any %.2f wait_inst_any %.2f active_any %.2f wait_inst_lds %.2f' % (
        g('SQ_WAIT_ANY') / wc, g('SQ_WAIT_INST_ANY') / wc, g('SQ_ACTIVE_INST_ANY') / wc, g('SQ_WAIT_INST_LDS') / wc))
    print('   lane util %.2f  lds bank conflict cycles %.0f  HBM MB (2*FETCH_SIZE+WRITE_SIZE, KB units) %.1f' % (
        g('SQ_THREAD_CYCLES_VALU') / max(1.0, g('SQ_ACTIVE_INST_VALU') * 64), g('SQ_LDS_BANK_CONFLICT'),
        (2 * g('FETCH_SIZE') + g('WRITE_SIZE')) / 1024))
    print('   mean resident waves/CU %.1f' % (g('SQ_WAVE_CYCLES') * 4 / max(1.0, g('GRBM_GUI_ACTIVE') / 8 * 256) ))
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out/pmc_summary.json'), 'w'), indent=1)
