"""Synchronised time per SOURCE LINE of the topology operators during the events of a training run (a line tracer that
synchronises the device at every line of the chosen functions: slow, but it charges device work to the line that issued it)."""
import os, sys, time, collections, linecache
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import torch
import train
from arguments import OptimizationParams
from synthetic import build_workload, build_pipeline_state, PIPELINE_STATES
from utils.general import safe_state
names = set((sys.argv[3] if len(sys.argv) > 3 else "split_strategy,compute_endpoint_pair_to_merge,merge_collapsed_segments,_sort_spatially_device,compute_strands_info,walk_chains_torch").split(","))
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
safe_state(True)
wl = sys.argv[1] if len(sys.argv) > 1 else "north_star"
opt = OptimizationParams()
if wl in PIPELINE_STATES:
    model, cams, extent, _ = build_pipeline_state(wl, device="cuda", seed=0)
    opt.iterations = 5000
    opt._finalise()
else:
    model, cams, extent = build_workload(wl, device="cuda", seed=0, n_views=16)
model.training_setup(opt)
acc, hits = collections.Counter(), collections.Counter()
state = {}


def local(frame, event, arg):
    key = id(frame)
    now_needed = event in ("line", "return")
    if now_needed and not torch.cuda.is_current_stream_capturing():
        torch.cuda.synchronize()
        t = time.perf_counter()
        prev = state.get(key)
        if prev is not None:
            acc[prev[0]] += t - prev[1]; hits[prev[0]] += 1
        if event == "line":
            state[key] = ((frame.f_code.co_filename, frame.f_code.co_name, frame.f_lineno), time.perf_counter())
        else:
            state.pop(key, None)
    return local


def tracer(frame, event, arg):
    if event == "call" and frame.f_code.co_name in names and not torch.cuda.is_current_stream_capturing():
        return local
    return None


warm = int(sys.argv[4]) if len(sys.argv) > 4 else 0       # iterations run before the tracer is on (first uses of torch kernels load code objects: tens of ms each)
if warm:
    train.training(model, cams, opt, iterations=warm, extent=extent)
sys.settrace(tracer)
train.training(model, cams, opt, iterations=iters, extent=extent, start_iteration=warm)
sys.settrace(None)
by_fn = collections.defaultdict(list)
for (fn, name, ln), v in acc.items():
    by_fn[name].append((v, ln, fn))
for name, rows in by_fn.items():
    print(f"== {name}: {1e3 * sum(r[0] for r in rows):.1f} ms in total")
    for v, ln, fn in sorted(rows, reverse=True)[:14]:
        print(f"   {1e3 * v:8.2f} ms {hits[(fn, name, ln)]:5d}x  {ln:4d}: {linecache.getline(fn, ln).strip()[:130]}")
