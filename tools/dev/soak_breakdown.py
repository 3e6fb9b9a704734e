"""Where the wall clock of a soak run goes, with synchronised timers (cProfile charges a device wait to whoever happens to
synchronise): graph replays, the event iterations (eager iteration + topology operators), captures -- and inside the events, every
operator and helper by name (nested calls are charged to both levels)."""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import torch
import train
from arguments import OptimizationParams
from synthetic import build_workload, build_pipeline_state, PIPELINE_STATES
from utils.general import safe_state
wl = sys.argv[1] if len(sys.argv) > 1 else "north_star"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
def fresh():
    safe_state(True)
    if wl in PIPELINE_STATES:       # (stage3_merged: the model Stage III starts from; the operators of a 5000-iteration Stage III)
        model, cams, extent, _ = build_pipeline_state(wl, device="cuda", seed=0, stage1_iters=int(os.environ["STAGE1_ITERS"]) if "STAGE1_ITERS" in os.environ else None)
        opt = OptimizationParams()
        opt.iterations = 5000
        opt._finalise()
        model.training_setup(opt)
        return model, cams, extent, opt
    model, cams, extent = build_workload(wl, device="cuda", seed=0, n_views=16)
    opt = OptimizationParams()
    model.training_setup(opt)
    return model, cams, extent, opt


if os.environ.get("COLD") != "1":      # a first run unmeasured: the first use of every torch kernel loads its code object (tens of ms each)
    model, cams, extent, opt = fresh()
    train.training(model, cams, opt, iterations=iters, extent=extent)
model, cams, extent, opt = fresh()
acc, cnt = collections.Counter(), collections.Counter()


def wrap(owner, name, label=None):
    orig = getattr(owner, name)
    label = label or name

    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        try:
            return orig(*a, **k)
        finally:
            torch.cuda.synchronize(); acc[label] += time.perf_counter() - t0; cnt[label] += 1
            if label == os.environ.get("EACH"):      # every call of one function, in ms
                print(f"   {label} call {cnt[label]}: {1e3 * (time.perf_counter() - t0):.2f} ms", flush=True)
    setattr(owner, name, w)


wrap(train, "_training_step", "EVENT ITERATION (training_step)")
wrap(train.GraphedStep, "capture", "CAPTURE")
wrap(train.GraphedStep, "__init__", "GraphedStep()")
M = type(model)
for n in ("densification", "merging", "clone_strategy", "split_strategy", "merge_collapsed_segments", "prune_strategy", "compute_strands_info",
          "compute_endpoint_pair_to_merge", "merge_endpoint_pairs", "sort_spatially", "prune_segments", "cat_segments", "reset_opacity",
          "get_complementary_endpoint_idx", "smoothness_index_pairs", "get_endpoint_pairs_row_indices", "_maybe_sort_spatially",
          "densify_and_split", "densify_and_clone", "prune_points", "densification_postfix"):       # (the last four: a Stage-I cloud)
    if hasattr(M, n):
        wrap(M, n)
t0 = time.perf_counter()
train.training(model, cams, opt, iterations=iters, extent=extent)
torch.cuda.synchronize()
total = time.perf_counter() - t0
print(f"{wl}: {iters} iterations in {total:.3f} s (with the timers' synchronisations) -> segments {model.get_xyz.shape[0]}")
top = acc["EVENT ITERATION (training_step)"] + acc["CAPTURE"] + acc["GraphedStep()"]
print(f"  replays and the rest of the loop: {total - top:.3f} s")
for k, v in acc.most_common():
    print(f"  {k:42s} {v:7.3f} s  {cnt[k]:4d} calls  {1e3 * v / cnt[k]:7.2f} ms each")
