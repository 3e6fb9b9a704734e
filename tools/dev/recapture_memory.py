#!/usr/bin/env python3
"""Does a long run with re-captures hold on to the memory of the graphs it has dropped?  (dev aid; round 6: BASELINE config 4's
three stages ran out of 288 GB after ~100 re-captures)
  python tools/dev/recapture_memory.py [workload=north_star] [iterations=1500] [gc=0|1]"""
import gc
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import torch  # noqa: E402

from arguments import OptimizationParams  # noqa: E402
from synthetic import build_workload  # noqa: E402
from train import training  # noqa: E402
from utils.general import safe_state  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "north_star"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
do_gc = len(sys.argv) > 3 and sys.argv[3] == "1"
safe_state(True)
model, cams, extent = build_workload(wl, device="cuda")
opt = OptimizationParams()
model.training_setup(opt)
done = 0
while done < n:
    training(model, cams, opt, iterations=100, extent=extent, start_iteration=done)
    done += 100
    if do_gc:
        gc.collect()
    torch.cuda.synchronize()
    if done % 500 == 0:
        import collections
        snap = torch.cuda.memory_snapshot()
        agg = collections.Counter()
        for seg in snap:
            live = sum(b["size"] for b in seg["blocks"] if b["state"] == "active_allocated")
            agg[(str(seg.get("segment_pool_id")), seg.get("stream"), seg["total_size"] >> 20, "live" if live else "free")] += 1
        for k, v in sorted(agg.items(), key=lambda kv: -kv[0][2] * kv[1])[:14]:
            print("   segments: pool", k[0], "stream", k[1], f"{k[2]} MiB x {v}", k[3], flush=True)
    print(f"it {done}: segments {model.get_xyz.shape[0]}  allocated {torch.cuda.memory_allocated() / 2**30:.2f} GiB  "
          f"reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB  gc objects {len(gc.get_objects())}", flush=True)
