"""Does the state a bench run ends in depend on the sustained leg?  Replays bench.py's protocol (warm-up, snapshot, regions
restarted from the snapshot, sustained chunks, eager continuation) and prints a digest of parameters + Adam state at every
stage: equal stages must give equal digests (the fused iteration has no float atomics)."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import torch
from arguments import OptimizationParams
from synthetic import build_workload
from train import GraphedStep, ViewSampler, training_step
from hgs_runtime.strand_step import ViewTable, fused_step_for
from diff_gaussian_rasterization import _C as raster
from utils.general import safe_state
safe_state(True)
raster.set_async(True)
model, cams, extent = build_workload("north_star", device=torch.device("cuda"), seed=0, n_views=8)
opt = OptimizationParams(); opt.enable_topology = False
model.training_setup(opt)
bg = torch.zeros(3, device="cuda")
sampler = ViewSampler(cams, seed=0)
views = ViewTable(cams); fused = fused_step_for(model, views, opt, bg); fused.defer_tail = True
gs = GraphedStep(model, cams, opt, bg, extent=extent, views=views, steps_per_graph=8)
gs.capture(cams, iteration=1)
it = 0
def run(n):
    global it
    while n >= 8:
        gs.step_many([sampler.next() for _ in range(8)], it + 1); it += 8; n -= 8
    for _ in range(n):
        it += 1; gs.step(sampler.next(), it)
def digest():
    torch.cuda.synchronize()
    h = hashlib.sha1()
    for g in model.optimizer.param_groups:
        p = g["params"][0]
        h.update(p.detach().cpu().numpy().tobytes())
        for v in model.optimizer.state[p].values():
            if torch.is_tensor(v): h.update(v.detach().cpu().numpy().tobytes())
    return h.hexdigest()[:12]
run(10)
torch.cuda.synchronize()
ts = [p.data for g in model.optimizer.param_groups for p in g["params"]]
for g in model.optimizer.param_groups:
    for p in g["params"]:
        ts += [v for v in model.optimizer.state.get(p, {}).values() if torch.is_tensor(v)]
ts += [model.max_radii2D, model.xyz_gradient_accum, model.denom]
snap = [t.clone() for t in ts]
it0 = it
def restore():
    global it
    with torch.no_grad():
        for t, s in zip(ts, snap): t.copy_(s)
    it = it0; sampler.rng.seed(12345); sampler.stack = []; torch.cuda.synchronize()
print("after warmup", digest())
for rep in range(3):
    restore(); run(100); print("region", rep, digest())
restore()
for chunk in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    with torch.no_grad():
        for t, s in zip(ts, snap): t.copy_(s, non_blocking=True)
    it = it0
    run(100)
print("after sustained chunks", digest())
restore(); print("restored", digest())
run(100); print("region after sustained", digest())
restore()
for _ in range(50):
    it += 1
    training_step(model, sampler.next(), opt, bg, it, extent=extent, fused=fused)
print("after 50 eager steps", digest())
restore()
for _ in range(50):
    it += 1
    training_step(model, sampler.next(), opt, bg, it, extent=extent, fused=fused)
print("after 50 eager steps again", digest())
for rep in range(3):
    restore()
    for _ in range(50):
        it += 1
        training_step(model, sampler.next(), opt, bg, it, extent=extent, fused=fused)
    print("50 eager steps, no sync, run", rep, digest())
for rep in range(2):
    restore()
    for _ in range(50):
        it += 1
        training_step(model, sampler.next(), opt, bg, it, extent=extent, fused=fused)
        torch.cuda.synchronize()
    print("50 eager steps, sync per step, run", rep, digest())

