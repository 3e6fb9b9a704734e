"""Where the topology operators synchronise with the device: every aten operator that blocks on a device-to-host copy
(_local_scalar_dense = .item() / int() / bool(); nonzero, also behind boolean-mask indexing; copies to the host) during one
densification + merging event on a trained strand model, grouped by the Python line that issued it."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import torch
from arguments import OptimizationParams
from synthetic import build_workload
from train import training
from utils.general import safe_state
safe_state(True)
model, cams, extent = build_workload("north_star", device="cuda", seed=0, n_views=16)
opt = OptimizationParams()
model.training_setup(opt)
training(model, cams, opt, iterations=int(sys.argv[1]) if len(sys.argv) > 1 else 1450, extent=extent)
torch.cuda.synchronize()
import traceback
sites = collections.Counter()


def note(kind):
    for f in reversed(traceback.extract_stack()[:-2]):
        if "hair-gs_amd/scene" in f.filename or "hgs_runtime" in f.filename:
            sites[(kind, f"{os.path.basename(f.filename)}:{f.lineno} {f.name}")] += 1
            return
    sites[(kind, "?")] += 1


def wrap(owner, attr, kind, cond=lambda *a, **k: True):
    orig = getattr(owner, attr)

    def w(*a, **k):
        if cond(*a, **k):
            note(kind)
        return orig(*a, **k)
    setattr(owner, attr, w)


T = torch.Tensor
on_gpu = lambda t, *a, **k: isinstance(t, T) and t.is_cuda
has_mask = lambda t, i, *a: on_gpu(t) and any(isinstance(x, T) and x.dtype in (torch.bool, torch.uint8) for x in (i if isinstance(i, tuple) else (i,)))
wrap(T, "item", "item", on_gpu); wrap(T, "cpu", "cpu", on_gpu); wrap(T, "tolist", "tolist", on_gpu); wrap(T, "numpy", "numpy", on_gpu)
wrap(T, "__bool__", "bool()", on_gpu); wrap(T, "__int__", "int()", on_gpu); wrap(T, "__float__", "float()", on_gpu); wrap(T, "__index__", "index()", on_gpu)
wrap(T, "nonzero", "nonzero", on_gpu); wrap(torch, "nonzero", "nonzero", on_gpu); wrap(torch, "unique", "unique", on_gpu)
wrap(T, "__getitem__", "x[mask]", has_mask); wrap(T, "__setitem__", "x[mask] = ", has_mask)
for name, fn in (("densification", lambda: model.densification(extent, None, None)), ("merging", lambda: model.merging())):
    sites.clear()
    fn()
    torch.cuda.synchronize()
    print(f"== {name}: {sum(sites.values())} synchronising calls")
    for (op, site), n in sites.most_common(60):
        print(f"   {n:3d}  {op:12s} {site}")
