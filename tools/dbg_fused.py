import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import torch
from arguments import OptimizationParams
from hgs_runtime.strand_step import FusedStrandStep
from loss.losses import loss_function_single_pass
from synthetic import build_workload
from utils.general import safe_state
safe_state(True)
model, cams, _ = build_workload("tiny", device="cuda", with_targets=True)
opt = OptimizationParams(); model.training_setup(opt)
bg = torch.zeros(3, device="cuda")
params = [model._endpoints, model._width, model._opacity, model._mask, model._features_dc, model._features_rest]
fused = FusedStrandStep(model, cams, opt, bg)
cam = cams[2]
loss, terms, pkg = loss_function_single_pass(model, cam, opt, bg); loss.backward()
ref = [p.grad.clone() for p in params]
for p in params: p.grad = None
model._derived = None
fused.views.select(2); fl, _ = fused.loss(); fl.backward()
for n, p, r in zip("ep w op m dc rest".split(), params, ref):
    d = (p.grad - r).abs()
    i = int(d.reshape(-1).argmax())
    print(n, "ref max", float(r.abs().max()), "fused max", float(p.grad.abs().max()), "maxdiff", float(d.max()), "at", i,
          "ref", float(r.reshape(-1)[i]), "fused", float(p.grad.reshape(-1)[i]), "sh deg", model.active_sh_degree)
print("image mean", float(pkg["render"].mean()), "terms", {k: float(v) for k, v in terms.items()})
