"""Timestamps of pix_fwd_kernel's side workgroups (library built with -DHGS_PIX_TRACE=1): when do the block-list builder
and the head's first sums start and end relative to the pixel workgroups?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import torch
import hgs_runtime as rt
from arguments import OptimizationParams
from hgs_runtime.strand_step import FusedStrandStep
from synthetic import build_workload
model, cams, _ = build_workload("north_star", device="cuda", with_targets=True, n_views=2)
opt = OptimizationParams()
model.training_setup(opt)
fused = FusedStrandStep(model, cams, opt, torch.zeros(3, device="cuda"))
for skip in (False, True, False, True):
    fused.skip_unread_blocks = skip
    for _ in range(3):
        fused.views.select(1)
        loss, _ = fused.loss()
        scratch = loss.grad_fn.saved_tensors[-2]
        fused.backward(loss)
        for p in (model._endpoints, model._width, model._opacity, model._mask, model._features_dc):
            p.grad = None
    torch.cuda.synchronize()
    n = scratch.numel()
    w = scratch.view(torch.int32)[n - 12: n - 2].cpu().numpy().astype("uint32").astype("int64")
    t0 = min(w[0], w[2])
    us = lambda x: ((int(x) - int(t0)) & 0xFFFFFFFF) / 100.0
    print(f"hint {skip}: builder {us(w[0]):.1f}..{us(w[1]):.1f} us, sums {us(w[2]):.1f}..{us(w[3]):.1f} us, "
          f"pixel wg 0 ends {us(w[4]):.1f}, wg 4000 ends {us(w[5]):.1f}, last wg ends {us(w[6]):.1f}; "
          f"builder: bitmaps at {us(w[7]):.1f}, tests at {us(w[8]):.1f}, scans at {us(w[9]):.1f}")
