import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import numpy as np, torch
from synthetic import build_pipeline_state
from arguments import OptimizationParams
from train import training
model, cams, extent, _ = build_pipeline_state("stage3_merged", device="cuda", seed=0)
opt = OptimizationParams(); opt.iterations = 5000; opt._finalise(); model.training_setup(opt)
training(model, cams, opt, iterations=950, extent=extent)
torch.cuda.synchronize()
self = model
T = {}
def tick(name, t0):
    torch.cuda.synchronize(); T[name] = T.get(name, 0) + time.perf_counter() - t0; return time.perf_counter()
for rep in range(3):
    T.clear()
    t = time.perf_counter()
    dir_th = np.cos(np.deg2rad(self.merge_angle_th))
    deg = self._endpoint_degree_table()
    ends = torch.nonzero(deg == 1).squeeze(1)
    is_fg = torch.zeros(deg.shape[0], dtype=torch.bool, device=self.device)
    is_fg[self.endpoint_pairs[self.compute_foreground_mask()].flatten()] = True
    ends = ends[is_fg[ends]]
    t = tick("ends", t)
    comp, _ = self.get_complementary_endpoint_idx(ends)
    pos_t = self._endpoints[ends].detach()
    dirs_t = self._endpoints[comp].detach() - pos_t
    dirs_t = dirs_t / torch.norm(dirs_t, dim=1, keepdim=True)
    t = tick("dirs", t)
    pos, dirs, ends_np = pos_t.cpu().numpy(), dirs_t.cpu().numpy(), ends.cpu().numpy()
    partner = self.strands_info.strand_endpoint_id_to_complementary
    t = tick("to host", t)
    a, b = self._radius_pairs_gpu(pos_t, dirs_t, float(self.merge_dist_th), float(dir_th), bool(self.training_args.bidirectional_merge))
    t = tick("radius pairs", t)
    ok = partner[ends_np[a]] != ends_np[b]
    a, b = a[ok], b[ok]
    dist = np.linalg.norm(pos[a] - pos[b], axis=1)
    t = tick("filter+dist", t)
    order = np.lexsort((b, a, dist))
    a, b, dist = a[order], b[order], dist[order]
    order = np.argsort(dist, kind="stable")
    cand = np.stack([ends_np[a[order]], ends_np[b[order]]], 1)
    t = tick("sorts", t)
    flat = cand.reshape(-1)
    _, first_at = np.unique(flat, return_index=True)
    is_first = np.zeros(flat.shape[0], bool); is_first[first_at] = True
    cand = cand[is_first.reshape(-1, 2).all(axis=1)]
    t = tick("unique", t)
    blocked, out = set(), []
    partner_of = partner[cand]
    for (p, q), (pp, pq) in zip(cand.tolist(), partner_of.tolist()):
        if p in blocked or q in blocked:
            continue
        blocked.add(pp); blocked.add(pq); out.append((p, q))
    t = tick("greedy loop", t)
    r = torch.as_tensor(np.asarray(out, np.int64), device=self.device)
    t = tick("to device", t)
    print("ends", ends.numel(), "candidates", a.shape[0], "after stage 1", cand.shape[0], "merged", len(out), {k: round(1e3 * v, 2) for k, v in T.items()}, "total ms", round(1e3 * sum(T.values()), 2))
