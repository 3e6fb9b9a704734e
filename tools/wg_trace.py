"""Workgroup timeline of the blend kernels on a workload (development aid, hgs_debug_set_wg_trace): residency over time,
phase durations of the split lists, the latest finishers."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import numpy as np, torch
import hgs_runtime as rt
from gaussian_renderer import render_multi
from synthetic import PIPELINE_STATES, build_pipeline_state, build_workload
wl = sys.argv[1] if len(sys.argv) > 1 else "north_star"
n_train = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # trace the state this many iterations of the FULL loop leave
if wl in PIPELINE_STATES:     # a state of the three-stage workflow (its own Stage-I loop runs here; `n_train` is ignored)
    model, cams, extent, _info = build_pipeline_state(wl, device="cuda", n_views=8)
    n_train = 0
else:
    model, cams, extent = build_workload(wl, device="cuda", with_targets=n_train > 0, n_views=8 if n_train else 4)
bg = torch.zeros(3, device="cuda")
if n_train:
    from arguments import OptimizationParams
    from train import training
    from utils.general import safe_state
    safe_state(True)
    opt = OptimizationParams()
    model.training_setup(opt)
    training(model, cams, opt, iterations=n_train, extent=extent, seed=1)
    print("trained", n_train, "iterations:", model.get_xyz.shape[0], "segments")
cam = cams[0]
H, W = cam.image_height, cam.image_width
T = ((W + 15) // 16) * ((H + 15) // 16)
G = 2 * T + 1024
w3 = torch.randn(3, H, W, device="cuda"); w1 = torch.randn(H, W, device="cuda"); wo = torch.randn(3, H, W, device="cuda")
tf = torch.zeros(8 * G, dtype=torch.int64, device="cuda"); tb = torch.zeros(8 * G, dtype=torch.int64, device="cuda")
def run():
    extra = torch.cat((model.get_mask, model.get_orientation), dim=1)
    pkg = render_multi(cam, model, bg, extra, splits=(1, 3), black_background=True)   # bg is the zeros made above
    loss = (pkg["render"] * w3).sum() + (pkg["extra"][0] * w1).sum() + (pkg["extra"][1] * wo).sum()
    loss.backward()
for _ in range(3): run()
torch.cuda.synchronize()
rt.check(rt.lib().hgs_debug_set_wg_trace(tf.data_ptr(), tb.data_ptr()))
run(); torch.cuda.synchronize()
rt.check(rt.lib().hgs_debug_set_wg_trace(None, None))
for name, t in (("fwd", tf), ("bwd", tb)):
    a = t.cpu().numpy().reshape(G, 8)
    ok = a[:, 7] > 0
    a = a[ok]
    item = a[:, 6]
    tile, seg, nseg, length = item & 0xFFFFFF, (item >> 24) & 0xFF, (item >> 32) & 0xFF, item >> 40
    us = a.astype(np.float64) * 0.01
    t0, t1 = us[:, 0].min(), us[:, 7].max()
    dur = us[:, 7] - us[:, 0]
    split = nseg > 1
    print(f"{name}: kernel span {t1 - t0:.1f} us, workgroups {len(a)} ({split.sum()} segments of {len(np.unique(tile[split]))} split tiles), "
          f"entries {length.sum()} ({length[split].sum()} in split lists), sum of WG time {dur.sum():.0f} us")
    edges = np.linspace(t0, t1, 24)
    print("   resident WGs over time:", [int(((us[:, 0] <= e) & (us[:, 7] > e)).sum()) for e in edges])
    print("   start times percentiles 50/90/99/100:", np.round(np.percentile(us[:, 0] - t0, [50, 90, 99, 100]), 1).tolist())
    for label, sel in (("unsplit", ~split), ("split", split)):
        if sel.any():
            print(f"   {label}: n {sel.sum()}, len mean {length[sel].mean():.0f} max {length[sel].max()}, dur p50/p90/max "
                  f"{np.percentile(dur[sel], 50):.1f}/{np.percentile(dur[sel], 90):.1f}/{dur[sel].max():.1f} us, "
                  f"ns per entry {1e3 * dur[sel].sum() / max(1, length[sel].sum()):.0f}")
    if name == "fwd" and split.any():
        s = us[split]
        ph = np.stack([s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2], s[:, 4] - s[:, 3], s[:, 7] - s[:, 4]], 1)
        print("   split fwd phases (phase1, wait, phase2, publish+ticket, finalize) mean us:", np.round(ph.mean(0), 2).tolist(),
              " p90:", np.round(np.percentile(ph, 90, axis=0), 2).tolist(), " max:", np.round(ph.max(0), 2).tolist())
        print("   split fwd absolute times of start / mark1 / mark2 / mark3 / end, percentiles 10/50/90/100:",
              [np.round(np.percentile(s[:, k] - t0, [10, 50, 90, 100]), 1).tolist() for k in (0, 1, 2, 3, 7)])
        by_seg = [(int(k), round(float((s[seg[split] == k, 1] - s[seg[split] == k, 0]).mean()), 1), round(float((s[seg[split] == k, 2] - s[seg[split] == k, 1]).mean()), 1),
                   round(float((s[seg[split] == k, 3] - s[seg[split] == k, 2]).mean()), 1)) for k in range(0, int(seg[split].max()) + 1, 2)]
        print("   by segment index (seg, passA, wait, passB):", by_seg)
        later = seg[split] > 0
        if later.any() and (s[later, 5] > 0).all():      # mark 5: the predecessors' products are published (flag seen)
            sl = s[later]
            print("   segments > 0: pass A end -> flag seen -> T_in known, mean us:", round(float((sl[:, 5] - sl[:, 1]).mean()), 2),
                  round(float((sl[:, 2] - sl[:, 5]).mean()), 2), " p90:", round(float(np.percentile(sl[:, 5] - sl[:, 1], 90)), 2),
                  round(float(np.percentile(sl[:, 2] - sl[:, 5], 90)), 2))
            first = s[seg[split] == 0]
            print("   segment 0: pass A mean / p90 / max us:", np.round([(first[:, 1] - first[:, 0]).mean(), np.percentile(first[:, 1] - first[:, 0], 90), (first[:, 1] - first[:, 0]).max()], 1).tolist(),
                  " segments 1..: ", np.round([(sl[:, 1] - sl[:, 0]).mean(), np.percentile(sl[:, 1] - sl[:, 0], 90), (sl[:, 1] - sl[:, 0]).max()], 1).tolist())
    if name == "fwd":
        for i in np.argsort(-us[:, 7])[:6]:
            print("      late:", int(tile[i]), f"{int(seg[i])}/{int(nseg[i])}", "marks (us from kernel start):", np.round(us[i, [0, 1, 2, 3, 4, 7]] - t0, 1).tolist())
    if name == "bwd":
        pairs, empty, lanes = int(a[:, 1].sum()), int(a[:, 2].sum()), int(a[:, 3].sum())
        print(f"   (entry, wavefront) pairs evaluated {pairs} = {pairs / max(1, length.sum()):.2f} per entry; without any blending lane {empty} "
              f"({100.0 * empty / max(1, pairs):.0f} %); blending lanes per remaining pair {lanes / max(1, pairs - empty):.1f} of 64")
    late = np.argsort(-us[:, 7])[:8]
    print("   latest finishers (tile, seg/nseg, len, start, dur):",
          [(int(tile[i]), f"{int(seg[i])}/{int(nseg[i])}", int(length[i]), round(float(us[i, 0] - t0), 1), round(float(dur[i]), 1)) for i in late])
