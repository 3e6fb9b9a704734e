"""Generates tests/golden/ref_topology_pins.npz by RUNNING the reference's own strand-topology code on the CPU of the
authoring container (never on the GPU box): scene/hair_gaussian_model.py is written against `self.device`, and with
device="cpu" its clone / split / merge_collapsed / prune / merging / compute_strands_info operators are plain torch +
numpy + scipy.  Only numeric inputs and outputs are stored.

Two stages, two processes (both packages are called `scene` / `utils` / `arguments`, so they never share an interpreter):
  --stage inputs     (this repository's package) random strand models -- tests/test_topology_restatement_cpu.py's
                     `_random_model`, the very states the scalar restatement is checked on -- dumped as arrays;
  --stage reference  (/root/reference only on sys.path) each state is loaded into the REFERENCE's HairGaussianModel and the
                     reference's own methods are run on it, unedited:
                       densification(extent, max_screen_size, info)           :788-817   (seeds x three extents)
                       clone_strategy / split_strategy / merge_collapsed_segments / prune_strategy (both modes),
                       clean_gaussians (both modes), update_densification_stats, one at a time   :828-1077, 1401-1408, 1502-1516
                       compute_strands_info()                                 :1410-1496
                       compute_endpoint_pair_to_merge()                       :1205-1362
                       merging(info)                                          :1079-1096
                       reset_opacity()                                        :1364-1371
                       the Stage-II merging loop of merge.py:113-177 (candidates -> merge -> strands_info to a fixed point)
The reference module's import line pulls in third-party packages this image lacks (pytorch3d, plyfile, simple_knn, cv2,
pyrr, pyvista, pyvistaqt, dreifus, wandb, tensorboard): they are satisfied by EMPTY placeholder modules whose every attribute
raises when called (tests/golden/_ref_harness.py).  None of the methods run here calls into them (a call would abort the
generation); nothing of the reference is edited, wrapped or re-implemented.
"""
import argparse
import os
import subprocess
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
OUT = os.path.join(HERE, "ref_topology_pins.npz")
TMP_IN = os.path.join(HERE, "_topology_inputs.npz")

DENS_SEEDS = list(range(12))
DENS_EXTENTS = [1e-3, 0.05, 50.0]
OP_SEEDS = [100 + s for s in range(8)]
MERGE_CASES = list(range(12))        # cut-up curves: seed = case // 2, bidirectional_merge = case % 2
GROUPS = ("endpoints", "f_dc", "f_rest", "opacity", "mask", "width")


# ---- stage 1: input states from this repository's random models ------------------------------------------------------------
def stage_inputs():
    sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd"), os.path.join(ROOT, "tests")]
    import torch
    import test_topology_restatement_cpu as T
    out = {}
    models = [(f"s{seed}", T._random_model(seed)) for seed in sorted(set(DENS_SEEDS + OP_SEEDS))]
    models += [(f"c{case}", T._cut_model(case // 2, bool(case % 2))) for case in MERGE_CASES]
    for tag, m in models:
        k = f"{tag}_"
        out[k + "pairs"] = m.endpoint_pairs.numpy().astype(np.int64)
        for g in m.optimizer.param_groups:
            p = g["params"][0]
            st = m.optimizer.state.get(p, {})
            out[k + g["name"]] = p.detach().numpy()
            out[k + g["name"] + "_exp_avg"] = st["exp_avg"].numpy() if "exp_avg" in st else np.zeros(tuple(p.shape), np.float32)
            out[k + g["name"] + "_exp_avg_sq"] = st["exp_avg_sq"].numpy() if "exp_avg_sq" in st else np.zeros(tuple(p.shape), np.float32)
            out[k + "has_state"] = np.bool_("exp_avg" in st)
        out[k + "grad_accum"] = m.xyz_gradient_accum.numpy()
        out[k + "denom"] = m.denom.numpy()
        out[k + "max_radii2D"] = m.max_radii2D.numpy()
        out[k + "ref_strand_root"] = np.asarray(m.ref_strand_root, dtype=np.float64)
        out[k + "root_idx"] = m.strand_root_endpoint_idx.numpy().astype(np.int64)
        out[k + "max_segment_length"] = np.float32(float(m.max_segment_length))
        out[k + "active_sh_degree"] = np.int64(m.active_sh_degree)
    np.savez_compressed(TMP_IN, **out)
    print("inputs:", len(out), "arrays")


# ---- stage 2: the reference's own methods ------------------------------------------------------------------------------------
class _Info:
    def __init__(self):
        self.densification_info = {}


def _ref_model(inp, seed, torch, HairGaussianModel, opt):
    k = f"{seed}_" if isinstance(seed, str) else f"s{seed}_"
    m = HairGaussianModel(sh_degree=3, device="cpu")
    m.active_sh_degree = int(inp[k + "active_sh_degree"])
    m.ref_strand_root = inp[k + "ref_strand_root"]
    m.strand_root_endpoint_idx = torch.from_numpy(inp[k + "root_idx"])
    m.endpoint_pairs = torch.from_numpy(inp[k + "pairs"])
    P = lambda a: torch.nn.Parameter(torch.from_numpy(a.copy()).requires_grad_(True))
    m._endpoints, m._features_dc, m._features_rest = P(inp[k + "endpoints"]), P(inp[k + "f_dc"]), P(inp[k + "f_rest"])
    m._opacity, m._mask, m._width = P(inp[k + "opacity"]), P(inp[k + "mask"]), P(inp[k + "width"])
    m.training_setup(opt)
    for g in m.optimizer.param_groups:
        p = g["params"][0]
        if bool(inp[k + "has_state"]):
            m.optimizer.state[p] = {"step": torch.tensor(3.0), "exp_avg": torch.from_numpy(inp[k + g["name"] + "_exp_avg"].copy()),
                                    "exp_avg_sq": torch.from_numpy(inp[k + g["name"] + "_exp_avg_sq"].copy())}
    m.xyz_gradient_accum = torch.from_numpy(inp[k + "grad_accum"].copy())
    m.denom = torch.from_numpy(inp[k + "denom"].copy())
    m.max_radii2D = torch.from_numpy(inp[k + "max_radii2D"].copy())
    return m


def _dump(out, key, m, torch):
    out[key + "pairs"] = m.endpoint_pairs.numpy().astype(np.int64)
    for g in m.optimizer.param_groups:
        p = g["params"][0]
        st = m.optimizer.state.get(p, {})
        out[key + g["name"]] = p.detach().numpy().copy()
        out[key + g["name"] + "_exp_avg"] = st.get("exp_avg", torch.zeros_like(p)).detach().numpy().copy()
        out[key + g["name"] + "_exp_avg_sq"] = st.get("exp_avg_sq", torch.zeros_like(p)).detach().numpy().copy()
    out[key + "grad_accum"] = m.xyz_gradient_accum.numpy().copy()
    out[key + "denom"] = m.denom.numpy().copy()
    out[key + "max_radii2D"] = m.max_radii2D.numpy().copy()


def _dump_strands(out, key, si):
    # ragged lists as one flat array + offsets
    ls = [np.asarray(s, dtype=np.int64).reshape(-1, 2) for s in si.list_strands]
    lid = [np.asarray(s, dtype=np.int64).reshape(-1) for s in si.list_strands_segments_id]
    out[key + "strand_offsets"] = np.cumsum([0] + [len(s) for s in ls]).astype(np.int64)
    out[key + "strand_points"] = np.concatenate(ls, 0) if ls else np.zeros((0, 2), np.int64)
    out[key + "strand_segment_ids"] = np.concatenate(lid, 0) if lid else np.zeros((0,), np.int64)
    out[key + "id_to_strand_id"] = np.asarray(si.id_to_strand_id, dtype=np.int64)
    out[key + "complementary"] = np.asarray(si.strand_endpoint_id_to_complementary, dtype=np.int64)


def stage_reference():
    sys.path.insert(0, HERE)
    from _ref_harness import enter_reference
    enter_reference()
    import torch
    from arguments import OptimizationParams
    from scene.hair_gaussian_model import HairGaussianModel
    opt = OptimizationParams(argparse.ArgumentParser())
    inp = np.load(TMP_IN)
    out = {k: inp[k] for k in inp.files}
    out["meta_dens_seeds"], out["meta_dens_extents"] = np.array(DENS_SEEDS), np.array(DENS_EXTENTS)
    out["meta_op_seeds"], out["meta_merge_cases"] = np.array(OP_SEEDS), np.array(MERGE_CASES)
    n_runs = 0

    def info_arr(info, names):
        return np.array([int(info.densification_info.get(n, -1)) for n in names], dtype=np.int64)

    DENS_INFO = ("clone", "split", "merge_collapsed", "prune_collapsed", "prune_low_opacity", "prune_big_ws", "prune_avoided", "prune_total")
    out["meta_dens_info_names"] = np.array(DENS_INFO)
    for seed in DENS_SEEDS:
        for xi, extent in enumerate(DENS_EXTENTS):
            m = _ref_model(inp, seed, torch, HairGaussianModel, opt)
            # the reference asserts nothing about max_segment_length between the two packages: recorded and compared by the test
            out[f"s{seed}_ref_max_segment_length"] = np.float32(float(m.max_segment_length))
            info = _Info()
            with torch.no_grad():
                m.densification(extent, None, info)
            key = f"dens_s{seed}_x{xi}_"
            _dump(out, key, m, torch)
            out[key + "info"] = info_arr(info, DENS_INFO)
            _dump_strands(out, key, m.strands_info)
            n_runs += 1
    for seed in OP_SEEDS:
        extent = 0.02
        for op in ("clone", "split", "merge_collapsed", "prune", "prune_free", "clean", "clean_all", "stats"):
            m = _ref_model(inp, seed, torch, HairGaussianModel, opt)
            info = _Info()
            with torch.no_grad():
                grads = m.xyz_gradient_accum / m.denom
                grads[grads.isnan()] = 0.0
                if op == "clone":
                    m.clone_strategy(grads, extent, info)
                elif op == "split":
                    m.split_strategy(grads, extent, info)
                elif op == "merge_collapsed":
                    m.merge_collapsed_segments(info)
                elif op == "prune":
                    m.prune_strategy(extent, 20, info, avoid_connected=True)
                elif op == "prune_free":
                    m.prune_strategy(extent, 20, info, avoid_connected=False)
                elif op == "clean":
                    m.clean_gaussians()                                   # :1502-1516 (render.py:56)
                elif op == "clean_all":
                    m.clean_gaussians(avoid_connected=False)
                else:                                                     # update_densification_stats :1401-1408
                    rng = np.random.default_rng(seed)
                    Pn = m.endpoint_pairs.shape[0]
                    vs = torch.zeros((Pn, 3), requires_grad=True)
                    vs.grad = torch.from_numpy((rng.normal(size=(Pn, 3)) * 1e-3).astype(np.float32))
                    radii = torch.from_numpy(rng.integers(0, 40, size=Pn).astype(np.float32))
                    out[f"op_stats_s{seed}_in_vs_grad"], out[f"op_stats_s{seed}_in_radii"] = vs.grad.numpy().copy(), radii.numpy().copy()
                    m.update_densification_stats(vs, radii, radii > 0)
            key = f"op_{op}_s{seed}_"
            _dump(out, key, m, torch)
            out[key + "info"] = info_arr(info, DENS_INFO)
            n_runs += 1
    for case in MERGE_CASES:
        seed = f"c{case}"
        opt.bidirectional_merge = bool(case % 2)
        m = _ref_model(inp, seed, torch, HairGaussianModel, opt)
        out[f"{seed}_ref_max_segment_length"] = np.float32(float(m.max_segment_length))
        with torch.no_grad():
            m.compute_strands_info()
            key = f"merge_{seed}_"
            _dump_strands(out, key + "before_", m.strands_info)
            pairs = m.compute_endpoint_pair_to_merge()
            out[key + "pairs_to_merge"] = pairs.numpy().astype(np.int64).reshape(-1, 2)
            out[key + "merge_dist_th"], out[key + "merge_angle_th"] = np.float64(m.merge_dist_th), np.float64(m.merge_angle_th)
            info = _Info()
            m.merging(info)
            _dump(out, key + "after_", m, torch)
            _dump_strands(out, key + "after_", m.strands_info)
            out[key + "info_merge"] = np.int64(info.densification_info["merge"])
            # merge.py:113-177: the Stage-II loop -- candidates, merge, strands_info -- until no candidate is left
            ml = _ref_model(inp, seed, torch, HairGaussianModel, opt)
            ml.compute_strands_info()
            per_round = []
            for _ in range(50):
                pr = ml.compute_endpoint_pair_to_merge()
                per_round.append(int(pr.shape[0]))
                if pr.shape[0] == 0:
                    break
                ml.merge_endpoint_pairs(pr)
                ml.compute_strands_info()
            out[key + "loop_pairs_per_round"] = np.array(per_round, dtype=np.int64)
            _dump(out, key + "loop_", ml, torch)
            _dump_strands(out, key + "loop_", ml.strands_info)
            m.reset_opacity()
            out[key + "reset_opacity"] = m._opacity.detach().numpy().copy()
            st = m.optimizer.state.get(m._opacity, {})
            out[key + "reset_opacity_exp_avg"] = st["exp_avg"].numpy().copy() if "exp_avg" in st else np.zeros(tuple(m._opacity.shape), np.float32)
            out[key + "reset_opacity_exp_avg_sq"] = st["exp_avg_sq"].numpy().copy() if "exp_avg_sq" in st else np.zeros(tuple(m._opacity.shape), np.float32)
        n_runs += 1
    opt.bidirectional_merge = False
    np.savez_compressed(OUT, **out)
    os.remove(TMP_IN)
    print(f"reference runs: {n_runs}; wrote {OUT} ({os.path.getsize(OUT) / 1024:.0f} KB, {len(out)} arrays)")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--stage", choices=["all", "inputs", "reference"], default="all")
    a = ap.parse_args()
    if a.stage == "inputs":
        stage_inputs()
    elif a.stage == "reference":
        stage_reference()
    else:
        for st in ("inputs", "reference"):
            subprocess.run([sys.executable, os.path.abspath(__file__), "--stage", st], check=True)
