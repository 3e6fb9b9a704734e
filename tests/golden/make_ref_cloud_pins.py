"""Generates tests/golden/ref_cloud_pins.npz by RUNNING the reference's Stage-I GaussianModel (scene/gaussian_model.py,
device="cpu") in the authoring container.  Only numeric inputs and outputs are stored.

Executed, unedited, on random clouds with distinct rows, Adam moments and statistics:
  * the getters get_scaling / get_rotation / get_opacity / get_mask / get_features          :118-150
  * training_setup + update_learning_rate(iteration)                                        :209-263
  * densify_and_clone(grads, threshold, extent, info)  (-> densification_postfix, cat_tensors_to_optimizer)   :471-520, 606-640
  * prune_points(mask)  (-> _prune_optimizer)                                                :434-469
  * reset_opacity()  (-> replace_tensor_to_optimizer)
  * update_densification_stats(viewspace_points, radii, filter)                              :676-684
  * compute_foreground_mask()                                                                :727-733
NOT reachable on a CPU: densify_and_split / get_orientation / get_covariance / get_segment_endpoint (utils/transform.py
allocates on device="cuda"), create_from_pcd (distCUDA2, .cuda()).  Absent third-party imports: tests/golden/_ref_harness.py.
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "ref_cloud_pins.npz")
SEEDS = list(range(6))
GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "mask", "rotation")


def random_state(seed):
    rng = np.random.default_rng(1000 + seed)
    N = int(rng.integers(40, 90))
    f32 = lambda a: np.asarray(a, dtype=np.float32)
    st = {"xyz": f32(rng.normal(size=(N, 3))), "f_dc": f32(rng.normal(size=(N, 1, 3)) * 0.3), "f_rest": f32(rng.normal(size=(N, 3, 3)) * 0.05),
          "opacity": f32(rng.normal(size=(N, 1)) * 2), "scaling": f32(rng.normal(-4, 1.2, size=(N, 3))), "mask": f32(rng.normal(size=(N, 1)) * 2),
          "rotation": f32(rng.normal(size=(N, 4)))}
    st["opacity"][rng.uniform(size=N) < 0.15] = -8.0
    for g in GROUPS:
        st[g + "_exp_avg"], st[g + "_exp_avg_sq"] = f32(rng.normal(size=st[g].shape)), f32(rng.uniform(size=st[g].shape))
    st["denom"] = f32(rng.integers(0, 3, size=(N, 1)))
    st["grad_accum"] = f32(rng.uniform(0, 6e-4, size=(N, 1)) * np.maximum(st["denom"], 1))
    st["max_radii2D"] = f32(rng.uniform(0, 30, size=N))
    st["sel_prune"] = rng.uniform(size=N) < 0.3
    st["vs_grad"] = f32(rng.normal(size=(N, 3)) * 1e-3)
    st["radii"] = f32(rng.integers(0, 40, size=N))
    st["filter"] = st["radii"] > 0
    return st


def main():
    sys.path.insert(0, HERE)
    from _ref_harness import enter_reference
    enter_reference()
    import torch
    from arguments import OptimizationParams
    from scene.gaussian_model import GaussianModel
    opt = OptimizationParams(argparse.ArgumentParser())
    out = {"meta_seeds": np.array(SEEDS)}

    class Info:
        def __init__(self):
            self.densification_info = {}

    def model(st):
        m = GaussianModel(sh_degree=3, device="cpu")
        P = lambda a: torch.nn.Parameter(torch.from_numpy(a.copy()).requires_grad_(True))
        m._xyz, m._features_dc, m._features_rest = P(st["xyz"]), P(st["f_dc"]), P(st["f_rest"])
        m._opacity, m._scaling, m._mask, m._rotation = P(st["opacity"]), P(st["scaling"]), P(st["mask"]), P(st["rotation"])
        m.training_setup(opt)
        for g in m.optimizer.param_groups:
            m.optimizer.state[g["params"][0]] = {"step": torch.tensor(3.0), "exp_avg": torch.from_numpy(st[g["name"] + "_exp_avg"].copy()),
                                                 "exp_avg_sq": torch.from_numpy(st[g["name"] + "_exp_avg_sq"].copy())}
        m.xyz_gradient_accum, m.denom = torch.from_numpy(st["grad_accum"].copy()), torch.from_numpy(st["denom"].copy())
        m.max_radii2D = torch.from_numpy(st["max_radii2D"].copy())
        return m

    def dump(key, m):
        for g in m.optimizer.param_groups:
            p = g["params"][0]
            s = m.optimizer.state.get(p, {})
            out[key + g["name"]] = p.detach().numpy().copy()
            out[key + g["name"] + "_exp_avg"] = s.get("exp_avg", torch.zeros_like(p)).detach().numpy().copy()
            out[key + g["name"] + "_exp_avg_sq"] = s.get("exp_avg_sq", torch.zeros_like(p)).detach().numpy().copy()
        out[key + "grad_accum"], out[key + "denom"], out[key + "max_radii2D"] = m.xyz_gradient_accum.numpy().copy(), m.denom.numpy().copy(), m.max_radii2D.numpy().copy()

    for seed in SEEDS:
        st = random_state(seed)
        k = f"s{seed}_"
        for n, v in st.items():
            out[k + n] = v
        m = model(st)
        with torch.no_grad():
            for n, v in (("scaling", m.get_scaling), ("rotation", m.get_rotation), ("opacity", m.get_opacity), ("mask", m.get_mask), ("features", m.get_features)):
                out[k + "get_" + n] = v.numpy().copy()
            out[k + "foreground"] = m.compute_foreground_mask().numpy().copy()
            its = [1, 100, 1000, 7000, 30000]
            out["meta_lr_iterations"] = np.array(its)
            out[k + "xyz_lr"] = np.array([m.update_learning_rate(it) for it in its], dtype=np.float64)
            # clone
            for xi, extent in enumerate((1e-3, 1.0, 50.0)):
                m = model(st)
                grads = m.xyz_gradient_accum / m.denom
                grads[grads.isnan()] = 0.0
                info = Info()
                m.densify_and_clone(grads, opt.densify_grad_threshold, extent, training_info=info)
                dump(k + f"clone{xi}_", m)
                out[k + f"clone{xi}_count"] = np.int64(info.densification_info["clone"])
            out["meta_clone_extents"] = np.array([1e-3, 1.0, 50.0])
            m = model(st)
            m.prune_points(torch.from_numpy(st["sel_prune"]))
            dump(k + "prune_", m)
            m = model(st)
            m.reset_opacity()
            dump(k + "reset_", m)
        m = model(st)
        vs = torch.zeros((st["xyz"].shape[0], 3), requires_grad=True)
        vs.grad = torch.from_numpy(st["vs_grad"].copy())
        with torch.no_grad():
            m.update_densification_stats(vs, torch.from_numpy(st["radii"]), torch.from_numpy(st["filter"]))
        dump(k + "stats_", m)
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT} ({os.path.getsize(OUT) / 1024:.0f} KB, {len(out)} arrays)")


if __name__ == "__main__":
    main()
