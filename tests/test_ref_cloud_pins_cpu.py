"""The Stage-I GaussianModel against what the REFERENCE's own scene/gaussian_model.py (device="cpu") made of the same random
clouds in the authoring container (tests/golden/ref_cloud_pins.npz, generator tests/golden/make_ref_cloud_pins.py): getters,
schedule, densify_and_clone, prune_points, reset_opacity, update_densification_stats, compute_foreground_mask -- the same
rows in the same order, the same Adam moments and statistics, bit for bit (the same torch statements on the same CPU)."""
import os

import numpy as np
import pytest
import torch

from arguments import OptimizationParams

PINS = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_cloud_pins.npz"))
SEEDS = [int(s) for s in PINS["meta_seeds"]]
GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "mask", "rotation")


class _Info:
    def __init__(self):
        self.densification_info = {}


def _model(seed):
    from scene.gaussian_model import GaussianModel
    k = f"s{seed}_"
    m = GaussianModel(sh_degree=3, device="cpu")
    P = lambda n: torch.nn.Parameter(torch.from_numpy(PINS[k + n].copy()).requires_grad_(True))
    m._xyz, m._features_dc, m._features_rest = P("xyz"), P("f_dc"), P("f_rest")
    m._opacity, m._scaling, m._mask, m._rotation = P("opacity"), P("scaling"), P("mask"), P("rotation")
    opt = OptimizationParams()
    opt.spatial_sort = False
    m.training_setup(opt)
    assert [g["name"] for g in m.optimizer.param_groups] == list(GROUPS)          # the reference's group order (:214-246)
    for g in m.optimizer.param_groups:
        m.optimizer.state[g["params"][0]] = {"step": torch.tensor(3.0), "exp_avg": torch.from_numpy(PINS[k + g["name"] + "_exp_avg"].copy()),
                                             "exp_avg_sq": torch.from_numpy(PINS[k + g["name"] + "_exp_avg_sq"].copy())}
    m.xyz_gradient_accum, m.denom = torch.from_numpy(PINS[k + "grad_accum"].copy()), torch.from_numpy(PINS[k + "denom"].copy())
    m.max_radii2D = torch.from_numpy(PINS[k + "max_radii2D"].copy())
    return m, opt


def _same(m, key, what):
    for g in m.optimizer.param_groups:
        p = g["params"][0]
        s = m.optimizer.state.get(p, {})
        for name, got in ((g["name"], p.detach()), (g["name"] + "_exp_avg", s.get("exp_avg", torch.zeros_like(p))),
                          (g["name"] + "_exp_avg_sq", s.get("exp_avg_sq", torch.zeros_like(p)))):
            want = PINS[key + name]
            assert tuple(got.shape) == want.shape and np.array_equal(got.detach().numpy(), want), (what, name)
    for name, got in (("grad_accum", m.xyz_gradient_accum), ("denom", m.denom), ("max_radii2D", m.max_radii2D)):
        assert np.array_equal(got.numpy().reshape(-1), PINS[key + name].reshape(-1)), (what, name)


@pytest.mark.parametrize("seed", SEEDS)
def test_stage1_model_equals_the_reference_run(seed):
    k = f"s{seed}_"
    m, opt = _model(seed)
    with torch.no_grad():
        for n, v in (("scaling", m.get_scaling), ("rotation", m.get_rotation), ("opacity", m.get_opacity), ("mask", m.get_mask),
                     ("features", m.get_features)):
            assert np.array_equal(v.numpy(), PINS[k + "get_" + n]), n
        assert np.array_equal(m.compute_foreground_mask().numpy(), PINS[k + "foreground"])
    lrs = [m.update_learning_rate(int(it)) for it in PINS["meta_lr_iterations"]]
    assert np.allclose(lrs, PINS[k + "xyz_lr"], rtol=1e-12, atol=0)
    for xi, extent in enumerate(PINS["meta_clone_extents"]):
        m, opt = _model(seed)
        grads = m.xyz_gradient_accum / m.denom
        grads[grads.isnan()] = 0.0
        info = _Info()
        m.densify_and_clone(grads, opt.densify_grad_threshold, float(extent), training_info=info)
        assert info.densification_info["clone"] == int(PINS[k + f"clone{xi}_count"])
        _same(m, k + f"clone{xi}_", f"densify_and_clone extent {extent}")
    m, _ = _model(seed)
    m.prune_points(torch.from_numpy(PINS[k + "sel_prune"]))
    _same(m, k + "prune_", "prune_points")
    m, _ = _model(seed)
    m.reset_opacity()
    _same(m, k + "reset_", "reset_opacity")
    m, _ = _model(seed)
    vs = torch.zeros((PINS[k + "xyz"].shape[0], 3), requires_grad=True)
    vs.grad = torch.from_numpy(PINS[k + "vs_grad"].copy())
    m.update_densification_stats(vs, torch.from_numpy(PINS[k + "radii"]), torch.from_numpy(PINS[k + "filter"]))
    _same(m, k + "stats_", "update_densification_stats")
