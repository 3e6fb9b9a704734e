"""Generates tests/golden/ref_fullsize_pins.npz: the reference's loss code (loss/losses.py) RUN on the CPU of the authoring
container AT THE NORTH-STAR SIZE (1920 x 1080), with inputs that are not stored but re-created from a seed (numpy PCG64 streams
are the same everywhere): the fixture holds the scalars and a few thousand SAMPLED gradient elements (index, value), plus the
float64 sum of |gradient| as a checksum.

Executed, unedited: l1_loss, ssim (:16-17, 43-84) on a dense random pair and on a hair-like pair (black outside a region);
loss_function (:319-355) with mask + orientation targets, `render` bound to prescribed images as in make_ref_loss_pins.py
(same three stand-ins, stated there); angle_smoothness_loss on a strand model of the north-star size (100 k segments) whose
arrays arrive through a temporary file written by this repository's synthetic workload (stage 1: `--stage inputs`).
"""
import argparse
import os
import subprocess
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
OUT = os.path.join(HERE, "ref_fullsize_pins.npz")
TMP = os.path.join(HERE, "_fullsize_inputs.npz")
H, W = 1080, 1920
NS = 4096


def images(seed, hair):
    """(image, target) [3, H, W] float32 from one seed -- shared with tests/test_ref_fullsize_pins.py."""
    rng = np.random.default_rng(seed)
    a = rng.random((3, H, W), dtype=np.float32)
    b = np.clip(a + rng.standard_normal((3, H, W), dtype=np.float32) * np.float32(0.1), 0, 1).astype(np.float32)
    if hair:
        yy, xx = np.mgrid[0:H, 0:W]
        m = (((xx - 960) / 520.0) ** 2 + ((yy - 500) / 420.0) ** 2 < 1.0)
        a, b = a * m, b * np.roll(m, 3, axis=1)
    return np.ascontiguousarray(a, dtype=np.float32), np.ascontiguousarray(b, dtype=np.float32)


def head_inputs(seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    m = (((xx - 960) / 520.0) ** 2 + ((yy - 500) / 420.0) ** 2 < 1.0)
    omap = (rng.standard_normal((3, H, W), dtype=np.float32) * m).astype(np.float32)
    mlog = (rng.standard_normal((3, H, W), dtype=np.float32) * np.float32(2.0)).astype(np.float32)
    ori = (rng.random((H, W), dtype=np.float32) * np.float32(np.pi)).astype(np.float32)
    conf = rng.random((H, W), dtype=np.float32)
    mask = np.roll(m, 5, axis=0) & (rng.random((H, W), dtype=np.float32) > 0.1)
    R = np.linalg.qr(rng.standard_normal((3, 3)))[0]
    wvt = np.eye(4, dtype=np.float32)
    wvt[:3, :3] = R
    wvt[3, :3] = rng.standard_normal(3)
    return omap, mlog, ori, conf, mask, wvt


def sample(rng, g):
    idx = rng.choice(g.size, size=NS, replace=False)
    return idx.astype(np.int64), g.reshape(-1)[idx].copy()


def stage_inputs():
    sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
    from synthetic import build_workload
    model, _, _ = build_workload("north_star", device="cpu", seed=0, with_targets=False, n_views=1)
    np.savez(TMP, endpoints=model._endpoints.detach().numpy(), pairs=model.endpoint_pairs.numpy(), opacity=model._opacity.detach().numpy(),
             mask=model._mask.detach().numpy(), width=model._width.detach().numpy(), f_dc=model._features_dc.detach().numpy(),
             f_rest=model._features_rest.detach().numpy(), ref_strand_root=np.asarray(model.ref_strand_root, dtype=np.float64),
             root_idx=model.strand_root_endpoint_idx.numpy())
    print("inputs written")


def stage_reference():
    sys.path.insert(0, ROOT)
    from oracle import build_ref
    build_ref.build()
    sys.modules["c_utils"] = build_ref.load()
    sys.path.remove(ROOT)
    sys.path.insert(0, HERE)
    from _ref_harness import enter_reference
    enter_reference()
    import torch
    real_tensor = torch.tensor

    def cpu_tensor(*a, **k):
        if str(k.get("device", "")).startswith("cuda"):
            k["device"] = "cpu"
        return real_tensor(*a, **k)
    torch.tensor = cpu_tensor
    try:
        from loss import losses as RL
    finally:
        torch.tensor = real_tensor
    from arguments import OptimizationParams
    from scene.hair_gaussian_model import HairGaussianModel
    opt = OptimizationParams(argparse.ArgumentParser())
    out = {"meta_hw": np.array([H, W])}
    srng = np.random.default_rng(99)
    for ci, (seed, hair) in enumerate(((11, False), (12, True))):
        a, b = images(seed, hair)
        x, y = torch.from_numpy(a).requires_grad_(True), torch.from_numpy(b)
        s = RL.ssim(x, y)
        gs, = torch.autograd.grad(s, x)
        l = RL.l1_loss(x, y)
        gl, = torch.autograd.grad(l, x)
        k = f"ssim{ci}_"
        out[k + "seed"], out[k + "hair"] = np.int64(seed), np.bool_(hair)
        out[k + "ssim"], out[k + "l1"] = np.float64(s.item()), np.float64(l.item())
        out[k + "d_ssim_idx"], out[k + "d_ssim_val"] = sample(srng, gs.numpy())
        out[k + "d_l1_idx"], out[k + "d_l1_val"] = sample(srng, gl.numpy())
        out[k + "d_ssim_abs_sum"], out[k + "d_ssim_abs_max"] = np.float64(gs.double().abs().sum().item()), np.float64(gs.abs().max().item())
    # the strand model of the north-star size
    inp = np.load(TMP)
    m = HairGaussianModel(sh_degree=3, device="cpu")
    m.ref_strand_root = inp["ref_strand_root"]
    m.strand_root_endpoint_idx = torch.from_numpy(inp["root_idx"])
    m.endpoint_pairs = torch.from_numpy(inp["pairs"])
    rngk = np.random.default_rng(5)
    ep = (inp["endpoints"] + rngk.normal(0, 2e-4, inp["endpoints"].shape)).astype(np.float32)      # kinked: the smoothness term is not empty
    P = lambda a: torch.nn.Parameter(torch.from_numpy(a.copy()).requires_grad_(True))
    m._endpoints, m._features_dc, m._features_rest = P(ep), P(inp["f_dc"]), P(inp["f_rest"])
    m._opacity, m._mask, m._width = P(inp["opacity"]), P(inp["mask"]), P(inp["width"])
    m.training_setup(opt)
    m.compute_strands_info()
    out["model_segments"] = np.int64(m.endpoint_pairs.shape[0])
    for th in (30.0, 3.0):
        v = RL.angle_smoothness_loss(m, threshold=th)
        g, = torch.autograd.grad(v, m._endpoints)
        k = f"smooth{int(th)}_"
        out[k + "value"] = np.float64(v.item())
        out[k + "idx"], out[k + "val"] = sample(srng, g.numpy())
        out[k + "abs_sum"], out[k + "abs_max"] = np.float64(g.double().abs().sum().item()), np.float64(g.abs().max().item())
    with torch.no_grad():
        for n, v in (("scaling", m.get_scaling), ("xyz", m.get_xyz), ("orientation", m.get_orientation)):
            out["get_" + n + "_idx"], out["get_" + n + "_val"] = sample(srng, v.numpy())
    # the whole loss function at 1080p
    a, b = images(13, True)
    omap, mlog, ori, conf, mask, wvt = head_inputs(14)
    cam = types.SimpleNamespace(original_image=torch.from_numpy(b), world_view_transform=torch.from_numpy(wvt), orientation_field=torch.from_numpy(ori),
                                orientation_confidence=torch.from_numpy(conf), mask=torch.from_numpy(mask), float_mask=torch.from_numpy(mask).float())
    x, xo, xm = (torch.from_numpy(t).requires_grad_(True) for t in (a, omap, mlog))
    calls = []

    def render(camera, pc, bg, scaling_modifier=1.0, override_color=None, debug=False):
        calls.append(1)
        return {"render": xm if len(calls) == 1 else xo}
    RL.render = render
    loss, d = RL.loss_function(m, x, cam, opt)
    gi, go, gm = torch.autograd.grad(loss, [x, xo, xm])
    out["head_total"] = np.float64(loss.item())
    for n in ("l1", "dssim", "mask", "orientation", "smooth"):
        out["head_term_" + n] = np.float64(float(d[n]))
    empty = ~omap.any(axis=0)
    go = go.numpy().copy()
    go[:, empty] = 0.0          # (pixels of the mask the render left at 0: ~1e10, compared separately at small size; sampled values exclude them)
    for n, g in (("d_image", gi.numpy()), ("d_omap", go), ("d_mask", gm.numpy()[0])):
        out["head_" + n + "_idx"], out["head_" + n + "_val"] = sample(srng, g)
        out["head_" + n + "_abs_max"] = np.float64(np.abs(g).max())
    np.savez_compressed(OUT, **out)
    os.remove(TMP)
    print(f"wrote {OUT} ({os.path.getsize(OUT) / 1024:.0f} KB, {len(out)} arrays)")


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
