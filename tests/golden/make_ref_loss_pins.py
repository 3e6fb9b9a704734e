"""Generates tests/golden/ref_loss_pins.npz by RUNNING the reference's own loss code (loss/losses.py) and the reference
model's own getters on the CPU of the authoring container.  Only numeric inputs and outputs are stored.

Executed, unedited (values and, through autograd, gradients):
  * l1_loss, ssim                                   loss/losses.py:16-17, 43-84     (F.conv2d on the CPU)
  * angle_smoothness_loss(gaussians, threshold)     :175-221    on reference HairGaussianModels (device="cpu") built from the
                                                                states of ref_topology_pins.npz; c_utils = the reference's
                                                                own .pyx, built by oracle/build_ref.py outside the repository
  * orientation_loss_rast, mask_loss_rast           :224-316    see (2) below
  * loss_function                                   :319-355    the five terms and the weighted total
  * HairGaussianModel.get_scaling / get_xyz / get_orientation / get_opacity / get_mask   scene/hair_gaussian_model.py:135-201
    and GaussianModel.set_pval                      scene/gaussian_model.py:696-704
    (get_rotation needs pytorch3d.transforms.matrix_to_quaternion: absent, NOT pinned)
  * update_learning_rate(iteration)                 scene/hair_gaussian_model.py:285-293   (the three schedules)

What stands in for things this image lacks (all of it stated here, none of it inside the executed statements):
  (1) tests/golden/_ref_harness.py: empty placeholder modules for absent third-party imports;
  (2) the reference's `render` is the CUDA rasterizer: inside loss/losses.py the NAME `render` is bound to a function that
      returns a prescribed image ({"render": X}); the statements of orientation_loss_rast / mask_loss_rast / loss_function
      before and after that call run as written, and X is recorded as an input of the fixture;
  (3) loss/losses.py evaluates two default arguments `bg=torch.tensor(..., device="cuda")` while it is imported: for the
      duration of that import torch.tensor places device="cuda" requests on the CPU;
  (4) cameras are plain attribute holders with the fields the loss functions read.
"""
import argparse
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
OUT = os.path.join(HERE, "ref_loss_pins.npz")
SMOOTH_SEEDS = [0, 1, 2, 3, 100, 101]


def main():
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
    rng = np.random.default_rng(20261003)
    out = {}
    f32 = lambda a: np.asarray(a, dtype=np.float32)

    # ---- L1 + SSIM ---------------------------------------------------------------------------------------------------------
    cases = [(61, 97), (64, 100), (33, 17), (96, 128), (40, 40)]
    out["meta_ssim_cases"] = np.array(cases)
    for ci, (H, W) in enumerate(cases):
        a = f32(rng.uniform(0, 1, (3, H, W)))
        b = f32(np.clip(a + rng.normal(0, 0.1, (3, H, W)), 0, 1))
        if ci >= 3:                                        # black background outside a box, like a hair render and its target
            keep = np.zeros((H, W), bool)
            keep[H // 4:H // 4 + H // 3, W // 5:W // 5 + W // 2] = True
            a, b = a * keep, b * np.roll(keep, 2, axis=1)
        x = torch.from_numpy(a).requires_grad_(True)
        y = torch.from_numpy(b)
        s = RL.ssim(x, y)
        gs, = torch.autograd.grad(s, x)
        l = RL.l1_loss(x, y)
        gl, = torch.autograd.grad(l, x)
        k = f"ssim{ci}_"
        out[k + "img"], out[k + "gt"] = a, b
        out[k + "ssim"], out[k + "l1"] = np.float64(s.item()), np.float64(l.item())
        out[k + "d_ssim"], out[k + "d_l1"] = gs.numpy(), gl.numpy()

    # ---- strand models: getters, smoothness, schedules ----------------------------------------------------------------------
    topo = np.load(os.path.join(HERE, "ref_topology_pins.npz"))
    out["meta_smooth_seeds"] = np.array(SMOOTH_SEEDS)

    def model(seed, kink=0.0):
        """The state `seed` of ref_topology_pins.npz; kink > 0: endpoints jittered (the synthetic polylines are smooth: at the
        default 30 degree threshold their smoothness term is empty)."""
        k = f"s{seed}_"
        m = HairGaussianModel(sh_degree=3, device="cpu")
        m.ref_strand_root = topo[k + "ref_strand_root"]
        m.strand_root_endpoint_idx = torch.from_numpy(topo[k + "root_idx"])
        m.endpoint_pairs = torch.from_numpy(topo[k + "pairs"])
        P = lambda a: torch.nn.Parameter(torch.from_numpy(a.copy()).requires_grad_(True))
        ep = topo[k + "endpoints"]
        if kink > 0:
            ep = f32(ep + np.random.default_rng(seed).normal(0, kink, ep.shape))
        m._endpoints, m._features_dc, m._features_rest = P(ep), P(topo[k + "f_dc"]), P(topo[k + "f_rest"])
        m._opacity, m._mask, m._width = P(topo[k + "opacity"]), P(topo[k + "mask"]), P(topo[k + "width"])
        m.training_setup(opt)
        m.compute_strands_info()
        return m

    for si, seed in enumerate(SMOOTH_SEEDS):
        m = model(seed, kink=0.0004 * (si % 2))
        k = f"model{seed}_"
        out[k + "endpoints"] = m._endpoints.detach().numpy().copy()
        out[k + "dist_to_scale_factor"] = np.float64(float(m.dist_to_scale_factor))
        P = m.endpoint_pairs.shape[0]
        up = {n: torch.from_numpy(f32(rng.normal(size=s))) for n, s in
              (("scaling", (P, 3)), ("xyz", (P, 3)), ("orientation", (P, 3)), ("opacity", (P, 1)), ("mask", (P, 1)))}
        vals = {"scaling": m.get_scaling, "xyz": m.get_xyz, "orientation": m.get_orientation, "opacity": m.get_opacity, "mask": m.get_mask}
        total = sum((vals[n] * up[n]).sum() for n in vals)
        grads = torch.autograd.grad(total, [m._endpoints, m._width, m._opacity, m._mask])
        for n in vals:
            out[k + n], out[k + "up_" + n] = vals[n].detach().numpy(), up[n].numpy()
        for n, g in zip(("endpoints", "width", "opacity", "mask"), grads):
            out[k + "d_" + n] = g.numpy()
        for th in (30.0, 3.0, 179.0):
            v = RL.angle_smoothness_loss(m, threshold=th)
            kk = k + f"smooth{int(th)}_"
            if torch.is_tensor(v) and v.requires_grad:
                g, = torch.autograd.grad(v, m._endpoints)
                out[kk + "value"], out[kk + "d_endpoints"] = np.float64(v.item()), g.numpy()
            else:
                out[kk + "value"], out[kk + "d_endpoints"] = np.float64(float(v)), np.zeros(tuple(m._endpoints.shape), np.float32)
        its = [1, 100, 1000, 7000, 15000, 30000]
        out["meta_lr_iterations"] = np.array(its)
        sched = []
        for it in its:
            m.update_learning_rate(it)
            lr = [g["lr"] for g in m.optimizer.param_groups if g["name"] == "endpoints"][0]
            sched.append([lr, m.merge_dist_th, m.merge_angle_th])
        out[k + "schedules"] = np.array(sched, dtype=np.float64)

    # ---- per-pixel terms and the whole loss function (render = prescribed images) ------------------------------------------
    head_cases = [(48, 80, True), (37, 53, True), (64, 64, False)]
    out["meta_head_cases"] = np.array([[h, w, int(mk)] for h, w, mk in head_cases])
    for ci, (H, W, with_mask) in enumerate(head_cases):
        m = model(SMOOTH_SEEDS[ci], kink=0.0004)
        k = f"head{ci}_"
        out[k + "endpoints"] = m._endpoints.detach().numpy().copy()
        out[k + "seed"] = np.int64(SMOOTH_SEEDS[ci])
        box = np.zeros((H, W), bool)
        box[H // 5:H // 5 + H // 2, W // 6:W // 6 + W // 2] = True
        img = f32(rng.uniform(0, 1, (3, H, W)) * box)
        gt = f32(np.clip(img + rng.normal(0, 0.08, (3, H, W)), 0, 1) * np.roll(box, 1, axis=0))
        omap = f32(rng.normal(0, 1, (3, H, W)) * box)                 # world-space direction image (black outside the hair)
        mlog = f32(rng.normal(0, 2, (3, H, W)))                       # blended mask channel (its [0] is the logit image)
        R = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        wvt = np.eye(4, dtype=np.float32)
        wvt[:3, :3] = R
        wvt[3, :3] = rng.normal(size=3)
        cam = types.SimpleNamespace(
            original_image=torch.from_numpy(gt),
            world_view_transform=torch.from_numpy(wvt),
            orientation_field=torch.from_numpy(f32(rng.uniform(0, np.pi, (H, W)))),
            orientation_confidence=torch.from_numpy(f32(rng.uniform(0, 1, (H, W)))),
            mask=torch.from_numpy(rng.uniform(size=(H, W)) > 0.4) if with_mask else None,
            float_mask=None)
        cam.float_mask = cam.mask.float() if with_mask else None
        x = torch.from_numpy(img).requires_grad_(True)
        xo = torch.from_numpy(omap).requires_grad_(True)
        xm = torch.from_numpy(mlog).requires_grad_(True)
        calls = []

        def render(camera, pc, bg, scaling_modifier=1.0, override_color=None, debug=False):
            # (2): mask_loss_rast passes get_mask.repeat(1, 3), orientation_loss_rast get_orientation -- told apart by call order
            calls.append(tuple(override_color.shape))
            return {"render": xm if len(calls) == 1 and with_mask else xo}
        RL.render = render
        loss, d = RL.loss_function(m, x, cam, opt)
        assert calls == ([(m.endpoint_pairs.shape[0], 3)] * (2 if with_mask else 1)), calls
        gi, go = torch.autograd.grad(loss, [x, xo], retain_graph=True)
        gm = torch.autograd.grad(loss, xm, retain_graph=True)[0] if with_mask else torch.zeros_like(xm)
        ge, = torch.autograd.grad(loss, m._endpoints, allow_unused=True)
        ge = torch.zeros_like(m._endpoints) if ge is None else ge
        out[k + "image"], out[k + "gt"], out[k + "omap"], out[k + "mask_render"], out[k + "world_view_transform"] = img, gt, omap, mlog, wvt
        out[k + "orientation_field"], out[k + "orientation_confidence"] = cam.orientation_field.numpy(), cam.orientation_confidence.numpy()
        out[k + "mask"] = cam.mask.numpy() if with_mask else np.zeros((0, 0), bool)
        out[k + "total"] = np.float64(loss.item())
        for n in ("l1", "dssim", "mask", "orientation", "smooth"):
            v = d.get(n, None)
            out[k + "term_" + n] = np.float64(float(v)) if v is not None else np.float64("nan")
        out[k + "d_image"], out[k + "d_omap"], out[k + "d_mask_render"], out[k + "d_endpoints"] = gi.numpy(), go.numpy(), gm.numpy(), ge.numpy()
        # the two per-pixel functions on their own, explicit bg (the second case of the mask rule: camera.mask None -> pixels != bg)
        calls.clear()
        bg = torch.zeros(3)

        def render_o(camera, pc, bg_, scaling_modifier=1.0, override_color=None, debug=False):
            return {"render": xo}
        RL.render = render_o
        v = RL.orientation_loss_rast(m, cam, opt, bg)
        g, = torch.autograd.grad(v, xo)
        out[k + "orientation_alone"], out[k + "d_omap_alone"] = np.float64(v.item()), g.numpy()
    for name in ("lambda_dssim", "lambda_mask", "lambda_orientation", "lambda_smooth", "lambda_magnet"):
        out["opt_" + name] = np.float64(getattr(opt, name))
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT} ({os.path.getsize(OUT) / 1024:.0f} KB, {len(out)} arrays)")


if __name__ == "__main__":
    main()
