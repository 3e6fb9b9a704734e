"""Generates tests/golden/ref_python_pins.npz by RUNNING the reference's own code in the authoring
container (never on the GPU box).  Only numeric inputs/outputs are stored.

Reference pieces executed (imported by file path, no stubs, no edits):
  * utils/sh.py::eval_sh, RGB2SH, SH2RGB            -> pins SH->RGB of oracle/raster_oracle.c
  * utils/graphics.py::getWorld2View2, getProjectionMatrix, focal2fov, fov2focal
                                                     -> pins the camera-matrix conventions (tests/scenes.py,
                                                        hair-gs_amd/utils/graphics.py)
  * c_utils/c_utils.pyx::filter_strand_list_segments (built by oracle/build_ref.py outside the repository, $HGS_REF_OUT)
                                                     -> pins oracle/strand_oracle.c
  * arguments/__init__.py                            -> default hyper-parameters (training-step constants)
Everything else on the hot path is CUDA and cannot be executed here ("parity unpinned").
"""
import importlib.util
import math
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    sh = _load("ref_sh", "utils/sh.py")
    gr = _load("ref_graphics", "utils/graphics.py")
    out = {}
    rng = np.random.default_rng(20251002)

    # ---- SH evaluation: degrees 0..3, fp32 torch, exactly as gaussian_renderer/__init__.py:92-100 uses it
    P = 257
    xyz = rng.normal(size=(P, 3)).astype(np.float32)
    campos = np.array([0.3, -0.2, -1.5], np.float32)
    feats = (rng.normal(size=(P, 16, 3)) * 0.5).astype(np.float32)  # [P, M, 3] like pc.get_features
    out["sh_xyz"], out["sh_campos"], out["sh_feats"] = xyz, campos, feats
    t_feats = torch.from_numpy(feats)
    shs_view = t_feats.transpose(1, 2).reshape(-1, 3, 16)
    dir_pp = torch.from_numpy(xyz) - torch.from_numpy(campos)[None]
    dirn = dir_pp / dir_pp.norm(dim=1, keepdim=True)
    for deg in range(4):
        rgb = sh.eval_sh(deg, shs_view, dirn)
        out[f"sh_rgb_deg{deg}"] = torch.clamp_min(rgb + 0.5, 0.0).numpy()
        out[f"sh_raw_deg{deg}"] = (rgb + 0.5).numpy()
    c = rng.uniform(0, 1, (11, 3)).astype(np.float32)
    out["rgb2sh_in"], out["rgb2sh_out"] = c, sh.RGB2SH(torch.from_numpy(c)).numpy()
    out["sh2rgb_out"] = sh.SH2RGB(torch.from_numpy(c)).numpy()

    # ---- camera matrices (scene/cameras.py:93-108 recipe executed with the reference's graphics.py)
    cams = []
    for i in range(6):
        A = rng.normal(size=(3, 3))
        Q, _ = np.linalg.qr(A)
        if np.linalg.det(Q) < 0:
            Q[:, 0] *= -1
        R = Q  # camera-to-world rotation as stored by the COLMAP reader (transposed w2c)
        T = rng.normal(size=3)
        W, H = [(1920, 1080), (800, 800), (1000, 1000), (97, 61), (640, 480), (33, 17)][i]
        focal = [960.0, 400.0, 500.0, 80.0, 700.0, 20.0][i]
        fovx, fovy = gr.focal2fov(focal, W), gr.focal2fov(focal, H)
        wv = torch.tensor(gr.getWorld2View2(R, T, np.array([0.0, 0.0, 0.0]), 1.0)).transpose(0, 1)
        proj = gr.getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)
        full = (wv.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
        center = wv.inverse()[3, :3]
        cams.append(dict(R=R, T=T, W=W, H=H, focal=focal, fovx=fovx, fovy=fovy, wv=wv.numpy(), proj=proj.numpy(),
                         full=full.numpy(), center=center.numpy(), focal_back=gr.fov2focal(fovx, W)))
    for k in cams[0]:
        out["cam_" + k] = np.stack([np.asarray(cm[k], np.float64 if k in ("R", "T") else None) for cm in cams])

    # ---- c_utils.filter_strand_list_segments (reference .pyx built by oracle/build_ref.py)
    from oracle import build_ref
    build_ref.build()
    cu = build_ref.load()
    lens = [5, 1, 0, 2, 9, 3, 1, 40]
    strands = np.empty(len(lens), dtype=object)
    rows = []
    for j, n in enumerate(lens):
        a = rng.integers(0, 100000, size=(n, 2)).astype(np.int64)
        strands[j] = a
        rows.append(a)
    out["strand_lens"] = np.array(lens, np.int64)
    out["strand_rows"] = np.concatenate(rows, 0)
    out["strand_pairs"] = cu.filter_strand_list_segments(strands)
    empty = np.empty(0, dtype=object)
    out["strand_pairs_empty"] = cu.filter_strand_list_segments(empty)

    # ---- defaults from arguments/__init__.py (training-step constants)
    args = _load("ref_arguments", "arguments/__init__.py")
    import argparse
    p = argparse.ArgumentParser()
    op = args.OptimizationParams(p)
    mp = args.ModelParams(p)
    for k, v in vars(op).items():
        if isinstance(v, (int, float, bool)):
            out["opt_" + k.lstrip("_")] = np.array(v)
    out["model_sh_degree"] = np.array(mp.sh_degree)

    np.savez_compressed(os.path.join(HERE, "ref_python_pins.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
