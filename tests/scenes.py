"""Seeded synthetic inputs for the parity tests (numpy only; no product code, no oracle code).

Camera conventions restate scene/cameras.py:93-108 + utils/graphics.py:38-71 of the reference
(row-vector convention: tensors hold the TRANSPOSED matrices); pinned against the reference's own
utils/graphics.py through tests/golden/ref_python_pins.npz (tests/test_oracle_pins.py).
"""
import math

import numpy as np


def world2view(R, t):
    """getWorld2View2(R, t) with translate=0, scale=1 (utils/graphics.py:38-50): R is camera-to-world."""
    Rt = np.zeros((4, 4), np.float64)
    Rt[:3, :3] = R.T
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    return Rt.astype(np.float32)


def projection(znear, zfar, fovx, fovy):
    """getProjectionMatrix (utils/graphics.py:52-71), evaluated in fp32 like the torch original."""
    tx, ty = math.tan(fovx / 2), math.tan(fovy / 2)
    top, right = ty * znear, tx * znear
    Pm = np.zeros((4, 4), np.float32)
    Pm[0, 0] = 2.0 * znear / (right - (-right))
    Pm[1, 1] = 2.0 * znear / (top - (-top))
    Pm[0, 2] = (right + (-right)) / (right - (-right))
    Pm[1, 2] = (top + (-top)) / (top - (-top))
    Pm[3, 2] = 1.0
    Pm[2, 2] = zfar / (zfar - znear)
    Pm[2, 3] = -(zfar * znear) / (zfar - znear)
    return Pm


def make_camera(eye, target, W, H, fovx_deg=60.0, up=(0.0, -1.0, 0.0), znear=0.01, zfar=100.0):
    """OpenCV-style camera (x right, y down, z forward) looking from `eye` at `target`."""
    eye = np.asarray(eye, np.float64)
    fwd = np.asarray(target, np.float64) - eye
    fwd /= np.linalg.norm(fwd)
    down = -np.asarray(up, np.float64)
    right = np.cross(down, fwd)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    R_c2w = np.stack([right, down, fwd], axis=1)  # columns = camera axes in world
    t = -R_c2w.T @ eye
    fovx = math.radians(fovx_deg)
    focal = W / (2 * math.tan(fovx / 2))
    fovy = 2 * math.atan(H / (2 * focal))
    wv = world2view(R_c2w, t).T.copy()  # world_view_transform (transposed)
    pr = projection(znear, zfar, fovx, fovy).T.copy()
    full = (wv @ pr).astype(np.float32)
    campos = np.linalg.inv(wv)[3, :3].astype(np.float32)
    return dict(viewmatrix=wv, projmatrix=full, campos=campos, tanfovx=math.tan(fovx * 0.5),
                tanfovy=math.tan(fovy * 0.5), W=int(W), H=int(H), FoVx=fovx, FoVy=fovy)


def random_scene(P=500, W=96, H=64, seed=0, sh_degree=0, M=None, use_colors_precomp=False, use_cov3D_precomp=False,
                 bg=(0.0, 0.0, 0.0), spread=0.35, scale_lo=0.005, scale_hi=0.05, depth=1.2, behind_frac=0.05,
                 neg_colors=False, fovx_deg=60.0, opacity_lo=0.05, opacity_hi=0.95, depth_levels=0, eye=None,
                 scale_modifier=1.0):
    """Random blob scene in front of one camera.  Some Gaussians sit behind the near plane / off-screen."""
    rng = np.random.default_rng(seed)
    cam = make_camera(eye=(0.1, -0.05, -depth) if eye is None else eye, target=(0, 0, 0), W=W, H=H, fovx_deg=fovx_deg)
    xyz = rng.uniform(-spread, spread, (P, 3)).astype(np.float32)
    xyz[:, 0] *= 1.6 * W / max(W, H) * 1.5
    nb = int(P * behind_frac)
    if nb:
        xyz[:nb, 2] = -depth - rng.uniform(0.0, 0.5, nb).astype(np.float32)  # behind / near the camera
    if depth_levels:
        # quantise positions along the view axis -> many exactly equal depths (sort-stability stress)
        xyz[:, 2] = np.round(xyz[:, 2] * depth_levels) / depth_levels
        xyz[:, 0] = np.round(xyz[:, 0] * 64) / 64
        xyz[:, 1] = np.round(xyz[:, 1] * 64) / 64
    scales = np.exp(rng.uniform(math.log(scale_lo), math.log(scale_hi), (P, 3))).astype(np.float32)
    q = rng.normal(size=(P, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    opac = rng.uniform(opacity_lo, opacity_hi, (P,)).astype(np.float32)
    M = (sh_degree + 1) ** 2 if M is None else M
    shs = (rng.normal(size=(P, M, 3)) * 0.4).astype(np.float32)
    shs[:, 0, :] += 0.8
    scene = dict(cam)
    scene.update(means3D=xyz, opacities=opac, bg=np.asarray(bg, np.float32), sh_degree=sh_degree,
                 scale_modifier=float(scale_modifier), shs=None, colors_precomp=None, scales=None, rotations=None,
                 cov3D_precomp=None)
    if use_colors_precomp:
        col = rng.uniform(0, 1, (P, 3)).astype(np.float32)
        if neg_colors:
            col = col * 2 - 1
        scene["colors_precomp"] = col
    else:
        scene["shs"] = shs
    if use_cov3D_precomp:
        # cov = R S^2 R^T, upper triangle (scene/gaussian_model.py:61-65 convention)
        Rm = quat_to_rotmat(q)
        L = Rm * scales[:, None, :]
        cov = L @ np.transpose(L, (0, 2, 1))
        scene["cov3D_precomp"] = np.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2],
                                           cov[:, 2, 2]], 1).astype(np.float32)
    else:
        scene["scales"] = scales
        scene["rotations"] = q
    return scene


def quat_to_rotmat(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.empty((q.shape[0], 3, 3), q.dtype)
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - r * z); R[:, 0, 2] = 2 * (x * z + r * y)
    R[:, 1, 0] = 2 * (x * y + r * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - r * x)
    R[:, 2, 0] = 2 * (x * z - r * y); R[:, 2, 1] = 2 * (y * z + r * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def strand_scene(n_strands=20, n_seg=50, W=160, H=96, seed=0, bg=(0.0, 0.0, 0.0), fovx_deg=50.0, dist=0.5,
                 use_colors_precomp=False):
    """Thin strand-Gaussians (SURVEY.md 8d C3 generator, reduced): random-walk strands on a sphere."""
    rng = np.random.default_rng(seed)
    cam = make_camera(eye=(0.0, 0.0, -dist), target=(0, 0, 0), W=W, H=H, fovx_deg=fovx_deg)
    roots = rng.normal(size=(n_strands, 3))
    roots = 0.10 * roots / np.linalg.norm(roots, axis=1, keepdims=True)
    d = roots / np.linalg.norm(roots, axis=1, keepdims=True)
    pts = [roots]
    for _ in range(n_seg):
        d = d + rng.normal(size=d.shape) * math.radians(5.0) + np.array([0, 0.02, 0])
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        pts.append(pts[-1] + 0.0025 * d)
    pts = np.stack(pts, 1)  # [S, n_seg+1, 3]
    e0, e1 = pts[:, :-1].reshape(-1, 3), pts[:, 1:].reshape(-1, 3)
    mu = 0.5 * (e0 + e1)
    delta = e1 - e0
    ln = np.linalg.norm(delta, axis=1)
    dirn = delta / ln[:, None]
    f = 0.5102
    scales = np.stack([np.maximum(ln / 2 * f, 1e-7), np.full_like(ln, 1e-4), np.full_like(ln, 1e-4)], 1)
    q = np.stack([1 + dirn[:, 0], np.zeros_like(ln), -dirn[:, 2], dirn[:, 1]], 1)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    P = mu.shape[0]
    scene = dict(cam)
    scene.update(means3D=mu.astype(np.float32), opacities=rng.uniform(0.3, 0.95, P).astype(np.float32),
                 bg=np.asarray(bg, np.float32), sh_degree=0, scale_modifier=1.0, shs=None, colors_precomp=None,
                 scales=scales.astype(np.float32), rotations=q.astype(np.float32), cov3D_precomp=None)
    if use_colors_precomp:
        scene["colors_precomp"] = rng.uniform(-1, 1, (P, 3)).astype(np.float32)
    else:
        scene["shs"] = ((rng.uniform(0, 1, (P, 1, 3)) - 0.5) / 0.28209479177387814).astype(np.float32)
    return scene


def c1_cloud(device="cpu", n_strands=50, n_seg=20, seed=3):
    """BASELINE.json config C1: a 1k-Gaussian Stage-I cloud for merge.py -- every Gaussian is one segment of one of
    n_strands polylines (line-like: main axis = half length x dist_to_scale_factor, 0.1 mm across), stored in shuffled
    order, so that to_hair_gaussian_model() yields 1000 disconnected segments whose ends coincide with their former
    neighbours' to rounding.  Returns (GaussianModel, polylines [S, n_seg+1, 3])."""
    import torch
    from torch import nn
    from scene.gaussian_model import GaussianModel
    from synthetic import strand_polylines
    from utils.transform import calculate_rotation_from_vectors
    pts = strand_polylines(n_strands, n_seg, seed=seed)
    e0, e1 = torch.from_numpy(pts[:, :-1].reshape(-1, 3).copy()), torch.from_numpy(pts[:, 1:].reshape(-1, 3).copy())
    n = e0.shape[0]
    perm = torch.from_numpy(np.random.default_rng(seed).permutation(n))
    e0, e1 = e0[perm], e1[perm]
    m = GaussianModel(sh_degree=0, device=device)
    d = e1 - e0
    half = d.norm(dim=1, keepdim=True) / 2
    x_hat = torch.zeros_like(d)
    x_hat[:, 0] = 1.0
    quat = calculate_rotation_from_vectors(x_hat, d, representation="quat")
    scale = torch.cat((half * m.dist_to_scale_factor, torch.full((n, 2), 1e-4)), dim=1)
    g = torch.Generator().manual_seed(seed)
    to = lambda t: nn.Parameter(t.to(device).contiguous())   # noqa: E731
    m._xyz = to((e0 + e1) / 2)
    m._features_dc, m._features_rest = to(torch.rand(n, 1, 3, generator=g)), to(torch.zeros(n, 0, 3))
    m._scaling, m._rotation = to(torch.log(scale)), to(quat)
    m._opacity, m._mask = to(torch.full((n, 1), 2.0)), to(torch.full((n, 1), 2.0))
    m.max_radii2D = torch.zeros(n, device=device)
    m.ref_strand_root = pts[:, 0].copy()
    return m, pts
