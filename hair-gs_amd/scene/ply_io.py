"""On-disk formats of the two models, through utils/ply.py instead of plyfile.

Gaussian cloud (reference scene/gaussian_model.py:268-412): ONE element `vertex`, float properties
    x y z nx ny nz  f_dc_0..2  f_rest_0..(3(M-1)-1)  opacity mask  scale_0..2  rot_0..3
with the SH tensors stored channel-major ([P,K,3] -> transpose(1,2) -> flatten), normals zero, all raw (pre-activation).
Strand model (reference scene/hair_gaussian_model.py:292-466): FIVE elements, in this order
    vertex          x y z nx ny nz                 endpoints (float)
    edge            vertex1 vertex2                endpoint_pairs (int)
    segment         f_dc_* f_rest_* opacity mask width   per-Gaussian attributes (float)
    strand_root_idx strand_root_idx                (int)
    ref_strand_root x y z                          (float)
Loading sets active_sh_degree = max_sh_degree and, for strands, rebuilds strands_info, as the reference does."""
import os

import numpy as np
import torch
from torch import nn

from utils.ply import element, read_ply, table, write_ply


def _sh_columns(t):
    """[P,K,3] -> [P,3K] channel-major, the reference's transpose(1,2).flatten(1)."""
    return t.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy()


def _sh_tensor(cols, k, device):
    """[P,3k] channel-major columns -> parameter tensor [P,k,3]."""
    arr = np.asarray(cols, dtype=np.float32).reshape(cols.shape[0], 3, k)
    return torch.tensor(arr, dtype=torch.float, device=device).transpose(1, 2).contiguous()


def _param(t):
    return nn.Parameter(t.requires_grad_(True))


def _numbered(arr, prefix):
    names = [n for n in arr.dtype.names if n.startswith(prefix)]
    names.sort(key=lambda n: int(n.split("_")[-1]))
    return names


def _columns(arr, names):
    return np.stack([np.asarray(arr[n], dtype=np.float32) for n in names], axis=1) if names else np.zeros((arr.shape[0], 0), np.float32)


# ---- Gaussian cloud --------------------------------------------------------------------------------------------------
def gaussian_attributes(model):
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(model._features_dc.shape[1] * model._features_dc.shape[2])]
    names += [f"f_rest_{i}" for i in range(model._features_rest.shape[1] * model._features_rest.shape[2])]
    names += ["opacity", "mask"]
    names += [f"scale_{i}" for i in range(model._scaling.shape[1])]
    names += [f"rot_{i}" for i in range(model._rotation.shape[1])]
    return names


def save_gaussian_ply(model, path):
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    xyz = model._xyz.detach().cpu().numpy()
    cols = np.concatenate((xyz, np.zeros_like(xyz), _sh_columns(model._features_dc), _sh_columns(model._features_rest),
                           model._opacity.detach().cpu().numpy(), model._mask.detach().cpu().numpy(),
                           model._scaling.detach().cpu().numpy(), model._rotation.detach().cpu().numpy()), axis=1)
    write_ply(path, [("vertex", table(cols, gaussian_attributes(model)))])


def load_gaussian_ply(model, path):
    v = read_ply(path)[0][1]
    dev = model.device
    n_rest = 3 * (model.max_sh_degree + 1) ** 2 - 3
    rest_names = _numbered(v, "f_rest_")
    if len(rest_names) != n_rest:
        raise ValueError(f"{path}: {len(rest_names)} f_rest properties, expected {n_rest} for sh degree {model.max_sh_degree}")
    f32 = dict(dtype=torch.float, device=dev)
    model._xyz = _param(torch.tensor(_columns(v, ["x", "y", "z"]), **f32))
    model._features_dc = _param(_sh_tensor(_columns(v, ["f_dc_0", "f_dc_1", "f_dc_2"]), 1, dev))
    model._features_rest = _param(_sh_tensor(_columns(v, rest_names), n_rest // 3, dev))
    model._opacity = _param(torch.tensor(_columns(v, ["opacity"]), **f32))
    model._mask = _param(torch.tensor(_columns(v, ["mask"]), **f32))
    model._scaling = _param(torch.tensor(_columns(v, _numbered(v, "scale_")), **f32))
    model._rotation = _param(torch.tensor(_columns(v, _numbered(v, "rot")), **f32))
    model.max_radii2D = torch.zeros((model._xyz.shape[0],), device=dev)
    model.active_sh_degree = model.max_sh_degree


# ---- strand model ----------------------------------------------------------------------------------------------------
def hair_attributes(model):
    names = [f"f_dc_{i}" for i in range(model._features_dc.shape[1] * model._features_dc.shape[2])]
    names += [f"f_rest_{i}" for i in range(model._features_rest.shape[1] * model._features_rest.shape[2])]
    return names + ["opacity", "mask", "width"]


def save_hair_ply(model, path):
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    ep = model._endpoints.detach().cpu().numpy()
    pairs = model.endpoint_pairs.detach().cpu().numpy()
    seg = np.concatenate((_sh_columns(model._features_dc), _sh_columns(model._features_rest),
                          model._opacity.detach().cpu().numpy(), model._mask.detach().cpu().numpy(),
                          model._width.detach().cpu().numpy()), axis=1)
    roots = torch.as_tensor(model.strand_root_endpoint_idx).detach().cpu().numpy().reshape(-1, 1)
    ref = np.asarray(model.ref_strand_root, dtype=np.float32).reshape(-1, 3)
    write_ply(path, [("vertex", table(np.concatenate((ep, np.zeros_like(ep)), axis=1), ["x", "y", "z", "nx", "ny", "nz"])),
                     ("edge", table(pairs, ["vertex1", "vertex2"], "i4")),
                     ("segment", table(seg, hair_attributes(model))),
                     ("strand_root_idx", table(roots, ["strand_root_idx"], "i4")),
                     ("ref_strand_root", table(ref, ["x", "y", "z"]))])


def load_hair_ply(model, path):
    els = read_ply(path)
    if len(els) != 5:
        raise ValueError(f"{path}: a strand model has 5 elements (vertex, edge, segment, strand_root_idx, ref_strand_root), got {len(els)}")
    vert, edge, seg, root, ref = (e[1] for e in els)
    dev = model.device
    n_rest = 3 * (model.max_sh_degree + 1) ** 2 - 3
    rest_names = _numbered(seg, "f_rest_")
    if len(rest_names) != n_rest:
        raise ValueError(f"{path}: {len(rest_names)} f_rest properties, expected {n_rest} for sh degree {model.max_sh_degree}")
    f32 = dict(dtype=torch.float, device=dev)
    model._endpoints = _param(torch.tensor(_columns(vert, ["x", "y", "z"]), **f32))
    model.endpoint_pairs = torch.tensor(np.stack((edge["vertex1"], edge["vertex2"]), axis=1).astype(np.int64), device=dev)
    model._features_dc = _param(_sh_tensor(_columns(seg, ["f_dc_0", "f_dc_1", "f_dc_2"]), 1, dev))
    model._features_rest = _param(_sh_tensor(_columns(seg, rest_names), n_rest // 3, dev))
    model._opacity = _param(torch.tensor(_columns(seg, ["opacity"]), **f32))
    model._mask = _param(torch.tensor(_columns(seg, ["mask"]), **f32))
    model._width = _param(torch.tensor(_columns(seg, ["width"]), **f32))
    model.active_sh_degree = model.max_sh_degree
    model.strand_root_endpoint_idx = torch.tensor(np.asarray(root["strand_root_idx"]).astype(np.int64), device=dev)
    model.ref_strand_root = _columns(ref, ["x", "y", "z"])
    n = model._opacity.shape[0]
    model.max_radii2D = torch.zeros((n,), device=dev)
    model.xyz_gradient_accum = torch.zeros((n, 1), device=dev)
    model.denom = torch.zeros((n, 1), device=dev)
    model._smooth_pairs = None
    model.compute_strands_info()
