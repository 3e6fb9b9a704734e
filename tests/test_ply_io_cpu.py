"""PLY formats of the two models: property order as the reference writes it, byte layout of a
hand-checked file, save -> load round trips, ASCII and big-endian inputs."""
import struct

import numpy as np
import pytest
import torch

from scene.gaussian_model import GaussianModel
from scene.hair_gaussian_model import HairGaussianModel
from utils import ply


def _cloud(P=17, deg=2):
    g = torch.Generator().manual_seed(3)
    m = GaussianModel(sh_degree=deg, device="cpu")
    K = (deg + 1) ** 2
    m._xyz = torch.nn.Parameter(torch.randn(P, 3, generator=g))
    m._features_dc = torch.nn.Parameter(torch.randn(P, 1, 3, generator=g))
    m._features_rest = torch.nn.Parameter(torch.randn(P, K - 1, 3, generator=g))
    m._opacity = torch.nn.Parameter(torch.randn(P, 1, generator=g))
    m._mask = torch.nn.Parameter(torch.randn(P, 1, generator=g))
    m._scaling = torch.nn.Parameter(torch.randn(P, 3, generator=g))
    m._rotation = torch.nn.Parameter(torch.randn(P, 4, generator=g))
    return m


def test_gaussian_ply_layout_and_round_trip(tmp_path):
    m = _cloud()
    names = m.construct_list_of_attributes()
    # reference scene/gaussian_model.py:268-281
    assert names == (["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(24)]
                     + ["opacity", "mask", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"])
    path = tmp_path / "point_cloud.ply"
    m.save_ply(str(path))
    raw = path.read_bytes()
    header, body = raw.split(b"end_header\n", 1)
    lines = header.decode().splitlines()
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 17"]
    assert lines[3:] == [f"property float {n}" for n in names]
    assert len(body) == 17 * 4 * len(names)
    row0 = struct.unpack("<" + "f" * len(names), body[:4 * len(names)])
    assert np.allclose(row0[:3], m._xyz[0].detach().numpy()) and row0[3:6] == (0.0, 0.0, 0.0)
    # SH stored channel-major: f_rest_0..7 = red of coefficients 1..8
    assert np.allclose(row0[9:17], m._features_rest[0, :, 0].detach().numpy())
    m2 = GaussianModel(sh_degree=2, device="cpu")
    m2.load_ply(str(path))
    for a in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_mask", "_scaling", "_rotation"):
        assert torch.equal(getattr(m, a).detach(), getattr(m2, a).detach()), a
        assert getattr(m2, a).requires_grad
    assert m2.active_sh_degree == 2 and m2.max_radii2D.shape == (17,)
    with pytest.raises(ValueError):
        GaussianModel(sh_degree=3, device="cpu").load_ply(str(path))


def test_hair_ply_five_elements_and_round_trip(tmp_path):
    pts = np.cumsum(np.random.default_rng(0).normal(size=(6, 9, 3)) * 0.01, axis=1).astype(np.float32)
    pts += np.random.default_rng(1).normal(size=(6, 1, 3)).astype(np.float32)
    m = HairGaussianModel.from_strands(pts, sh_degree=1, device="cpu", ref_strand_root=pts[:, 0])
    path = tmp_path / "hair.ply"
    m.save_ply(str(path))
    els = ply.read_ply(str(path))
    assert [n for n, _ in els] == ["vertex", "edge", "segment", "strand_root_idx", "ref_strand_root"]
    assert els[1][1].dtype.names == ("vertex1", "vertex2") and els[1][1].dtype["vertex1"] == np.dtype("<i4")
    assert els[2][1].dtype.names == tuple(["f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(9)] + ["opacity", "mask", "width"])
    m2 = HairGaussianModel(sh_degree=1, device="cpu")
    m2.load_ply(str(path))
    for a in ("_endpoints", "_features_dc", "_features_rest", "_opacity", "_mask", "_width"):
        assert torch.equal(getattr(m, a).detach(), getattr(m2, a).detach()), a
    assert torch.equal(m.endpoint_pairs, m2.endpoint_pairs)
    assert torch.equal(torch.as_tensor(m.strand_root_endpoint_idx), m2.strand_root_endpoint_idx)
    assert m2.strands_info is not None and len(m2.strands_info.list_strands) == 6
    assert torch.allclose(m.get_xyz, m2.get_xyz)


def test_reader_accepts_ascii_and_big_endian(tmp_path):
    p = tmp_path / "a.ply"
    p.write_text("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 2\nproperty float x\nproperty float y\n"
                 "property uchar red\nelement edge 1\nproperty int vertex1\nproperty int vertex2\nend_header\n"
                 "0.5 1.5 255\n-2 3 7\n0 1\n")
    els = ply.read_ply(str(p))
    assert els[0][1]["x"].tolist() == [0.5, -2.0] and els[0][1]["red"].tolist() == [255, 7]
    assert els[1][1]["vertex2"].tolist() == [1]
    q = tmp_path / "b.ply"
    q.write_bytes(b"ply\nformat binary_big_endian 1.0\nelement vertex 1\nproperty double x\nproperty short k\nend_header\n"
                  + struct.pack(">dh", 1.25, -3))
    e = ply.read_ply(str(q))[0][1]
    assert e["x"][0] == 1.25 and e["k"][0] == -3
    with pytest.raises(ValueError):
        (tmp_path / "c.ply").write_bytes(b"ply\nformat binary_little_endian 1.0\nelement face 1\nproperty list uchar int vertex_indices\nend_header\n")
        ply.read_ply(str(tmp_path / "c.ply"))
