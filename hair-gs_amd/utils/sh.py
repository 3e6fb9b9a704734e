"""Real spherical-harmonics colour model, degrees 0..3 (counterpart of the reference's utils/sh.py:55-126;
coefficient convention [..., C, (deg+1)^2]; pinned by tests/golden/ref_python_pins.npz)."""
import math

C0 = 0.5 / math.sqrt(math.pi)                                   # 0.28209479177387814
C1 = math.sqrt(3.0 / (4.0 * math.pi))                           # 0.4886025119029199
_k15 = 0.5 * math.sqrt(15.0 / math.pi)                          # 1.0925484305920792
C2 = [_k15, -_k15, 0.25 * math.sqrt(5.0 / math.pi), -_k15, 0.5 * _k15]
_a = 0.25 * math.sqrt(35.0 / (2.0 * math.pi))                   # 0.5900435899266435
_b = 0.25 * math.sqrt(21.0 / (2.0 * math.pi))                   # 0.4570457994644658
C3 = [-_a, 0.5 * math.sqrt(105.0 / math.pi), -_b, 0.25 * math.sqrt(7.0 / math.pi), -_b,
      0.25 * math.sqrt(105.0 / math.pi), -_a]


def sh_basis(deg, dirs):
    """List of the (deg+1)^2 basis values (each [..., 1]) at unit directions dirs[..., 3]."""
    x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
    basis = [C0 + 0 * x]
    if deg > 0:
        basis += [-C1 * y, C1 * z, -C1 * x]
    if deg > 1:
        xx, yy, zz = x * x, y * y, z * z
        basis += [C2[0] * (x * y), C2[1] * (y * z), C2[2] * (2.0 * zz - xx - yy), C2[3] * (x * z), C2[4] * (xx - yy)]
    if deg > 2:
        basis += [C3[0] * y * (3 * xx - yy), C3[1] * (x * y) * z, C3[2] * y * (4 * zz - xx - yy),
                  C3[3] * z * (2 * zz - 3 * xx - 3 * yy), C3[4] * x * (4 * zz - xx - yy), C3[5] * z * (xx - yy),
                  C3[6] * x * (xx - 3 * yy)]
    return basis


def eval_sh(deg, sh, dirs):
    """sum_k basis_k(dirs) * sh[..., k]  -> [..., C]   (deg 0..3)."""
    assert 0 <= deg <= 3, "SH degree 0..3 supported (the rasterizer kernels stop at 3)"
    assert sh.shape[-1] >= (deg + 1) ** 2
    out = None
    for k, bk in enumerate(sh_basis(deg, dirs)):
        term = bk * sh[..., k]
        out = term if out is None else out + term
    return out


def RGB2SH(rgb):
    return (rgb - 0.5) / C0


def SH2RGB(sh):
    return sh * C0 + 0.5
