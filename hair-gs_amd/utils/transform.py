"""Rotation helpers (counterparts of the reference's utils/transform.py:7-86), device-agnostic and without
pytorch3d: the x-axis-to-direction quaternion is evaluated in closed form."""
import numpy as np
import torch


def build_rotation(r):
    """Unit-normalised quaternion (w,x,y,z) [N,4] -> rotation matrices [N,3,3]."""
    q = r / torch.sqrt((r * r).sum(dim=1, keepdim=True))
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    rows = [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
            2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
            2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]
    return torch.stack(rows, dim=1).reshape(-1, 3, 3)


def build_scaling_rotation(s, r):
    """L = R(r) @ diag(s) so that cov = L L^T (reference :37-47)."""
    return build_rotation(r) * s[:, None, :]


def rot_to_wxyz_quat(rot: np.ndarray) -> np.ndarray:
    from scipy.spatial.transform import Rotation
    x, y, z, w = Rotation.from_matrix(rot).as_quat()
    return np.array([w, x, y, z])


def cross_product_to_skew_symmetric(v):
    z = torch.zeros_like(v[:, 0])
    return torch.stack([z, -v[:, 2], v[:, 1], v[:, 2], z, -v[:, 0], -v[:, 1], v[:, 0], z], dim=1).reshape(-1, 3, 3)


def _sqrt_positive_part(x):
    """sqrt(max(x, 0)) with a ZERO sub-gradient where x <= 0 (a plain clamp+sqrt back-propagates inf * 0 = NaN)."""
    out = torch.zeros_like(x)
    pos = x > 0
    out[pos] = torch.sqrt(x[pos])
    return out


def matrix_to_quaternion(R):
    """Rotation matrices [N,3,3] -> quaternions (w,x,y,z) with w >= 0 (largest-component branch selection, the
    same rule pytorch3d.transforms.matrix_to_quaternion uses; q and -q are the same rotation)."""
    m00, m11, m22 = R[:, 0, 0], R[:, 1, 1], R[:, 2, 2]
    q_abs = _sqrt_positive_part(torch.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22, 1 - m00 + m11 - m22,
                                             1 - m00 - m11 + m22], dim=1))
    cand = torch.stack([
        torch.stack([q_abs[:, 0] ** 2, R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]], -1),
        torch.stack([R[:, 2, 1] - R[:, 1, 2], q_abs[:, 1] ** 2, R[:, 1, 0] + R[:, 0, 1], R[:, 0, 2] + R[:, 2, 0]], -1),
        torch.stack([R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] + R[:, 0, 1], q_abs[:, 2] ** 2, R[:, 1, 2] + R[:, 2, 1]], -1),
        torch.stack([R[:, 1, 0] - R[:, 0, 1], R[:, 2, 0] + R[:, 0, 2], R[:, 2, 1] + R[:, 1, 2], q_abs[:, 3] ** 2], -1),
    ], dim=1)
    cand = cand / (2.0 * q_abs[:, :, None].clamp(min=0.1))
    best = q_abs.argmax(dim=1)
    q = cand[torch.arange(R.shape[0], device=R.device), best]
    return torch.where(q[:, :1] < 0, -q, q)


def calculate_rotation_from_vectors(v1, v2, representation="mat", eps=1e-7):
    """Rotation taking v1 (unit) onto v2 (any length): R = I + K + K^2/(1+c), K = skew(v1 x v2_hat),
    c = clamp(v1 . v2_hat) (reference :69-86).  Returns R, or its quaternion when representation == "quat"."""
    v2 = v2 / torch.norm(v2, dim=1, keepdim=True)
    c = torch.clamp((v1 * v2).sum(dim=1), -1 + eps, 1 - eps)
    K = cross_product_to_skew_symmetric(torch.cross(v1, v2, dim=1))
    R = torch.eye(3, device=K.device, dtype=K.dtype)[None] + K + torch.bmm(K, K) / (1 + c)[:, None, None]
    if representation == "quat":
        return matrix_to_quaternion(R)
    return R


def xaxis_to_direction_quaternion(d):
    """Closed form of calculate_rotation_from_vectors(x_hat, d, "quat") for unit d away from d = -x_hat:
    q = normalize(1 + d.x, 0, -d.z, d.y)   (half-angle form)."""
    q = torch.stack([1 + d[:, 0], torch.zeros_like(d[:, 0]), -d[:, 2], d[:, 1]], dim=1)
    return q / torch.norm(q, dim=1, keepdim=True)
