"""Camera-matrix helpers with the reference's conventions (utils/graphics.py:18-77); pinned by
tests/golden/ref_python_pins.npz (test_utils_pins.py)."""
import math
from typing import NamedTuple

import numpy as np
import torch


class BasicPointCloud(NamedTuple):
    points: np.ndarray
    colors: np.ndarray
    normals: np.ndarray


def geom_transform_points(points, transf_matrix):
    """Row-vector homogeneous transform with the reference's 1e-7 guard on w (utils/graphics.py:23-31)."""
    hom = torch.cat([points, torch.ones_like(points[:, :1])], dim=1) @ transf_matrix
    return hom[:, :3] / (hom[:, 3:] + 0.0000001)


def getWorld2View(R, t):
    return getWorld2View2(R, t)


def getWorld2View2(R, t, translate=np.array([0.0, 0.0, 0.0]), scale=1.0):
    """World-to-camera 4x4 (fp32).  R is the camera-to-world rotation (COLMAP reader convention), t the w2c
    translation; `translate`/`scale` recentre the camera centre exactly like the reference (:38-50)."""
    w2c = np.eye(4)
    w2c[:3, :3] = np.asarray(R).T
    w2c[:3, 3] = np.asarray(t)
    if np.any(np.asarray(translate) != 0.0) or scale != 1.0:
        c2w = np.linalg.inv(w2c)
        c2w[:3, 3] = (c2w[:3, 3] + translate) * scale
        w2c = np.linalg.inv(c2w)
    else:
        w2c = np.linalg.inv(np.linalg.inv(w2c))  # same round trip as the reference (keeps last-ulp behaviour)
    return np.float32(w2c)


def getProjectionMatrix(znear, zfar, fovX, fovY):
    """OpenCV-style perspective matrix, z in [0,1], w = +z (:52-71)."""
    tan_y, tan_x = math.tan(fovY / 2), math.tan(fovX / 2)
    top, right = tan_y * znear, tan_x * znear
    bottom, left = -top, -right
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def fov2focal(fov, pixels):
    return pixels / (2 * math.tan(fov / 2))


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))
