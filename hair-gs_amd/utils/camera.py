"""Synthetic camera rigs (counterpart of the reference's utils/camera.py:41-100 `generate_cameras`):
N-1 cameras on a circle around the anchor about one axis + one top view; OpenCV convention (x right, y down,
z forward); returns world-to-camera extrinsics."""
import collections

import numpy as np

ColmapCamera = collections.namedtuple("Camera", ["id", "model", "width", "height", "params"])


def _axis_rotation(axis, angle):
    c, s = np.cos(angle), np.sin(angle)
    if axis == "x":
        return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], float)
    if axis == "y":
        return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], float)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], float)


def generate_cameras(number_cameras, height, width, cam_pose=np.eye(4), anchor_pos=np.array([0, 0, 0]), offset=0.5,
                     rotation_axis="y", focal_length_px=500):
    """Returns ({id: ColmapCamera}, {id: w2c 4x4}).  `cam_pose` is the camera-to-world pose of camera 1."""
    ring = number_cameras - 1
    cams, Es = {}, {}
    anchor = np.asarray(anchor_pos, float)
    for i in range(ring):
        pose = np.array(cam_pose, float)
        pose[:3, 3] -= anchor
        Tm = np.eye(4)
        Tm[:3, :3] = _axis_rotation(rotation_axis, 2 * np.pi * i / ring)
        pose = Tm @ pose
        pose[:3, 3] += anchor
        Es[i + 1] = np.linalg.inv(pose)
        cams[i + 1] = ColmapCamera(i + 1, "SIMPLE_PINHOLE", width, height, [focal_length_px, width / 2, height / 2])
    pose = np.array(cam_pose, float)
    pose[:3, 3] = anchor + np.array([0, offset, 0])
    pose[:3, :3] = _axis_rotation("x", 3 * np.pi / 2) @ pose[:3, :3]
    Es[number_cameras] = np.linalg.inv(pose)
    cams[number_cameras] = ColmapCamera(number_cameras, "SIMPLE_PINHOLE", width, height,
                                        [focal_length_px, width / 2, height / 2])
    return cams, Es
