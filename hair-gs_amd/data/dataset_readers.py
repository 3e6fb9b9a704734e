"""COLMAP scene -> camera infos + point cloud (reference data/dataset_readers.py:34-266).  Images, masks and
orientation maps are read with PIL (the reference mixes PIL and cv2; cv2 is not in the image):
  <scene>/sparse/0/{cameras,images,points3D}.{bin,txt}   images/<name>   masks/<name>
  orientations/<stem>_orientation.png (theta = value * pi / 255)   orientations/<stem>_confidence.png (value / 255)"""
import os
from typing import NamedTuple

import numpy as np
from PIL import Image as PILImage

from data.colmap import (qvec2rotmat, read_extrinsics_binary, read_extrinsics_text, read_intrinsics_binary,
                         read_intrinsics_text, read_points3D_binary, read_points3D_text)
from utils.graphics import BasicPointCloud, focal2fov, getWorld2View2
from utils.ply import element, read_ply, write_ply


class CameraInfo(NamedTuple):
    uid: int
    R: np.ndarray
    T: np.ndarray
    FovY: float
    FovX: float
    image: object
    mask: object
    orientation_field: object
    orientation_confidence: object
    image_path: str
    image_name: str
    width: int
    height: int


class SceneInfo(NamedTuple):
    point_cloud: object
    cameras: list
    nerf_normalization: dict
    ply_path: str


def getNerfppNorm(cam_infos):
    """Scene extent as the reference defines it: 1.1 x the largest distance of a camera centre from their mean."""
    centres = np.stack([np.linalg.inv(getWorld2View2(c.R, c.T))[:3, 3] for c in cam_infos], axis=0)
    mean = centres.mean(axis=0)
    return {"translate": -mean, "radius": float(np.linalg.norm(centres - mean, axis=1).max() * 1.1)}


def _gray(path, width, height, what):
    a = np.asarray(PILImage.open(path).convert("L"))
    if a.shape != (height, width):
        raise AssertionError(f"{what} and image dimensions do not match: {path}")
    return a


def readColmapCameras(cam_extrinsics, cam_intrinsics, images_folder, masks_folder=None, orientations_folder=None):
    infos = []
    for key in cam_extrinsics:
        extr = cam_extrinsics[key]
        intr = cam_intrinsics[extr.camera_id]
        w, h = int(intr.width), int(intr.height)
        if intr.model == "SIMPLE_PINHOLE":
            fx = fy = intr.params[0]
        elif intr.model == "PINHOLE":
            fx, fy = intr.params[0], intr.params[1]
        else:
            raise AssertionError("Colmap camera model not handled: only undistorted datasets (PINHOLE or SIMPLE_PINHOLE cameras) supported!")
        fname = os.path.basename(extr.name)
        image_path = os.path.join(images_folder, fname)
        stem = fname.split(".")[0]
        mask = field = conf = None
        if masks_folder is not None and os.path.exists(os.path.join(masks_folder, fname)):
            mask = (_gray(os.path.join(masks_folder, fname), w, h, "Mask") / 255.0).astype(bool)   # truncation, as the reference
        if orientations_folder is not None:
            po = os.path.join(orientations_folder, f"{stem}_orientation.png")
            pc = os.path.join(orientations_folder, f"{stem}_confidence.png")
            if os.path.exists(po):
                field = _gray(po, w, h, "Orientation").astype(np.float32) * np.pi / 255.0
            if os.path.exists(pc):
                conf = _gray(pc, w, h, "Confidence").astype(np.float32) / 255.0
        infos.append(CameraInfo(uid=intr.id, R=np.transpose(qvec2rotmat(extr.qvec)), T=np.array(extr.tvec),
                                FovY=focal2fov(fy, h), FovX=focal2fov(fx, w), image=PILImage.open(image_path), mask=mask,
                                orientation_field=field, orientation_confidence=conf, image_path=image_path,
                                image_name=stem, width=w, height=h))
    return infos


def fetchPly(path):
    v = element(read_ply(path), "vertex")
    return BasicPointCloud(points=np.stack([v["x"], v["y"], v["z"]], axis=1),
                           colors=np.stack([v["red"], v["green"], v["blue"]], axis=1) / 255.0,
                           normals=np.stack([v["nx"], v["ny"], v["nz"]], axis=1))


def storePly(path, xyz, rgb):
    n = xyz.shape[0]
    arr = np.zeros(n, dtype=[("x", "f4"), ("y", "f4"), ("z", "f4"), ("nx", "f4"), ("ny", "f4"), ("nz", "f4"),
                             ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    arr["x"], arr["y"], arr["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    arr["red"], arr["green"], arr["blue"] = rgb[:, 0], rgb[:, 1], rgb[:, 2]
    write_ply(path, [("vertex", arr)])


def readColmapSceneInfo(path, images=None, llffhold=8):
    sparse = os.path.join(path, "sparse", "0")
    if os.path.exists(os.path.join(sparse, "images.bin")):
        extr = read_extrinsics_binary(os.path.join(sparse, "images.bin"))
        intr = read_intrinsics_binary(os.path.join(sparse, "cameras.bin"))
    else:
        extr = read_extrinsics_text(os.path.join(sparse, "images.txt"))
        intr = read_intrinsics_text(os.path.join(sparse, "cameras.txt"))
    cams = readColmapCameras(extr, intr, os.path.join(path, "images" if images is None else images),
                             os.path.join(path, "masks"), os.path.join(path, "orientations"))
    cams = sorted(cams, key=lambda c: c.image_name)
    ply_path = os.path.join(sparse, "points3D.ply")
    if not os.path.exists(ply_path):   # converted once, like the reference
        if os.path.exists(os.path.join(sparse, "points3D.bin")):
            xyz, rgb, _ = read_points3D_binary(os.path.join(sparse, "points3D.bin"))
        else:
            xyz, rgb, _ = read_points3D_text(os.path.join(sparse, "points3D.txt"))
        # (written under a private name and renamed: another rank of a view-parallel run may be looking for the file now)
        tmp = f"{ply_path}.{os.getpid()}.tmp"
        storePly(tmp, xyz, rgb)
        os.replace(tmp, ply_path)
    try:
        pcd = fetchPly(ply_path)
    except Exception:
        pcd = None
    return SceneInfo(point_cloud=pcd, cameras=cams, nerf_normalization=getNerfppNorm(cams), ply_path=ply_path)
