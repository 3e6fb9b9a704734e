"""COLMAP sparse models, dataset readers and Scene on a synthetic capture written to a temp dir."""
import os
import struct
from types import SimpleNamespace

import numpy as np
import pytest
from PIL import Image as PILImage

from data import colmap
from data.dataset_readers import getNerfppNorm, readColmapSceneInfo


def test_binary_layouts_against_hand_assembled_bytes(tmp_path):
    """Bytes laid out by hand from COLMAP's documented record formats (not produced by this repo's writers)."""
    cam = struct.pack("<Q", 1) + struct.pack("<iiQQ", 7, 1, 640, 480) + struct.pack("<4d", 500.0, 510.0, 320.0, 240.0)
    (tmp_path / "cameras.bin").write_bytes(cam)
    cams = colmap.read_intrinsics_binary(str(tmp_path / "cameras.bin"))
    assert cams[7].model == "PINHOLE" and (cams[7].width, cams[7].height) == (640, 480)
    assert cams[7].params.tolist() == [500.0, 510.0, 320.0, 240.0]
    img = struct.pack("<Q", 1) + struct.pack("<i7di", 3, 1.0, 0.0, 0.0, 0.0, 0.1, 0.2, 0.3, 7) + b"view_03.png\x00"
    img += struct.pack("<Q", 2) + struct.pack("<ddq", 1.5, 2.5, 11) + struct.pack("<ddq", 3.5, 4.5, -1)
    (tmp_path / "images.bin").write_bytes(img)
    ims = colmap.read_extrinsics_binary(str(tmp_path / "images.bin"))
    assert ims[3].name == "view_03.png" and ims[3].camera_id == 7 and ims[3].tvec.tolist() == [0.1, 0.2, 0.3]
    assert ims[3].xys.tolist() == [[1.5, 2.5], [3.5, 4.5]] and ims[3].point3D_ids.tolist() == [11, -1]
    pts = struct.pack("<Q", 2)
    pts += struct.pack("<Q3d3Bd", 11, 1.0, 2.0, 3.0, 10, 20, 30, 0.5) + struct.pack("<Q", 1) + struct.pack("<ii", 3, 0)
    pts += struct.pack("<Q3d3Bd", 12, -1.0, 0.0, 4.0, 255, 0, 7, 1.5) + struct.pack("<Q", 0)
    (tmp_path / "points3D.bin").write_bytes(pts)
    xyz, rgb, err = colmap.read_points3D_binary(str(tmp_path / "points3D.bin"))
    assert xyz.tolist() == [[1, 2, 3], [-1, 0, 4]] and rgb.tolist() == [[10, 20, 30], [255, 0, 7]] and err.ravel().tolist() == [0.5, 1.5]


def test_quaternion_conversions():
    rng = np.random.default_rng(0)
    for _ in range(20):
        q = rng.normal(size=4); q /= np.linalg.norm(q); q = q if q[0] >= 0 else -q
        R = colmap.qvec2rotmat(q)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(R) - 1) < 1e-12
        assert np.allclose(colmap.rotmat2qvec(R), q, atol=1e-9)
    assert np.allclose(colmap.qvec2rotmat([np.cos(0.25), 0, 0, np.sin(0.25)]),   # half a radian about z
                       [[np.cos(0.5), -np.sin(0.5), 0], [np.sin(0.5), np.cos(0.5), 0], [0, 0, 1]])


def _write_side_files(root, seed=3):
    """hair_eval_data.npz / head_reconstruction_data.npz as the reference's scripts write them (data/eval_data.py:23-37,
    data/head_reconstruction_data.py:20-45): ground-truth strand points + directions, head and scalp vertices."""
    rng = np.random.default_rng(seed)
    strands = np.cumsum(rng.normal(size=(6, 8, 3)) * 0.05, axis=1) + rng.normal(size=(6, 1, 3)) * 0.5
    pts = strands[:, :-1].reshape(-1, 3)
    dirs = (strands[:, 1:] - strands[:, :-1]).reshape(-1, 3) * 3.0                 # (not unit: the loader normalises)
    sid = np.repeat(np.arange(6), 7)
    edges = np.stack([np.arange(41), np.arange(1, 42)], 1)
    np.savez(root / "hair_eval_data.npz", points=pts, directions=dirs, points_id_to_strand_id=sid, edges=edges)
    np.savez(root / "head_reconstruction_data.npz", head_verts=rng.normal(size=(50, 3)), scalp_verts=rng.normal(size=(30, 3)))


def _write_capture(root, n_views=3, W=32, H=24, text=False):
    sparse = root / "sparse" / "0"
    for d in (sparse, root / "images", root / "masks", root / "orientations"):
        d.mkdir(parents=True, exist_ok=True)
    rng = np.random.default_rng(1)
    cams = {1: colmap.Camera(id=1, model="PINHOLE", width=W, height=H, params=np.array([40.0, 42.0, W / 2, H / 2]))}
    images = {}
    for i in range(n_views):
        a = 0.7 * i
        R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        name = f"v{i:02d}.png"
        images[i + 1] = colmap.Image(id=i + 1, qvec=colmap.rotmat2qvec(R), tvec=np.array([0.1 * i, 0.0, 2.0]), camera_id=1,
                                     name=name, xys=np.zeros((0, 2)), point3D_ids=np.zeros(0, np.int64))
        PILImage.fromarray(rng.integers(0, 255, (H, W, 3), dtype=np.uint8)).save(root / "images" / name)
        PILImage.fromarray(((rng.random((H, W)) > 0.5) * 255).astype(np.uint8)).save(root / "masks" / name)
        PILImage.fromarray(rng.integers(0, 255, (H, W), dtype=np.uint8)).save(root / "orientations" / f"v{i:02d}_orientation.png")
        PILImage.fromarray(rng.integers(0, 255, (H, W), dtype=np.uint8)).save(root / "orientations" / f"v{i:02d}_confidence.png")
    pts = {k: colmap.Point3D(id=k, xyz=rng.normal(size=3), rgb=rng.integers(0, 255, 3), error=0.1, image_ids=np.array([1]),
                             point2D_idxs=np.array([0])) for k in range(1, 41)}
    if text:
        with open(sparse / "cameras.txt", "w") as fh:
            fh.write("# Camera list\n1 PINHOLE %d %d 40.0 42.0 %f %f\n" % (W, H, W / 2, H / 2))
        with open(sparse / "images.txt", "w") as fh:
            fh.write("# Image list\n")
            for im in images.values():
                fh.write("%d %s %s 1 %s\n\n" % (im.id, " ".join(repr(float(v)) for v in im.qvec), " ".join(repr(float(v)) for v in im.tvec), im.name))
        with open(sparse / "points3D.txt", "w") as fh:
            for p in pts.values():
                fh.write("%d %r %r %r %d %d %d 0.1 1 0\n" % (p.id, *[float(v) for v in p.xyz], *[int(v) for v in p.rgb]))
    else:
        colmap.write_cameras_binary(cams, str(sparse / "cameras.bin"))
        colmap.write_images_binary(images, str(sparse / "images.bin"))
        colmap.write_points3D_binary(pts, str(sparse / "points3D.bin"))
    return cams, images, pts


@pytest.mark.parametrize("text", [False, True])
def test_scene_info_from_capture(tmp_path, text):
    cams, images, pts = _write_capture(tmp_path, text=text)
    info = readColmapSceneInfo(str(tmp_path))
    assert [c.image_name for c in info.cameras] == ["v00", "v01", "v02"]
    c1 = info.cameras[1]
    assert np.allclose(c1.R, colmap.qvec2rotmat(images[2].qvec).T) and np.allclose(c1.T, images[2].tvec)
    assert np.isclose(c1.FovX, 2 * np.arctan(32 / (2 * 40.0))) and np.isclose(c1.FovY, 2 * np.arctan(24 / (2 * 42.0)))
    assert c1.mask.dtype == bool and c1.mask.shape == (24, 32)
    assert c1.orientation_field.max() <= np.pi + 1e-6 and c1.orientation_confidence.max() <= 1.0
    assert info.point_cloud.points.shape == (40, 3) and info.point_cloud.colors.max() <= 1.0
    assert os.path.exists(info.ply_path)
    norm = getNerfppNorm(info.cameras)
    assert norm["radius"] > 0 and np.allclose(info.nerf_normalization["radius"], norm["radius"])


def test_orientation_colour_wheel():
    from utils.visualization import orientation_map_to_vis
    th = np.array([[0.0, np.pi / 3, 2 * np.pi / 3, np.pi / 2]])
    vis = orientation_map_to_vis(th, np.zeros_like(th))
    # hue = 2 * uint8(180 * theta / pi) degrees: the uint8 truncation (60 -> 59 after rounding) is the reference's
    for px, want in zip(vis[0], ([255, 0, 0], [0, 255, 0], [0, 0, 255], [0, 255, 255])):
        assert np.abs(px.astype(int) - np.array(want)).max() <= 9, (px, want)
    assert orientation_map_to_vis(th, np.ones_like(th)).max() == 0


def test_image_size_rule_of_the_reference():
    """scene/cameras.py:136-160: -r 1 / 2 / 4 / 8 divide; -1 keeps the size up to 1600 px of width and scales wider images to
    1600; any other value is the target width."""
    from scene.scene import target_size
    assert target_size(1920, 1080, -1) == (1600, 900)
    assert target_size(1920, 1080, 1) == (1920, 1080)
    assert target_size(1000, 800, -1) == (1000, 800)
    assert target_size(1920, 1080, 2) == (960, 540)
    assert target_size(1920, 1080, 800) == (800, 450)
    assert target_size(1001, 801, 2) == (round(1001 / 2), round(801 / 2))
    assert target_size(1920, 1080, 1, resolution_scale=2.0) == (960, 540)

