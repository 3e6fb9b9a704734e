"""COLMAP i/o and the scene normalisation against the REFERENCE's own data/colmap.py and dataset_readers.getNerfppNorm, run in
the authoring container (tests/golden/ref_colmap_pins.npz, generator tests/golden/make_ref_colmap_pins.py): this package's
readers parse the bytes the reference's writers produced to the same records, its writers produce the same bytes, the text
readers parse the same files to the same records, the quaternion conversions and the scene extent agree."""
import os

import numpy as np

PINS = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_colmap_pins.npz"))


def _files(tmp_path, names):
    paths = []
    for n in names:
        p = tmp_path / n.replace("_bin", ".bin").replace("_txt", ".txt")
        p.write_bytes(PINS["file_" + n].tobytes())
        paths.append(str(p))
    return paths


def _check_parse(tag, cams, imgs, pts, exact):
    assert list(cams) == [int(v) for v in PINS[tag + "cam_ids"]]
    for cid, c in cams.items():
        assert c.model == str(PINS[f"{tag}cam{cid}_model"]) and [c.width, c.height] == PINS[f"{tag}cam{cid}_wh"].tolist()
        assert np.array_equal(np.asarray(c.params, dtype=np.float64), PINS[f"{tag}cam{cid}_params"])
    assert list(imgs) == [int(v) for v in PINS[tag + "img_ids"]]
    for iid, im in imgs.items():
        k = f"{tag}img{iid}_"
        assert np.array_equal(np.asarray(im.qvec), PINS[k + "qvec"]) and np.array_equal(np.asarray(im.tvec), PINS[k + "tvec"])
        assert im.camera_id == int(PINS[k + "cam"]) and im.name == str(PINS[k + "name"])
        assert np.array_equal(np.asarray(im.xys, dtype=np.float64).reshape(-1, 2), PINS[k + "xys"])
        assert np.array_equal(np.asarray(im.point3D_ids, dtype=np.int64), PINS[k + "p3d"])
        assert np.array_equal(im.qvec2rotmat(), PINS[k + "rotmat"])
    for got, name in zip(pts, ("pts_xyz", "pts_rgb", "pts_err")):
        assert np.array_equal(np.asarray(got, dtype=np.float64), PINS[tag + name]), name


def test_binary_readers_parse_the_references_files(tmp_path):
    from data import colmap as C
    fc, fi, fp = _files(tmp_path, ("cameras_bin", "images_bin", "points3D_bin"))
    _check_parse("bin_", C.read_intrinsics_binary(fc), C.read_extrinsics_binary(fi), C.read_points3D_binary(fp), True)


def test_text_readers_parse_the_same_records(tmp_path):
    from data import colmap as C
    tc, ti, tp = _files(tmp_path, ("cameras_txt", "images_txt", "points3D_txt"))
    _check_parse("txt_", C.read_intrinsics_text(tc), C.read_extrinsics_text(ti), C.read_points3D_text(tp), True)


def test_binary_writers_produce_the_references_bytes(tmp_path):
    from data import colmap as C
    cams = {int(c): C.Camera(id=int(c), model=str(PINS[f"in_cam{c}_model"]), width=int(PINS[f"in_cam{c}_wh"][0]), height=int(PINS[f"in_cam{c}_wh"][1]),
                             params=PINS[f"in_cam{c}_params"]) for c in PINS["in_cam_ids"]}
    imgs = {int(i): C.Image(id=int(i), qvec=PINS[f"in_img{i}_qvec"], tvec=PINS[f"in_img{i}_tvec"], camera_id=int(PINS[f"in_img{i}_cam"]),
                            name=str(PINS[f"in_img{i}_name"]), xys=PINS[f"in_img{i}_xys"], point3D_ids=PINS[f"in_img{i}_p3d"]) for i in PINS["in_img_ids"]}
    pts = {int(p): C.Point3D(id=int(p), xyz=PINS[f"in_pt{p}_xyz"], rgb=PINS[f"in_pt{p}_rgb"], error=float(PINS[f"in_pt{p}_error"]),
                             image_ids=PINS[f"in_pt{p}_image_ids"], point2D_idxs=PINS[f"in_pt{p}_idxs"]) for p in PINS["in_pt_ids"]}
    fc, fi, fp = (str(tmp_path / n) for n in ("c.bin", "i.bin", "p.bin"))
    C.write_cameras_binary(cams, fc)
    C.write_images_binary(imgs, fi)
    C.write_points3D_binary(pts, fp)
    for f, name in ((fc, "cameras_bin"), (fi, "images_bin"), (fp, "points3D_bin")):
        assert open(f, "rb").read() == PINS["file_" + name].tobytes(), name


def test_quaternions_and_scene_extent():
    from types import SimpleNamespace
    from data import colmap as C
    from data.dataset_readers import getNerfppNorm
    R = np.stack([C.qvec2rotmat(q) for q in PINS["quat_in"]])
    assert np.array_equal(R, PINS["quat_rotmat"])
    back = np.stack([C.rotmat2qvec(r) for r in PINS["quat_rotmat"]])
    assert np.allclose(back, PINS["quat_back"], rtol=0, atol=1e-14)        # (an eigen-decomposition: LAPACK's last bits)
    norm = getNerfppNorm([SimpleNamespace(R=r, T=t) for r, t in zip(PINS["nerf_R"], PINS["nerf_T"])])
    assert np.allclose(np.asarray(norm["translate"], dtype=np.float64), PINS["nerf_translate"], rtol=1e-14, atol=0)
    assert abs(float(norm["radius"]) - float(PINS["nerf_radius"])) <= 1e-14 * float(PINS["nerf_radius"])
