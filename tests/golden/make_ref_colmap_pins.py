"""Generates tests/golden/ref_colmap_pins.npz by RUNNING the reference's own COLMAP i/o (data/colmap.py) and scene
normalisation (data/dataset_readers.py::getNerfppNorm) in the authoring container.  Stored: the BYTES the reference's writers
produce for a synthetic reconstruction (cameras.bin / images.bin / points3D.bin: data files, as uint8 arrays), what the
reference's readers parse from them and from hand-written text files of the public COLMAP text layout, quaternion <-> matrix
conversions, and getNerfppNorm of the cameras.

Executed, unedited:  write_cameras_binary / write_images_binary / write_points3D_binary (:471-530), read_intrinsics_binary /
read_extrinsics_binary / read_points3D_binary (:168-307), read_intrinsics_text / read_extrinsics_text / read_points3D_text
(:126-166, 203-228, 309-342), qvec2rotmat / rotmat2qvec (:56-96); dataset_readers.py:57-79 getNerfppNorm.
Absent third-party imports (cv2, plyfile, ...): tests/golden/_ref_harness.py; the reference's `data` package is entered
without running its __init__.py.
"""
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "ref_colmap_pins.npz")


def main():
    sys.path.insert(0, HERE)
    from _ref_harness import REF, enter_reference
    enter_reference()
    pkg = types.ModuleType("data")
    pkg.__path__ = [os.path.join(REF, "data")]
    sys.modules["data"] = pkg
    import data.colmap as RC
    import data.dataset_readers as RD
    rng = np.random.default_rng(11)
    out = {}
    cams = {1: RC.Camera(id=1, model="PINHOLE", width=1920, height=1080, params=np.array([1500.5, 1499.25, 960.0, 540.0])),
            2: RC.Camera(id=2, model="SIMPLE_PINHOLE", width=640, height=480, params=np.array([525.0, 320.0, 240.0])),
            7: RC.Camera(id=7, model="PINHOLE", width=800, height=1200, params=np.array([900.0, 905.0, 400.5, 600.5]))}
    images = {}
    for i, (iid, cid) in enumerate([(3, 1), (1, 2), (10, 7), (4, 1), (5, 1)]):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        n2d = int(rng.integers(0, 6))
        images[iid] = RC.Image(id=iid, qvec=q, tvec=rng.normal(size=3) * 2, camera_id=cid, name=f"view_{iid:03d}.png",
                               xys=rng.uniform(0, 500, size=(n2d, 2)), point3D_ids=rng.integers(-1, 20, size=n2d).astype(np.int64))
    points = {}
    for pid in (2, 5, 6, 11):
        tl = int(rng.integers(1, 5))
        points[pid] = RC.Point3D(id=pid, xyz=rng.normal(size=3), rgb=rng.integers(0, 256, size=3).astype(np.uint8), error=float(rng.uniform(0, 2)),
                                 image_ids=rng.integers(1, 11, size=tl).astype(np.int32), point2D_idxs=rng.integers(0, 6, size=tl).astype(np.int32))
    tmp = tempfile.mkdtemp(prefix="hgs_colmap_")
    fc, fi, fp = (os.path.join(tmp, n) for n in ("cameras.bin", "images.bin", "points3D.bin"))
    RC.write_cameras_binary(cams, fc)
    RC.write_images_binary(images, fi)
    RC.write_points3D_binary(points, fp)
    for name, f in (("cameras_bin", fc), ("images_bin", fi), ("points3D_bin", fp)):
        out["file_" + name] = np.frombuffer(open(f, "rb").read(), dtype=np.uint8)
    # what went in (so that this package's WRITERS can be given the same records)
    out["in_cam_ids"] = np.array(list(cams))
    for cid, c in cams.items():
        out[f"in_cam{cid}_model"], out[f"in_cam{cid}_wh"], out[f"in_cam{cid}_params"] = np.array(c.model), np.array([c.width, c.height]), c.params
    out["in_img_ids"] = np.array(list(images))
    for iid, im in images.items():
        k = f"in_img{iid}_"
        out[k + "qvec"], out[k + "tvec"], out[k + "cam"], out[k + "name"], out[k + "xys"], out[k + "p3d"] = im.qvec, im.tvec, np.array(im.camera_id), np.array(im.name), im.xys, im.point3D_ids
    out["in_pt_ids"] = np.array(list(points))
    for pid, pt in points.items():
        k = f"in_pt{pid}_"
        out[k + "xyz"], out[k + "rgb"], out[k + "error"], out[k + "image_ids"], out[k + "idxs"] = pt.xyz, pt.rgb, np.array(pt.error), pt.image_ids, pt.point2D_idxs

    def dump_parse(tag, rcams, rimgs, rpts):
        out[tag + "cam_ids"] = np.array(list(rcams))
        for cid, c in rcams.items():
            out[f"{tag}cam{cid}_model"], out[f"{tag}cam{cid}_wh"], out[f"{tag}cam{cid}_params"] = np.array(c.model), np.array([c.width, c.height]), np.asarray(c.params, dtype=np.float64)
        out[tag + "img_ids"] = np.array(list(rimgs))
        for iid, im in rimgs.items():
            k = f"{tag}img{iid}_"
            out[k + "qvec"], out[k + "tvec"], out[k + "cam"], out[k + "name"] = np.asarray(im.qvec), np.asarray(im.tvec), np.array(im.camera_id), np.array(im.name)
            out[k + "xys"], out[k + "p3d"] = np.asarray(im.xys, dtype=np.float64).reshape(-1, 2), np.asarray(im.point3D_ids, dtype=np.int64)
            out[k + "rotmat"] = im.qvec2rotmat()
        out[tag + "pts_xyz"], out[tag + "pts_rgb"], out[tag + "pts_err"] = rpts
    dump_parse("bin_", RC.read_intrinsics_binary(fc), RC.read_extrinsics_binary(fi), RC.read_points3D_binary(fp))
    # the public text layout, written by hand
    tc, ti, tp = (os.path.join(tmp, n) for n in ("cameras.txt", "images.txt", "points3D.txt"))
    with open(tc, "w") as f:
        f.write("# Camera list with one line of data per camera:\n#   CAMERA_ID, MODEL, WIDTH, HEIGHT, PARAMS[]\n# Number of cameras: 3\n")
        for cid, c in cams.items():
            if c.model == "PINHOLE":          # (the reference's text reader asserts PINHOLE, :217-221)
                f.write(f"{cid} {c.model} {c.width} {c.height} " + " ".join(repr(float(v)) for v in c.params) + "\n")
    with open(ti, "w") as f:
        f.write("# Image list with two lines of data per image:\n#   IMAGE_ID, QW, QX, QY, QZ, TX, TY, TZ, CAMERA_ID, NAME\n#   POINTS2D[] as (X, Y, POINT3D_ID)\n")
        for iid, im in images.items():
            f.write(f"{iid} " + " ".join(repr(float(v)) for v in list(im.qvec) + list(im.tvec)) + f" {im.camera_id} {im.name}\n")
            f.write(" ".join(f"{repr(float(x))} {repr(float(y))} {int(p)}" for (x, y), p in zip(im.xys, im.point3D_ids)) + "\n")
    with open(tp, "w") as f:
        f.write("# 3D point list with one line of data per point:\n#   POINT3D_ID, X, Y, Z, R, G, B, ERROR, TRACK[] as (IMAGE_ID, POINT2D_IDX)\n")
        for pid, pt in points.items():
            f.write(f"{pid} " + " ".join(repr(float(v)) for v in pt.xyz) + " " + " ".join(str(int(v)) for v in pt.rgb) + f" {repr(float(pt.error))} "
                    + " ".join(f"{int(a)} {int(b)}" for a, b in zip(pt.image_ids, pt.point2D_idxs)) + "\n")
    for name, f in (("cameras_txt", tc), ("images_txt", ti), ("points3D_txt", tp)):
        out["file_" + name] = np.frombuffer(open(f, "rb").read(), dtype=np.uint8)
    dump_parse("txt_", RC.read_intrinsics_text(tc), RC.read_extrinsics_text(ti), RC.read_points3D_text(tp))
    # quaternions
    qs = rng.normal(size=(64, 4)); qs /= np.linalg.norm(qs, axis=1, keepdims=True)
    out["quat_in"] = qs
    out["quat_rotmat"] = np.stack([RC.qvec2rotmat(q) for q in qs])
    out["quat_back"] = np.stack([RC.rotmat2qvec(R) for R in out["quat_rotmat"]])
    # scene normalisation (densification's scene extent = radius): dataset_readers.py:57-79
    Rs = [np.transpose(RC.qvec2rotmat(im.qvec)) for im in images.values()]          # readColmapCameras :107 stores R transposed
    Ts = [np.array(im.tvec) for im in images.values()]
    infos = [types.SimpleNamespace(R=R, T=T) for R, T in zip(Rs, Ts)]
    norm = RD.getNerfppNorm(infos)
    out["nerf_R"], out["nerf_T"] = np.stack(Rs), np.stack(Ts)
    out["nerf_translate"], out["nerf_radius"] = np.asarray(norm["translate"], dtype=np.float64), np.float64(norm["radius"])
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT} ({os.path.getsize(OUT) / 1024:.0f} KB, {len(out)} arrays)")


if __name__ == "__main__":
    main()
