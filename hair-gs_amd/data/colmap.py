"""COLMAP sparse-model files (cameras / images / points3D, binary and text): the camera source of the reference's
datasets (data/colmap.py:126-345 readers, :471-525 writers; called from data/dataset_readers.py:216-266).

The layouts are COLMAP's public ones (src/base/reconstruction.cc):
  cameras.bin   u64 n | n x { i32 camera_id, i32 model_id, u64 width, u64 height, f64 params[num_params(model)] }
  images.bin    u64 n | n x { i32 image_id, f64 qvec[4] (w,x,y,z), f64 tvec[3], i32 camera_id, name\0,
                              u64 m, m x { f64 x, f64 y, i64 point3D_id } }
  points3D.bin  u64 n | n x { u64 id, f64 xyz[3], u8 rgb[3], f64 error, u64 t, t x { i32 image_id, i32 point2D_idx } }
Each file is read in one go and decoded with offsets into the buffer (numpy views for the bulk arrays).
SURVEY.md 8f n4.  The reference module cannot be imported here (its package __init__ needs cv2 / plyfile), so these
readers are checked against bytes assembled by hand from the layout above and by write -> read round trips."""
import collections
import struct

import numpy as np

CameraModel = collections.namedtuple("CameraModel", ["model_id", "model_name", "num_params"])
Camera = collections.namedtuple("Camera", ["id", "model", "width", "height", "params"])
BaseImage = collections.namedtuple("Image", ["id", "qvec", "tvec", "camera_id", "name", "xys", "point3D_ids"])
Point3D = collections.namedtuple("Point3D", ["id", "xyz", "rgb", "error", "image_ids", "point2D_idxs"])

_MODELS = [("SIMPLE_PINHOLE", 3), ("PINHOLE", 4), ("SIMPLE_RADIAL", 4), ("RADIAL", 5), ("OPENCV", 8), ("OPENCV_FISHEYE", 8),
           ("FULL_OPENCV", 12), ("FOV", 5), ("SIMPLE_RADIAL_FISHEYE", 4), ("RADIAL_FISHEYE", 5), ("THIN_PRISM_FISHEYE", 12)]
CAMERA_MODELS = {CameraModel(i, n, k) for i, (n, k) in enumerate(_MODELS)}
CAMERA_MODEL_IDS = {m.model_id: m for m in CAMERA_MODELS}
CAMERA_MODEL_NAMES = {m.model_name: m for m in CAMERA_MODELS}


class Image(BaseImage):
    def qvec2rotmat(self):
        return qvec2rotmat(self.qvec)


def qvec2rotmat(qvec):
    """Unit quaternion (w, x, y, z) -> rotation matrix, in COLMAP's own operation order (every term doubled before it is
    added: the camera matrices, and with them every bit-exact key of the rasterizer, start from these nine numbers;
    tests/test_ref_colmap_pins_cpu.py holds them to the reference's last bit)."""
    w, x, y, z = qvec[0], qvec[1], qvec[2], qvec[3]
    return np.array([[1 - 2 * y ** 2 - 2 * z ** 2, 2 * x * y - 2 * w * z, 2 * z * x + 2 * w * y],
                     [2 * x * y + 2 * w * z, 1 - 2 * x ** 2 - 2 * z ** 2, 2 * y * z - 2 * w * x],
                     [2 * z * x - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x ** 2 - 2 * y ** 2]])


def rotmat2qvec(R):
    """Rotation matrix -> (w, x, y, z) with w >= 0, through the dominant eigenvector of the 4x4 symmetric form."""
    R = np.asarray(R, dtype=np.float64)
    K = np.array([[R[0, 0] - R[1, 1] - R[2, 2], 0, 0, 0],
                  [R[1, 0] + R[0, 1], R[1, 1] - R[0, 0] - R[2, 2], 0, 0],
                  [R[2, 0] + R[0, 2], R[2, 1] + R[1, 2], R[2, 2] - R[0, 0] - R[1, 1], 0],
                  [R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1], R[0, 0] + R[1, 1] + R[2, 2]]]) / 3.0
    vals, vecs = np.linalg.eigh(K)
    q = vecs[[3, 0, 1, 2], np.argmax(vals)]
    return -q if q[0] < 0 else q


# ---- binary ----------------------------------------------------------------------------------------------------------
class _Cursor:
    def __init__(self, path):
        with open(path, "rb") as fh:
            self.buf = fh.read()
        self.pos = 0

    def take(self, fmt):
        st = struct.Struct("<" + fmt)
        vals = st.unpack_from(self.buf, self.pos)
        self.pos += st.size
        return vals

    def array(self, dtype, count):
        dt = np.dtype(dtype)
        a = np.frombuffer(self.buf, dtype=dt, count=count, offset=self.pos)
        self.pos += dt.itemsize * count
        return a

    def cstring(self):
        end = self.buf.index(b"\x00", self.pos)
        s = self.buf[self.pos:end].decode("utf-8")
        self.pos = end + 1
        return s


def read_intrinsics_binary(path):
    c = _Cursor(path)
    (n,) = c.take("Q")
    cams = {}
    for _ in range(n):
        cam_id, model_id, w, h = c.take("iiQQ")
        model = CAMERA_MODEL_IDS[model_id]
        cams[cam_id] = Camera(id=cam_id, model=model.model_name, width=w, height=h,
                              params=np.array(c.array("<f8", model.num_params)))
    return cams


def read_extrinsics_binary(path):
    c = _Cursor(path)
    (n,) = c.take("Q")
    rec = np.dtype([("x", "<f8"), ("y", "<f8"), ("id", "<i8")])
    images = {}
    for _ in range(n):
        vals = c.take("i7di")
        name = c.cstring()
        (m,) = c.take("Q")
        obs = c.array(rec, m)
        images[vals[0]] = Image(id=vals[0], qvec=np.array(vals[1:5]), tvec=np.array(vals[5:8]), camera_id=vals[8], name=name,
                                xys=np.column_stack([obs["x"], obs["y"]]) if m else np.zeros((0, 2)),
                                point3D_ids=np.array(obs["id"], dtype=np.int64))
    return images


def read_points3D_binary(path):
    """(xyz [n,3] f64, rgb [n,3] f64 in 0..255, error [n,1] f64), as the reference returns them."""
    c = _Cursor(path)
    (n,) = c.take("Q")
    xyz, rgb, err = np.empty((n, 3)), np.empty((n, 3)), np.empty((n, 1))
    for i in range(n):
        vals = c.take("Q3d3Bd")
        xyz[i], rgb[i], err[i] = vals[1:4], vals[4:7], vals[7]
        (t,) = c.take("Q")
        c.pos += 8 * t                                  # the track (image id, 2D index pairs) is not used
    return xyz, rgb, err


def write_cameras_binary(cameras, path):
    with open(path, "wb") as fh:
        fh.write(struct.pack("<Q", len(cameras)))
        for cam in cameras.values():
            fh.write(struct.pack("<iiQQ", cam.id, CAMERA_MODEL_NAMES[cam.model].model_id, cam.width, cam.height))
            fh.write(np.asarray(cam.params, dtype="<f8").tobytes())
    return cameras


def write_images_binary(images, path):
    with open(path, "wb") as fh:
        fh.write(struct.pack("<Q", len(images)))
        for im in images.values():
            fh.write(struct.pack("<i7di", im.id, *[float(v) for v in im.qvec], *[float(v) for v in im.tvec], im.camera_id))
            fh.write(im.name.encode("utf-8") + b"\x00")
            fh.write(struct.pack("<Q", len(im.point3D_ids)))
            for xy, pid in zip(im.xys, im.point3D_ids):
                fh.write(struct.pack("<ddq", float(xy[0]), float(xy[1]), int(pid)))


def write_points3D_binary(points3D, path):
    with open(path, "wb") as fh:
        fh.write(struct.pack("<Q", len(points3D)))
        for pt in points3D.values():
            fh.write(struct.pack("<Q3d3Bd", pt.id, *[float(v) for v in pt.xyz], *[int(v) for v in pt.rgb], float(pt.error)))
            fh.write(struct.pack("<Q", len(pt.image_ids)))
            for a, b in zip(pt.image_ids, pt.point2D_idxs):
                fh.write(struct.pack("<ii", int(a), int(b)))


# ---- text ------------------------------------------------------------------------------------------------------------
def _data_lines(path):
    with open(path, "r") as fh:
        for line in fh:
            line = line.strip()
            if line and not line.startswith("#"):
                yield line


def read_intrinsics_text(path):
    cams = {}
    for line in _data_lines(path):
        t = line.split()
        if t[1] != "PINHOLE":   # the reference asserts this too: the rest of the pipeline assumes undistorted pinholes
            raise AssertionError("only PINHOLE cameras are supported in text models")
        cams[int(t[0])] = Camera(id=int(t[0]), model=t[1], width=int(t[2]), height=int(t[3]),
                                 params=np.array([float(v) for v in t[4:]]))
    return cams


def read_extrinsics_text(path):
    images = {}
    with open(path, "r") as fh:
        lines = fh.read().split("\n")
    i = 0
    while i < len(lines):
        line = lines[i].strip()
        i += 1
        if not line or line.startswith("#"):
            continue
        t = line.split()
        obs = lines[i].split() if i < len(lines) else []
        i += 1                                             # the observation line (may be empty) always follows
        images[int(t[0])] = Image(id=int(t[0]), qvec=np.array([float(v) for v in t[1:5]]),
                                  tvec=np.array([float(v) for v in t[5:8]]), camera_id=int(t[8]), name=t[9],
                                  xys=np.column_stack([[float(v) for v in obs[0::3]], [float(v) for v in obs[1::3]]]),
                                  point3D_ids=np.array([int(v) for v in obs[2::3]], dtype=np.int64))
    return images


def read_points3D_text(path):
    rows = [line.split() for line in _data_lines(path)]
    n = len(rows)
    xyz, rgb, err = np.empty((n, 3)), np.empty((n, 3)), np.empty((n, 1))
    for i, t in enumerate(rows):
        xyz[i] = [float(v) for v in t[1:4]]
        rgb[i] = [int(v) for v in t[4:7]]
        err[i] = float(t[7])
    return xyz, rgb, err
