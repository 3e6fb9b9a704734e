"""Scene: a COLMAP capture + the model being optimised (reference scene/__init__.py:30-134).
 * cameras from <source_path>/sparse/0 (+ images / masks / orientations), extent = nerf_normalization radius;
 * model = Gaussian cloud initialised from the sparse points, or the newest <model_path>/point_cloud/iteration_N/
   point_cloud.ply (a one-element file is a Gaussian cloud, a five-element file a strand model);
 * save(iteration) writes that file;
 * the two npz side files of a capture, when present (reference :103-122): `hair_eval_data.npz` (ground-truth strands for the
   metrics: points, directions, points_id_to_strand_id, edges -> `scene.gt`) and `head_reconstruction_data.npz` (head_verts,
   scalp_verts -> `scene.head_reconstruction`; the scalp vertices become the model's `ref_strand_root`, which Stage II / III
   orient the strands by: merge.py cannot run without them)."""

import json
import os
import random

import numpy as np
import torch

from data.dataset_readers import readColmapSceneInfo
from scene.cameras import Camera
from scene.gaussian_model import GaussianModel
from scene.hair_gaussian_model import HairGaussianModel
from utils.ply import read_ply


def search_for_max_iteration(folder):
    return max(int(name.split("_")[-1]) for name in os.listdir(folder))


_WARNED_LARGE = [False]


def target_size(orig_w, orig_h, resolution, resolution_scale=1.0):
    """The reference's image size rule (scene/cameras.py:136-160): --resolution 1 / 2 / 4 / 8 divides both sides (rounded);
    -1 keeps the size unless the image is wider than 1600 px, which is then scaled to 1600 wide (with a one-time notice);
    any other value is the target WIDTH.  (True 1080p therefore needs -r 1.)"""
    if resolution in (1, 2, 4, 8):
        return round(orig_w / (resolution_scale * resolution)), round(orig_h / (resolution_scale * resolution))
    if resolution == -1:
        if orig_w > 1600:
            if not _WARNED_LARGE[0]:
                print("[ INFO ] Encountered quite large input images (>1.6K pixels width), rescaling to 1.6K.\n "
                      "If this is not desired, please explicitly specify '--resolution/-r' as 1")
                _WARNED_LARGE[0] = True
            down = orig_w / 1600
        else:
            down = 1
    else:
        down = orig_w / resolution
    scale = float(down) * float(resolution_scale)
    return int(orig_w / scale), int(orig_h / scale)


def camera_from_info(uid, info, resolution=-1, data_device="cuda", resolution_scale=1.0):
    """CameraInfo -> Camera (reference scene/cameras.py _loadCam): image to [3,H,W] in 0..1 at the size of target_size(),
    optional alpha as a mask multiplier; the mask / orientation planes follow the image's size (nearest neighbour) where they
    differ -- the reference hands them over as they are."""
    img = info.image
    w, h = img.size
    tw, th = target_size(w, h, resolution, resolution_scale)
    if (tw, th) != (w, h):
        img = img.resize((tw, th))
    arr = torch.from_numpy(np.asarray(img, dtype=np.float32) / 255.0)
    arr = arr[..., None] if arr.ndim == 2 else arr
    arr = arr.permute(2, 0, 1)
    rgb, alpha = arr[:3], (arr[3:4] if arr.shape[0] == 4 else None)

    def plane(a, dtype):
        if a is None:
            return None
        t = torch.from_numpy(np.ascontiguousarray(a))
        if tuple(t.shape[-2:]) != tuple(rgb.shape[1:]):
            t = torch.nn.functional.interpolate(t[None, None].float(), size=rgb.shape[1:], mode="nearest")[0, 0]
        return t.to(dtype)
    return Camera(colmap_id=info.uid, R=info.R, T=info.T, FoVx=info.FovX, FoVy=info.FovY, image=rgb, gt_alpha_mask=alpha,
                  image_name=info.image_name, uid=uid, mask=plane(info.mask, torch.bool),
                  orientation_field=plane(info.orientation_field, torch.float32),
                  orientation_confidence=plane(info.orientation_confidence, torch.float32), data_device=data_device)


def camera_to_json(uid, info):
    w2c = np.eye(4)
    w2c[:3, :3], w2c[:3, 3] = info.R.transpose(), info.T
    c2w = np.linalg.inv(w2c)
    from utils.graphics import fov2focal
    return {"id": uid, "img_name": info.image_name, "width": info.width, "height": info.height,
            "position": c2w[:3, 3].tolist(), "rotation": [r.tolist() for r in c2w[:3, :3]],
            "fy": fov2focal(info.FovY, info.height), "fx": fov2focal(info.FovX, info.width)}


class Scene:
    def __init__(self, args, shuffle=True, resolution_scales=(1.0,)):
        self.model_path = args.model_path
        self.loaded_iter = None
        info = readColmapSceneInfo(args.source_path, getattr(args, "images", None))
        os.makedirs(self.model_path, exist_ok=True)
        pc_dir = os.path.join(self.model_path, "point_cloud")
        if os.path.isdir(pc_dir) and os.listdir(pc_dir):
            self.loaded_iter = search_for_max_iteration(pc_dir)
        elif int(os.environ.get("RANK", "0")) == 0:
            # first run: keep the input cloud and the camera list next to the outputs, like the reference (view-parallel run:
            # rank 0 alone writes to the model directory)
            with open(info.ply_path, "rb") as src, open(os.path.join(self.model_path, "input.ply"), "wb") as dst:
                dst.write(src.read())
            with open(os.path.join(self.model_path, "cameras.json"), "w") as fh:
                json.dump([camera_to_json(i, c) for i, c in enumerate(info.cameras)], fh)
        cams = list(info.cameras)
        if shuffle:
            random.shuffle(cams)
        self.cameras_extent = info.nerf_normalization["radius"]
        dev = getattr(args, "data_device", "cuda")
        self.cameras = {s: [camera_from_info(i, c, getattr(args, "resolution", -1), dev, resolution_scale=s) for i, c in enumerate(cams)]
                        for s in resolution_scales}
        if self.loaded_iter is None:
            self.gaussians = GaussianModel(args.sh_degree, self.cameras_extent, device=dev)
            self.gaussians.create_from_pcd(info.point_cloud)
            self.loaded_iter = 0
        else:
            path = os.path.join(pc_dir, f"iteration_{self.loaded_iter}", "point_cloud.ply")
            cls = GaussianModel if len(read_ply(path)) == 1 else HairGaussianModel
            self.gaussians = cls(args.sh_degree, self.cameras_extent, device=dev)
            self.gaussians.load_ply(path)
        # ground truth for the evaluation metrics (reference :103-107; data/eval_data.py:23-37: directions normalised on load)
        self.gt = None
        gt_path = os.path.join(args.source_path, "hair_eval_data.npz")
        if os.path.exists(gt_path):
            from data.eval_data import load_hair_eval_data_npz
            self.gt = load_hair_eval_data_npz(gt_path)
            self.gt_edges = self.gt.edges
        # head reconstruction (reference :109-122): the scalp vertices are the reference strand roots
        self.head_reconstruction = None
        head_path = os.path.join(args.source_path, "head_reconstruction_data.npz")
        if os.path.exists(head_path):
            from data.head_reconstruction_data import load_head_reconstruction_data_npz
            self.head_reconstruction = load_head_reconstruction_data_npz(head_path)
            self.gaussians.ref_strand_root = np.asarray(self.head_reconstruction.scalp_verts)
            if isinstance(self.gaussians, HairGaussianModel):
                self.gaussians.update_strand_root()
                self.gaussians.compute_strands_info()

    def save(self, iteration=0):
        """Writes point_cloud/iteration_<loaded_iter + iteration>/point_cloud.ply: `iteration` counts from the loaded state,
        as in the reference (:124-131)."""
        if self.loaded_iter:
            iteration += self.loaded_iter
        self.gaussians.save_ply(os.path.join(self.model_path, "point_cloud", f"iteration_{iteration}", "point_cloud.ply"))

    def getCameras(self, scale=1.0):
        return self.cameras[scale]
