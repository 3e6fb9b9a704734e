"""Camera containers feeding render() (counterparts of scene/cameras.py:19-132 of the reference).
Matrices are stored TRANSPOSED (row-vector convention): world_view_transform = W2C^T,
full_proj_transform = W2C^T @ P^T, camera_center = inverse(world_view_transform)[3,:3]."""
import numpy as np
import torch
from torch import nn

from utils.graphics import getProjectionMatrix, getWorld2View2


class Camera(nn.Module):
    def __init__(self, colmap_id, R, T, FoVx, FoVy, image, gt_alpha_mask, image_name, uid, mask=None,
                 orientation_field=None, orientation_confidence=None, trans=np.array([0.0, 0.0, 0.0]), scale=1.0,
                 data_device="cuda", image_width=None, image_height=None):
        super().__init__()
        self.uid, self.colmap_id, self.R, self.T = uid, colmap_id, R, T
        self.FoVx, self.FoVy, self.image_name = FoVx, FoVy, image_name
        try:
            self.data_device = torch.device(data_device)
        except Exception:
            self.data_device = torch.device("cuda")
        dev = self.data_device
        if image is not None:
            self.original_image = image.clamp(0.0, 1.0).to(dev)
            self.image_width, self.image_height = self.original_image.shape[2], self.original_image.shape[1]
            if gt_alpha_mask is not None:
                self.original_image = self.original_image * gt_alpha_mask.to(dev)
        else:  # synthetic / render-only camera
            self.original_image = None
            self.image_width, self.image_height = int(image_width), int(image_height)
        self.mask = None if mask is None else mask.to(dev)
        if self.mask is not None:
            self.float_mask = self.mask.to(torch.float32)
            if self.original_image is not None:
                self.masked_image = self.original_image.clone()
                self.masked_image[:, ~self.mask] = 0.0
        self.orientation_field = None if orientation_field is None else orientation_field.to(dev)
        self.orientation_confidence = None if orientation_confidence is None else orientation_confidence.to(dev)
        self.zfar, self.znear = 100.0, 0.01
        self.trans, self.scale = trans, scale
        self.world_view_transform = torch.tensor(getWorld2View2(R, T, trans, scale)).transpose(0, 1).contiguous().to(dev)
        self.projection_matrix = getProjectionMatrix(znear=self.znear, zfar=self.zfar, fovX=FoVx, fovY=FoVy) \
            .transpose(0, 1).contiguous().to(dev)
        self.full_proj_transform = self.world_view_transform @ self.projection_matrix
        self.camera_center = self.world_view_transform.inverse()[3, :3]


class MiniCam:
    def __init__(self, width, height, fovy, fovx, znear, zfar, world_view_transform, full_proj_transform):
        self.image_width, self.image_height = width, height
        self.FoVy, self.FoVx, self.znear, self.zfar = fovy, fovx, znear, zfar
        self.world_view_transform = world_view_transform
        self.full_proj_transform = full_proj_transform
        self.camera_center = torch.inverse(world_view_transform)[3][:3]
