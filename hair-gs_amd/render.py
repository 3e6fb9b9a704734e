#!/usr/bin/env python3
"""Render a trained model from its training cameras (reference render.py).
  python render.py -s <colmap scene> -m <model dir> [--type N]
Types: 0 rgb, 2 mask_foreground, 3 mask_other, 4 orientation_map, 1 rgb_foreground (cleans the model, runs last),
-1 all.  Outputs go to <model>/render/train/iteration_<it>/{renders,gt}/<type name>/<idx>.png, as the reference's."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from argparse import ArgumentParser

import numpy as np
import torch
from PIL import Image

from arguments import GeneralParams, ModelParams, OptimizationParams, get_combined_args
from gaussian_renderer import render
from utils.general import safe_state
from utils.visualization import orientation_map_to_vis

type_map = {-1: "all", 0: "rgb", 1: "rgb_foreground", 2: "mask_foreground", 3: "mask_other", 4: "orientation_map"}


def save_image(t, path):
    """[3,H,W] or [H,W] tensor / array in 0..1 -> 8-bit PNG (what torchvision.utils.save_image writes)."""
    a = t.detach().float().cpu().numpy() if torch.is_tensor(t) else np.asarray(t, dtype=np.float32)
    a = np.clip(a, 0.0, 1.0)
    if a.ndim == 3:
        a = np.transpose(a, (1, 2, 0))
        if a.shape[2] == 1:
            a = a[..., 0]
    Image.fromarray((a * 255.0 + 0.5).astype(np.uint8)).save(path)


def orientation_angles(gaussians, view, background):
    """Rendered strand direction -> per-pixel angle in [0, pi) w.r.t. the image y axis (reference render.py:88-116)."""
    omap = render(view, gaussians, background, override_color=gaussians.get_orientation)["render"].permute(1, 2, 0)
    pix = (omap.flatten(0, 1) @ view.world_view_transform[:3, :3])[:, :2]
    pix = pix / (torch.norm(pix, dim=1, keepdim=True) + gaussians.min_val)
    x, y = pix[:, 0], pix[:, 1]
    y = torch.where(y < gaussians.min_val, y + gaussians.min_val, y)
    theta = torch.atan2(x, y)
    return torch.where(theta < 0, theta + np.pi, theta).reshape(omap.shape[:2])


def render_set(args, name, iteration, views, gaussians, optimization, kind):
    background = torch.zeros(3, dtype=torch.float32, device=args.data_device)
    base = os.path.join(args.model_path, "render", name, f"iteration_{iteration}")
    render_path, gts_path = os.path.join(base, "renders", type_map[kind]), os.path.join(base, "gt", type_map[kind])
    os.makedirs(render_path, exist_ok=True)
    os.makedirs(gts_path, exist_ok=True)
    if kind == 1:
        gaussians.training_setup(optimization)
        gaussians.clean_gaussians()
    th = gaussians.foreground_binarization_th
    frames = None
    if kind in (0, 1) and background.is_cuda and len(views) > 1:
        # the RGB frames of a fixed model: one captured graph re-pointed per view (gaussian_renderer.frames; the image is
        # render()'s, bit for bit); views of different sizes keep the per-call path
        import hgs_runtime as rt
        from gaussian_renderer.frames import FrameRenderer
        try:
            frames = FrameRenderer(gaussians, views, background)
        except rt.HgsError:
            frames = None
    for idx, view in enumerate(views):
        if kind in (0, 1):
            rendering = frames.render(idx)["render"] if frames is not None else render(view, gaussians, background)["render"]
            gt = view.original_image[0:3]
        elif kind == 2:
            rendering = render(view, gaussians, background, override_color=(gaussians.get_mask.repeat(1, 3) >= th).float())["render"][0]
            gt = view.float_mask
        elif kind == 3:
            rendering = render(view, gaussians, background, override_color=(gaussians.get_mask.repeat(1, 3) < th).float())["render"][0]
            gt = (~view.mask).float() if view.mask is not None else None
        elif kind == 4:
            conf = view.orientation_confidence if view.orientation_confidence is not None else torch.zeros(view.image_height, view.image_width)
            rendering = torch.from_numpy(orientation_map_to_vis(orientation_angles(gaussians, view, background), conf) / 255.0).permute(2, 0, 1)
            gt = None if view.orientation_field is None else \
                torch.from_numpy(orientation_map_to_vis(view.orientation_field, conf) / 255.0).permute(2, 0, 1)
        else:
            raise ValueError("Invalid rendering type")
        save_image(rendering, os.path.join(render_path, f"{idx:05d}.png"))
        if gt is not None:
            save_image(gt, os.path.join(gts_path, f"{idx:05d}.png"))


def main(argv=None):
    parser = ArgumentParser(description="Testing script parameters")
    ModelParams(parser, sentinel=True)
    optimization = OptimizationParams(parser)
    GeneralParams(parser)
    parser.add_argument("--skip_train", action="store_true")
    parser.add_argument("--skip_test", action="store_true")
    parser.add_argument("--type", "-t", type=int, default=-1, help="Type of rendering")
    if argv is not None:
        sys.argv = [sys.argv[0]] + list(argv)
    args = get_combined_args(parser)
    for group in (ModelParams, OptimizationParams, GeneralParams):   # no cfg_args next to the model: class defaults
        for name, default, _ in group.FIELDS:
            if getattr(args, name, None) is None:
                setattr(args, name, default() if callable(default) else default)
    print("Rendering " + args.model_path)
    safe_state(getattr(args, "quiet", False))
    from scene import Scene
    with torch.no_grad():
        scene = Scene(args)     # (shuffled like the reference, render.py:136: the file numbers follow that order)
        kinds = [args.type] if args.type != -1 else [0, 2, 3, 4, 1]   # 1 deletes Gaussians: last
        for kind in kinds:
            if not args.skip_train:
                render_set(args, "train", scene.loaded_iter, scene.getCameras(), scene.gaussians, optimization.extract(args), kind)


if __name__ == "__main__":
    main()
