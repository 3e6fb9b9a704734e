#!/usr/bin/env python3
"""Stage II: turn the optimised Gaussian cloud of <model_path> into a strand model and merge strand ends until nothing
is left to merge (reference merge.py).  The newest iteration's point_cloud.ply must be a Gaussian
cloud; the result is saved as iteration_<it + rounds + 1> (the reference's loop index at exit).  Visualisation / logging of the reference are not reproduced.
  python merge.py -s <colmap scene> -m <model dir> [--iterations N]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from argparse import ArgumentParser

import torch

from arguments import GeneralParams, ModelParams, OptimizationParams
from scene.hair_gaussian_model import HairGaussianModel


def merge_rounds(hair_gs, max_rounds, log=print):
    """compute_endpoint_pair_to_merge -> merge_endpoint_pairs -> compute_strands_info until no candidate is left."""
    rounds = 0
    for i in range(1, max_rounds + 1):
        t0 = time.time()
        pairs = hair_gs.compute_endpoint_pair_to_merge()
        if pairs.shape[0] == 0:
            break
        hair_gs.merge_endpoint_pairs(pairs)
        hair_gs.compute_strands_info()
        rounds = i
        log(f"[merge {i}] merged {pairs.shape[0]} endpoint pairs in {time.time() - t0:.3f} s; "
            f"{hair_gs.strands_info.n_strands} strands, {hair_gs.get_xyz.shape[0]} segments")
    return rounds


def main(argv=None):
    parser = ArgumentParser(description="Merging script parameters")
    mp, op, gp = ModelParams(parser), OptimizationParams(parser), GeneralParams(parser)
    args = parser.parse_args(argv)
    from scene import Scene
    scene = Scene(args)
    gaussians = scene.gaussians
    opt = op.extract(args)
    gaussians.training_setup(opt)
    assert not isinstance(gaussians, HairGaussianModel), \
        "merge.py converts the Stage-I Gaussian cloud into strands: the newest iteration is already a strand model"
    with torch.inference_mode():
        hair_gs = gaussians.to_hair_gaussian_model()
        scene.gaussians = hair_gs
        rounds = merge_rounds(hair_gs, opt.iterations)
        # the reference saves under its loop index at exit (merge.py:114-190: the round that found nothing to merge, or the
        # last one allowed)
        scene.save(min(rounds + 1, max(int(opt.iterations), 1)))
    return hair_gs


if __name__ == "__main__":
    main()
