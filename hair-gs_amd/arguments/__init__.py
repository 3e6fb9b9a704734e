"""Hyper-parameter groups with the reference's names, defaults and CLI spelling (arguments/__init__.py:22-148).
Table-driven: each group lists (name, default, shorthand?).  Defaults are pinned against the reference by
tests/golden/ref_python_pins.npz (tests/test_utils_pins.py)."""
import os
from argparse import ArgumentParser, Namespace


class GroupParams:
    pass


class ParamGroup:
    NAME = "Parameters"
    FIELDS = ()  # (name, default, has_shorthand)

    def __init__(self, parser: ArgumentParser = None, fill_none=False):
        for name, default, _ in self.FIELDS:
            setattr(self, name, default() if callable(default) else default)
        self._finalise()
        if parser is not None:
            group = parser.add_argument_group(self.NAME)
            for name, _, short in self.FIELDS:
                value = getattr(self, name)
                flags = ["--" + name] + (["-" + name[0]] if short else [])
                default = None if fill_none else value
                if isinstance(value, bool):
                    group.add_argument(*flags, default=default, action="store_true")
                else:
                    group.add_argument(*flags, default=default, type=type(value))

    def _finalise(self):
        pass

    def extract(self, args):
        g = GroupParams()
        names = {f[0] for f in self.FIELDS}
        for k, v in vars(args).items():
            if k in names:
                setattr(g, k, v)
        return g


class ModelParams(ParamGroup):
    NAME = "Loading Parameters"
    FIELDS = (("source_path", "", True), ("model_path", "", True), ("images", "images", True), ("sh_degree", 0, False),
              ("resolution", -1, True), ("data_device", "cuda", False), ("eval", False, False))

    def __init__(self, parser=None, sentinel=False):
        super().__init__(parser, sentinel)

    def extract(self, args):
        g = super().extract(args)
        g.source_path = os.path.abspath(g.source_path)
        return g


class OptimizationParams(ParamGroup):
    NAME = "Optimization Parameters"
    FIELDS = (
        ("iterations", 30000, False), ("position_lr_init", 0.00016, False), ("position_lr_final", 0.0000016, False),
        ("position_lr_delay_mult", 0.01, False), ("position_lr_max_steps", 30000, False), ("scaling_lr", 0.005, False),
        ("feature_lr", 0.025, False), ("opacity_lr", 0.05, False), ("mask_lr", 0.01, False), ("lambda_dssim", 0.2, False),
        ("lambda_orientation", 100.0, False), ("lambda_mask", 0.01, False), ("pval", 0.05, False),
        ("bidirectional_eval", True, False), ("rotation_lr", 0.001, False), ("lambda_smooth", 0.005, False),
        ("lambda_magnet", 0.0, False), ("bidirectional_merge", False, False), ("num_points_strand", 80, False),
        ("merge_interval", 100, False), ("merge_dist_th_init", 2e-3, False), ("merge_dist_th_final", 4e-3, False),
        ("merge_angle_th_init", 20, False), ("merge_angle_th_final", 40, False), ("growth_interval", 100000, False),
        ("growth_averaging_points", 3, False), ("percent_dense", 0.01, False), ("opacity_reset_interval", 3000, False),
        ("densify_from_iter", 500, False), ("densify_until_iter", 27000.0, False), ("densification_interval", 100, False),
        ("prune_max_radii_2d", 1000, False), ("densify_grad_threshold", 0.0002, False),
        # ---- this package's own (not in the reference): the captured training loop (train.training / GraphedStep)
        # views a (re-)capture warms up on: its binning capacity is capacity_slack x their largest instance count; a view that
        # needs more is caught by the headroom check and the iterations since the last checkpoint are run again (exact, only time
        # is lost): raise it for captures whose views differ much in coverage
        ("capture_warmup_views", 2, False), ("capacity_slack", 2.0, False),
    )

    def _finalise(self):
        self.position_lr_max_steps = self.iterations
        self.densify_until_iter = self.iterations * 0.9


class GeneralParams(ParamGroup):
    NAME = "General Parameters"
    FIELDS = (("quiet", False, False), ("logger", "tensorboard", False), ("ip", "127.0.0.1", False), ("port", 6009, False),
              ("vis2d", False, False), ("update_vis2d_frequency", 30000, False), ("vis3d", False, False),
              ("save_frequency", 5000, False), ("eval_frequency", 30000, False))


def get_combined_args(parser: ArgumentParser):
    """Command line overrides <model_path>/cfg_args (a `Namespace(...)` repr written at training time)."""
    import sys
    cmd = parser.parse_args(sys.argv[1:])
    merged = {}
    try:
        with open(os.path.join(cmd.model_path, "cfg_args")) as f:
            merged = dict(vars(eval(f.read(), {"Namespace": Namespace})))
    except (TypeError, FileNotFoundError):
        pass
    merged.update({k: v for k, v in vars(cmd).items() if v is not None})
    return Namespace(**merged)
