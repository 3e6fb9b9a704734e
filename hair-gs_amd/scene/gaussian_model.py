"""Stage-I parameter container (counterpart of the reference's scene/gaussian_model.py:33-863): activations,
per-tensor Adam groups, densification (clone / split / prune) with optimizer-state surgery, conversion to the
strand model.  Behaviour follows the reference line by line where cited; the bookkeeping is table-driven
(`_PARAM_ATTRS`) instead of one hand-written branch per tensor.  PLY IO lives in scene/ply_io.py."""
import numpy as np
import torch
from torch import nn
from torch.distributions import Normal

from simple_knn._C import distCUDA2
from utils.general import get_expon_lr_func, inverse_sigmoid, strip_symmetric
from utils.graphics import BasicPointCloud
from utils.sh import RGB2SH
from utils.transform import build_rotation, build_scaling_rotation


def _covariance_from_scaling_rotation(scaling, scaling_modifier, rotation):
    L = build_scaling_rotation(scaling_modifier * scaling, rotation)
    return strip_symmetric(L @ L.transpose(1, 2))


class GaussianModel:
    min_val = 1e-7
    dist_to_scale_factor = 0.5102133812190369  # 1 / Phi^-1(1 - pval/2) at pval = 0.05 (reference :35)
    pval = 0.05
    opacity_th = 0.005
    foreground_binarization_th = 0.25

    # optimizer group name -> attribute, in the reference's group order (reference :216-248)
    _PARAM_ATTRS = (("xyz", "_xyz"), ("f_dc", "_features_dc"), ("f_rest", "_features_rest"), ("opacity", "_opacity"),
                    ("scaling", "_scaling"), ("mask", "_mask"), ("rotation", "_rotation"))

    def __init__(self, sh_degree: int = 3, spatial_lr_scale: float = 1.0, device: str = "cuda"):
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        for _, attr in self._PARAM_ATTRS:
            setattr(self, attr, torch.empty(0))
        self.max_radii2D = torch.empty(0)
        self.xyz_gradient_accum = torch.empty(0)
        self.denom = torch.empty(0)
        self.optimizer = None
        self.spatial_lr_scale = spatial_lr_scale
        self.device = device
        self.ref_strand_root = None
        self.setup_functions()

    def setup_functions(self):
        self.scaling_activation = torch.exp
        self.scaling_inverse_activation = torch.log
        self.covariance_activation = _covariance_from_scaling_rotation
        self.opacity_activation = torch.sigmoid
        self.inverse_opacity_activation = inverse_sigmoid
        self.mask_activation = torch.sigmoid
        self.inverse_mask_activation = inverse_sigmoid
        self.rotation_activation = torch.nn.functional.normalize

    # ---- checkpoint tuple (reference :80-116; no caller in the reference either) ----
    def capture(self):
        return (self.active_sh_degree,) + tuple(getattr(self, a) for _, a in self._capture_attrs()) + (
            self.max_radii2D, self.xyz_gradient_accum, self.denom, self.optimizer.state_dict(), self.spatial_lr_scale)

    def _capture_attrs(self):
        order = ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity", "_mask")
        return [(None, a) for a in order]

    def restore(self, model_args, training_args):
        attrs = [a for _, a in self._capture_attrs()]
        self.active_sh_degree = model_args[0]
        for a, v in zip(attrs, model_args[1:1 + len(attrs)]):
            setattr(self, a, v)
        max_radii2D, grad_accum, denom, opt_dict, self.spatial_lr_scale = model_args[1 + len(attrs):]
        # The checkpoint's Adam moments and statistics are in the checkpoint's order: set up WITHOUT the spatial sort,
        # attach them, and only then re-order -- one permutation moves parameters, moments and statistics together.
        self.training_setup(training_args, sort=False)
        self.max_radii2D, self.xyz_gradient_accum, self.denom = max_radii2D, grad_accum, denom
        self.optimizer.load_state_dict(opt_dict)
        self._maybe_sort_spatially()

    # ---- rasterizer-facing getters (reference :118-157) ----
    @property
    def get_scaling(self):
        return self.scaling_activation(self._scaling)

    @property
    def get_rotation(self):
        return self.rotation_activation(self._rotation)

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_opacity(self):
        return self.opacity_activation(self._opacity)

    @property
    def get_mask(self):
        return self.mask_activation(self._mask)

    def _main_axis_onehot(self, scale):
        onehot = torch.zeros_like(scale)
        onehot.scatter_(1, torch.argmax(scale, dim=1, keepdim=True), 1.0)
        return onehot

    @property
    def get_orientation(self):
        """World-space direction of the longest axis (reference :144-152)."""
        main_axis = self._main_axis_onehot(self.get_scaling)
        return torch.bmm(build_rotation(self._rotation), main_axis.unsqueeze(2)).squeeze(-1)

    def get_covariance(self, scaling_modifier=1):
        return self.covariance_activation(self.get_scaling, scaling_modifier, self._rotation)

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    # ---- initialisation from a point cloud (reference :163-208) ----
    # ---- on-disk format (reference :268-412), scene/ply_io.py ----
    def construct_list_of_attributes(self):
        from scene.ply_io import gaussian_attributes
        return gaussian_attributes(self)

    def save_ply(self, path):
        from scene.ply_io import save_gaussian_ply
        save_gaussian_ply(self, path)

    def load_ply(self, path):
        from scene.ply_io import load_gaussian_ply
        load_gaussian_ply(self, path)

    def create_from_pcd(self, pcd: BasicPointCloud):
        dev = self.device
        pts = torch.tensor(np.asarray(pcd.points)).float().to(dev)
        n = pts.shape[0]
        n_coef = (self.max_sh_degree + 1) ** 2
        features = torch.zeros((n, 3, n_coef), dtype=torch.float, device=dev)
        features[:, :3, 0] = RGB2SH(torch.tensor(np.asarray(pcd.colors)).float().to(dev))
        print("Number of points at initialisation : ", n)
        dist2 = torch.clamp_min(distCUDA2(pts), 0.0000001)
        scales = torch.log(torch.sqrt(dist2))[..., None].repeat(1, 3)
        rots = torch.zeros((n, 4), device=dev)
        rots[:, 0] = 1
        opacities = inverse_sigmoid(0.1 * torch.ones((n, 1), dtype=torch.float, device=dev))
        masks = inverse_sigmoid(0.5 * torch.ones((n, 1), dtype=torch.float, device=dev))
        self._xyz = nn.Parameter(pts.requires_grad_(True))
        self._features_dc = nn.Parameter(features[:, :, 0:1].transpose(1, 2).contiguous().requires_grad_(True))
        self._features_rest = nn.Parameter(features[:, :, 1:].transpose(1, 2).contiguous().requires_grad_(True))
        self._scaling = nn.Parameter(scales.requires_grad_(True))
        self._rotation = nn.Parameter(rots.requires_grad_(True))
        self._opacity = nn.Parameter(opacities.requires_grad_(True))
        self._mask = nn.Parameter(masks.requires_grad_(True))
        self.max_radii2D = torch.zeros((n,), device=dev)

    # ---- optimizer (reference :210-266): Adam(lr=0, eps=1e-15), one group per tensor ----
    def _group_lrs(self, ta):
        return {"xyz": ta.position_lr_init * self.spatial_lr_scale, "f_dc": ta.feature_lr, "f_rest": ta.feature_lr / 20.0,
                "opacity": ta.opacity_lr, "scaling": ta.scaling_lr, "mask": ta.mask_lr, "rotation": ta.rotation_lr}

    _POSITION_GROUP = "xyz"

    fused_adam = True

    def _make_optimizer(self, groups):
        """Adam exactly as the reference configures it (lr=0 default, eps=1e-15, one group per tensor).  On the GPU the
        update of all groups is one fused launch (hgs_runtime.fused.FusedAdam, same update rule, same state layout);
        `fused_adam = False` or CPU tensors use torch.optim.Adam."""
        on_gpu = str(self.device).startswith("cuda") or getattr(self.device, "type", "") == "cuda"
        if self.fused_adam and on_gpu:
            from hgs_runtime.fused import FusedAdam
            return FusedAdam(groups, lr=0.0, eps=1e-15)
        return torch.optim.Adam(groups, lr=0.0, eps=1e-15)

    def _num_primitives(self):
        return self.get_xyz.shape[0]

    def _reset_stats(self):
        n = self._num_primitives()
        self.xyz_gradient_accum = torch.zeros((n, 1), device=self.device)
        self.denom = torch.zeros((n, 1), device=self.device)
        self.max_radii2D = torch.zeros((n,), device=self.device)

    def training_setup(self, training_args, sort=True):
        """sort=False: keep the storage order (restore(): the checkpoint's optimizer state is attached first)."""
        n = self._num_primitives()
        self.xyz_gradient_accum = torch.zeros((n, 1), device=self.device)
        self.denom = torch.zeros((n, 1), device=self.device)
        lrs = self._group_lrs(training_args)
        groups = [{"params": [getattr(self, attr)], "lr": lrs[name], "name": name} for name, attr in self._PARAM_ATTRS]
        self.optimizer = self._make_optimizer(groups)
        self.xyz_scheduler_args = get_expon_lr_func(
            lr_init=training_args.position_lr_init * self.spatial_lr_scale,
            lr_final=training_args.position_lr_final * self.spatial_lr_scale,
            lr_delay_mult=training_args.position_lr_delay_mult, max_steps=training_args.position_lr_max_steps)
        self.set_pval(training_args.pval)
        self.training_args = training_args
        if sort:
            self._maybe_sort_spatially()

    def update_learning_rate(self, iteration):
        for group in self.optimizer.param_groups:
            if group["name"] == self._POSITION_GROUP:
                group["lr"] = self.xyz_scheduler_args(iteration)
                return group["lr"]

    # ---- optimizer-state surgery (reference :421-512) ----
    def _rebind(self, tensors):
        for name, attr in self._PARAM_ATTRS:
            if name in tensors:
                setattr(self, attr, tensors[name])

    def _swap_param(self, group, new_tensor, state_fn):
        old = group["params"][0]
        state = self.optimizer.state.pop(old, None)
        new_param = nn.Parameter(new_tensor.requires_grad_(True))
        group["params"][0] = new_param
        if state is not None:
            state["exp_avg"] = state_fn(state["exp_avg"])
            state["exp_avg_sq"] = state_fn(state["exp_avg_sq"])
            self.optimizer.state[new_param] = state
        return new_param

    def replace_tensor_to_optimizer(self, tensor, name):
        out = {}
        for group in self.optimizer.param_groups:
            if group["name"] == name:
                out[name] = self._swap_param(group, tensor, lambda m: torch.zeros_like(tensor))
        return out

    def _prune_optimizer(self, keep):
        return {g["name"]: self._swap_param(g, g["params"][0][keep], lambda m: m[keep])
                for g in self.optimizer.param_groups}

    def cat_tensors_to_optimizer(self, tensors_dict):
        out = {}
        for g in self.optimizer.param_groups:
            assert len(g["params"]) == 1
            ext = tensors_dict[g["name"]]
            out[g["name"]] = self._swap_param(g, torch.cat((g["params"][0], ext), dim=0),
                                              lambda m, e=ext: torch.cat((m, torch.zeros_like(e)), dim=0))
        return out

    def reset_opacity(self):
        """opacity <- min(opacity, 0.01), Adam moments of the group zeroed (reference :414-419)."""
        new = inverse_sigmoid(torch.min(self.get_opacity, torch.ones_like(self.get_opacity) * 0.01))
        self._rebind(self.replace_tensor_to_optimizer(new, "opacity"))

    def _maybe_sort_spatially(self):
        """training_args.spatial_sort (default on) for a cloud that lives on the GPU: see sort_spatially."""
        if getattr(getattr(self, "training_args", None), "spatial_sort", True) and self.get_xyz.is_cuda and self.get_xyz.shape[0] > 1:
            self.sort_spatially()

    def sort_spatially(self, bits=10):
        """Re-order the Gaussians (parameters, Adam moments, statistics alike) along a Morton curve through their centres.
        Not in the reference and invisible to it -- the order of a cloud's Gaussians is arbitrary -- but decisive for the
        binning kernels: 256 consecutive Gaussians of an unordered cloud touch every occupied tile of the frame, so every
        workgroup sends one atomic per tile to the same few dozen counter lines (C2: preprocess 34 us, scatter 43 us for
        50 k Gaussians, 90 % of it waiting); along the curve a workgroup's Gaussians share a handful of tiles.  Returns
        the permutation applied (new[i] = old[perm[i]])."""
        xyz = self.get_xyz.detach()
        lo, hi = xyz.min(dim=0).values, xyz.max(dim=0).values
        q = ((xyz - lo) / (hi - lo).clamp_min(1e-20) * ((1 << bits) - 1)).to(torch.int64).clamp_(0, (1 << bits) - 1)
        code = torch.zeros(xyz.shape[0], dtype=torch.int64, device=xyz.device)
        for b in range(bits):
            for a in range(3):
                code |= ((q[:, a] >> b) & 1) << (3 * b + a)
        perm = torch.argsort(code, stable=True)
        if self.optimizer is not None:
            self._rebind(self._prune_optimizer(perm))
        else:
            for name, attr in self._PARAM_ATTRS:
                setattr(self, attr, nn.Parameter(getattr(self, attr).detach()[perm].requires_grad_(True)))
        for attr in ("xyz_gradient_accum", "denom", "max_radii2D"):
            t = getattr(self, attr, None)
            if torch.is_tensor(t) and t.shape[0] == perm.shape[0]:
                setattr(self, attr, t[perm])
        return perm

    def prune_points(self, mask):
        keep = ~mask
        self._rebind(self._prune_optimizer(keep))
        self.xyz_gradient_accum = self.xyz_gradient_accum[keep]
        self.denom = self.denom[keep]
        self.max_radii2D = self.max_radii2D[keep]

    def densification_postfix(self, new_xyz, new_features_dc, new_features_rest, new_opacities, new_mask, new_scaling,
                              new_rotation):
        self._rebind(self.cat_tensors_to_optimizer({
            "xyz": new_xyz, "f_dc": new_features_dc, "f_rest": new_features_rest, "opacity": new_opacities,
            "mask": new_mask, "scaling": new_scaling, "rotation": new_rotation}))
        self._reset_stats()

    # ---- densification (reference :551-673) ----
    def densify_and_split(self, grads, grad_threshold, scene_extent, N=2, training_info=None):
        n0 = self.get_xyz.shape[0]
        padded = torch.zeros((n0,), device=self.device)
        padded[: grads.shape[0]] = grads.squeeze()
        sel = (padded >= grad_threshold) & (
            torch.max(self.get_scaling, dim=1).values > self.training_args.percent_dense * scene_extent)
        stds = self.get_scaling[sel].repeat(N, 1)
        samples = torch.normal(mean=torch.zeros((stds.size(0), 3), device=self.device), std=stds)
        rots = build_rotation(self._rotation[sel]).repeat(N, 1, 1)
        new_xyz = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + self.get_xyz[sel].repeat(N, 1)
        new_scaling = self.scaling_inverse_activation(self.get_scaling[sel].repeat(N, 1) / (0.8 * N))
        if training_info is not None:
            training_info.densification_info["split"] = int(sel.sum())
        self.densification_postfix(new_xyz, self._features_dc[sel].repeat(N, 1, 1),
                                   self._features_rest[sel].repeat(N, 1, 1), self._opacity[sel].repeat(N, 1),
                                   self._mask[sel].repeat(N, 1), new_scaling, self._rotation[sel].repeat(N, 1))
        self.prune_points(torch.cat((sel, torch.zeros(N * int(sel.sum()), device=self.device, dtype=bool))))

    def densify_and_clone(self, grads, grad_threshold, scene_extent, training_info=None):
        sel = (torch.norm(grads, dim=-1) >= grad_threshold) & (
            torch.max(self.get_scaling, dim=1).values <= self.training_args.percent_dense * scene_extent)
        if training_info is not None:
            training_info.densification_info["clone"] = int(sel.sum())
        self.densification_postfix(self._xyz[sel], self._features_dc[sel], self._features_rest[sel], self._opacity[sel],
                                   self._mask[sel], self._scaling[sel], self._rotation[sel])

    def densification(self, extent, max_screen_size, training_info=None):
        thr = self.training_args.densify_grad_threshold
        grads = self.xyz_gradient_accum / self.denom
        grads[grads.isnan()] = 0.0
        self.densify_and_clone(grads, thr, extent, training_info=training_info)
        self.densify_and_split(grads, thr, extent, training_info=training_info)
        prune = (self.get_opacity < self.opacity_th).squeeze()
        info = training_info.densification_info if training_info is not None else {}
        info["prune_low_opacity"] = int(prune.sum())
        if max_screen_size:
            big_vs = self.max_radii2D > max_screen_size
            big_ws = self.get_scaling.max(dim=1).values > 0.1 * extent
            prune = prune | big_vs | big_ws
            info["prune_big_ws"] = int(big_ws.sum())
        info["prune_total"] = int(prune.sum())
        if prune.sum() != self.get_xyz.shape[0]:
            self.prune_points(prune)
        self._maybe_sort_spatially()       # (clones and splits were appended at the end)

    def update_densification_stats(self, viewspace_point_tensor, radii, update_filter):
        """max screen radius + accumulated |dL/dmean2D| (pixel grad x (0.5W, 0.5H)) per visible primitive
        (reference gaussian_model.py:675-682 / hair_gaussian_model.py:1401-1408).  Written with torch.where instead
        of boolean-mask assignment: same values, but no nonzero() -> no host synchronisation per iteration."""
        f = update_filter
        # in-place on the persistent statistics tensors (also what makes the step capturable in a HIP graph)
        self.max_radii2D.copy_(torch.where(f, torch.max(self.max_radii2D, radii.to(self.max_radii2D.dtype)), self.max_radii2D))
        g = torch.norm(viewspace_point_tensor.grad[:, :2], dim=-1, keepdim=True)
        self.xyz_gradient_accum.add_(torch.where(f[:, None], g, torch.zeros_like(g)))
        self.denom.add_(f[:, None].to(self.denom.dtype))

    # ---- segment view of a Gaussian (reference :686-725) ----
    def set_dist_to_scale_factor(self, dist_to_scale_factor):
        f = torch.as_tensor(dist_to_scale_factor)
        self.dist_to_scale_factor = f
        self.pval = 2 * (1 - Normal(loc=0, scale=1).cdf(1 / f))

    def set_pval(self, pval):
        p = torch.as_tensor(pval)
        self.pval = p
        self.dist_to_scale_factor = 1 / Normal(loc=0, scale=1).icdf(1 - p / 2).item()

    def get_segment_endpoint(self):
        """Endpoints centre +- R * (main-axis scale / dist_to_scale_factor): [N, 2, 3]."""
        scale = self.get_scaling
        half = self._main_axis_onehot(scale) * scale * (1 / self.dist_to_scale_factor)
        rotated = torch.bmm(build_rotation(self._rotation), half.unsqueeze(2)).squeeze(-1)
        c = self.get_xyz
        return torch.stack((c + rotated, c - rotated), dim=1)

    def compute_foreground_mask(self, lines_only: bool = False):
        """opacity >= 0.005 and mask >= 0.25 (reference :727-733); `lines_only` keeps thin line-like Gaussians."""
        mask = (self.get_opacity >= self.opacity_th).squeeze(1) & (self.get_mask >= self.foreground_binarization_th).squeeze(1)
        if lines_only:
            s = self.get_scaling
            thr = 2.5e-5 * self.dist_to_scale_factor

            def line(a, b, c):  # axis a dominant, b/c thin (the reference's `1-eps < ratio OR ratio < 1+eps` is always true)
                return (s[:, a] / s[:, b] > 5) & (s[:, a] / s[:, c] > 5) & (s[:, b] <= thr) & (s[:, c] <= thr)
            mask = mask & (line(0, 1, 2) ^ line(1, 0, 2) ^ line(2, 0, 1))
        return mask

    def clean_gaussians(self):
        self.prune_points(~self.compute_foreground_mask())

    def to_hair_gaussian_model(self):
        """Every Gaussian becomes one disconnected segment (two fresh endpoints); colour/opacity/mask are cloned;
        width = log(mean over the 3 axes of the scales with the main axis zeroed) (reference :797-859)."""
        from scene.hair_gaussian_model import HairGaussianModel
        hair = HairGaussianModel(sh_degree=self.max_sh_degree, spatial_lr_scale=self.spatial_lr_scale, device=self.device)
        hair.set_dist_to_scale_factor(self.dist_to_scale_factor)
        hair.active_sh_degree = self.active_sh_degree
        n = self.get_xyz.shape[0]
        scale = self.get_scaling
        ends = self.get_segment_endpoint()
        endpoints = torch.cat((ends[:, 0], ends[:, 1]), dim=0)
        others = scale * (1.0 - self._main_axis_onehot(scale))
        width = self.scaling_inverse_activation(torch.mean(others, dim=1, keepdim=True))
        ar = torch.arange(n, device=self.device)
        hair._endpoints = nn.Parameter(endpoints.detach().clone(), requires_grad=True)
        hair.endpoint_pairs = torch.stack((ar, ar + n), dim=1)
        hair._features_dc = nn.Parameter(self._features_dc.detach().clone(), requires_grad=True)
        hair._features_rest = nn.Parameter(self._features_rest.detach().clone(), requires_grad=True)
        hair._opacity = nn.Parameter(self._opacity.detach().clone(), requires_grad=True)
        hair._mask = nn.Parameter(self._mask.detach().clone(), requires_grad=True)
        hair._width = nn.Parameter(width.detach().clone(), requires_grad=True)
        hair.ref_strand_root = self.ref_strand_root
        hair.update_strand_root()
        hair.compute_strands_info()
        hair.training_setup(self.training_args)
        return hair
