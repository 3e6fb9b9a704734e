"""End-to-end plumbing around the hot path on the GPU: COLMAP capture -> Scene -> Stage-I steps ->
PLY -> render.py; strand model PLY -> Scene resume -> render.py orientation / mask outputs."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch
from PIL import Image as PILImage

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _args(src, model, **kw):
    return SimpleNamespace(source_path=str(src), model_path=str(model), images="images", sh_degree=0, resolution=-1,
                           data_device="cuda", eval=False, **kw)


def test_scene_train_save_resume_render(tmp_path):
    from tests.test_dataset_io_cpu import _write_capture
    from arguments import OptimizationParams
    from scene import Scene
    from scene.gaussian_model import GaussianModel
    from train import training_step
    import render as render_cli
    src, model = tmp_path / "capture", tmp_path / "out"
    _write_capture(src, n_views=3, W=64, H=48)
    scene = Scene(_args(src, model), shuffle=False)
    assert isinstance(scene.gaussians, GaussianModel) and scene.loaded_iter == 0
    assert os.path.exists(model / "input.ply") and os.path.exists(model / "cameras.json")
    cams = scene.getCameras()
    assert cams[0].original_image.shape == (3, 48, 64) and cams[0].mask.dtype == torch.bool
    opt = OptimizationParams()
    opt.enable_topology = False
    scene.gaussians.training_setup(opt)
    bg = torch.zeros(3, device="cuda")
    for it in range(1, 4):
        loss, _, _ = training_step(scene.gaussians, cams[it % 3], opt, bg, it, extent=scene.cameras_extent)
    assert torch.isfinite(loss)
    scene.save(3)
    before = scene.gaussians.get_xyz.detach().clone()
    scene2 = Scene(_args(src, model), shuffle=False)
    assert scene2.loaded_iter == 3 and torch.equal(scene2.gaussians.get_xyz.detach(), before)
    render_cli.main(["-s", str(src), "-m", str(model), "--type", "0", "--quiet"])
    out = model / "render" / "train" / "iteration_3" / "renders" / "rgb"
    files = sorted(os.listdir(out))
    assert files == ["00000.png", "00001.png", "00002.png"]
    assert np.asarray(PILImage.open(out / files[0])).shape == (48, 64, 3)
    assert sorted(os.listdir(model / "render" / "train" / "iteration_3" / "gt" / "rgb")) == files


def test_strand_model_ply_resume_and_render_types(tmp_path):
    from tests.test_dataset_io_cpu import _write_capture
    from scene import Scene
    from scene.hair_gaussian_model import HairGaussianModel
    import render as render_cli
    src, model = tmp_path / "capture", tmp_path / "out"
    _write_capture(src, n_views=2, W=64, H=48)
    rng = np.random.default_rng(0)
    pts = np.cumsum(rng.normal(size=(12, 10, 3)) * 0.02, axis=1).astype(np.float32) + rng.normal(size=(12, 1, 3)).astype(np.float32) * 0.2
    hair = HairGaussianModel.from_strands(pts, sh_degree=0, device="cuda", ref_strand_root=pts[:, 0])
    os.makedirs(model / "point_cloud" / "iteration_7")
    hair.save_ply(str(model / "point_cloud" / "iteration_7" / "point_cloud.ply"))
    scene = Scene(_args(src, model), shuffle=False)
    assert isinstance(scene.gaussians, HairGaussianModel) and scene.loaded_iter == 7
    assert torch.allclose(scene.gaussians.get_xyz, hair.get_xyz)
    for kind in ("2", "3", "4"):
        render_cli.main(["-s", str(src), "-m", str(model), "--type", kind, "--quiet"])
    base = model / "render" / "train" / "iteration_7" / "renders"
    assert sorted(os.listdir(base)) == ["mask_foreground", "mask_other", "orientation_map"]
    assert np.asarray(PILImage.open(base / "orientation_map" / "00000.png")).shape == (48, 64, 3)
    assert np.asarray(PILImage.open(base / "mask_foreground" / "00000.png")).shape == (48, 64)


def test_train_cli_stage_one_then_resume(tmp_path):
    from tests.test_dataset_io_cpu import _write_capture
    import train as train_cli
    src, model = tmp_path / "capture", tmp_path / "out"
    _write_capture(src, n_views=3, W=64, H=48)
    scene = train_cli.main(["-s", str(src), "-m", str(model), "--iterations", "6", "--save_frequency", "4", "--quiet"])
    assert sorted(os.listdir(model / "point_cloud")) == ["iteration_4", "iteration_6"]
    assert os.path.exists(model / "cfg_args")
    scene2 = train_cli.main(["-s", str(src), "-m", str(model), "--iterations", "2", "--quiet"])
    assert scene2.loaded_iter == 6 and os.path.isdir(model / "point_cloud" / "iteration_8")
    # every invocation counts its iterations from 1 (reference train.py:91): the schedules restart, only the saved name adds up
    g = scene2.gaussians
    lr = [float(grp["lr"]) for grp in g.optimizer.param_groups if grp["name"] == g._POSITION_GROUP][0]
    assert abs(lr - g.xyz_scheduler_args(2)) <= 1e-12 + 1e-6 * lr and abs(lr - g.xyz_scheduler_args(8)) > 1e-9 * lr


def test_train_cli_on_two_ranks(tmp_path):
    """train.py under torchrun (2 ranks sharing the GPU through gloo; RCCL needs one GPU per rank): Stage-I training through
    densification and an opacity reset; rank 0 alone writes the model directory (cfg_args, input.ply, cameras.json, the
    checkpoints) and both ranks end with the same cloud, bit for bit."""
    import subprocess
    from tests.gpu_util import free_port
    from tests.test_dataset_io_cpu import _write_capture
    src, model = tmp_path / "capture", tmp_path / "out"
    _write_capture(src, n_views=4, W=64, H=48)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", HGS_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "_train_cli_worker.py"), str(src), str(model)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "TRAIN_CLI_RANK_0_OK" in out.stdout and "TRAIN_CLI_RANK_1_OK" in out.stdout
    assert sorted(os.listdir(model / "point_cloud")) == ["iteration_12", "iteration_24"]
    for f in ("cfg_args", "input.ply", "cameras.json"):
        assert os.path.exists(model / f)
    a, b = torch.load(model / "rank0.pt"), torch.load(model / "rank1.pt")
    from utils.ply import read_ply
    n_input = len(read_ply(str(model / "input.ply"))[0][1])
    assert a["xyz"].shape[0] > 0 and a["xyz"].shape[0] != n_input        # the densification did change the cloud
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_three_stage_pipeline_through_the_clis(tmp_path):
    """The reference's workflow end to end on a small capture: train.py (Stage I: Gaussian cloud) -> merge.py (Stage II: cloud ->
    strands, merged until nothing is left) -> train.py again on the strand model (Stage III: fused strand iteration, graph
    replays, topology operators at short intervals) -> render.py (all output types)."""
    from tests.test_dataset_io_cpu import _write_capture, _write_side_files
    import merge as merge_cli
    import render as render_cli
    import train as train_cli
    from scene.gaussian_model import GaussianModel
    from scene.hair_gaussian_model import HairGaussianModel
    src, model = tmp_path / "capture", tmp_path / "out"
    _write_capture(src, n_views=4, W=96, H=64)
    _write_side_files(src)                 # (scalp vertices: what Stage II / III orient the strands by)
    s1 = train_cli.main(["-s", str(src), "-m", str(model), "--iterations", "30", "--save_frequency", "30", "--quiet",
                         "--densify_from_iter", "5", "--densification_interval", "10", "--densify_grad_threshold", "1e-7"])
    assert isinstance(s1.gaussians, GaussianModel) and os.path.isdir(model / "point_cloud" / "iteration_30")
    # (hair_eval_data.npz: evaluated at the end; bidirectional_eval is the default: the reference's "(b)" keys)
    assert set(s1.eval_metrics) >= {"precision(b)", "recall(b)", "f1(b)"} and len(s1.eval_thresholds) == 4
    hair = merge_cli.main(["-s", str(src), "-m", str(model), "--iterations", "5"])
    assert isinstance(hair, HairGaussianModel) and hair.endpoint_pairs.shape[0] > 0   # (built under inference_mode: not a model to train on)
    saved = sorted(os.listdir(model / "point_cloud"), key=lambda d: int(d.split("_")[1]))
    assert len(saved) == 2 and int(saved[-1].split("_")[1]) > 30
    it2 = int(saved[-1].split("_")[1])
    s3 = train_cli.main(["-s", str(src), "-m", str(model), "--iterations", "24", "--save_frequency", "24", "--quiet",
                         "--densify_from_iter", "3", "--densification_interval", "6", "--merge_interval", "8",
                         "--opacity_reset_interval", "12"])
    assert isinstance(s3.gaussians, HairGaussianModel) and s3.loaded_iter == it2 and "f1(b)" in s3.eval_metrics
    assert os.path.isdir(model / "point_cloud" / f"iteration_{it2 + 24}")
    for p in (s3.gaussians._endpoints, s3.gaussians._opacity, s3.gaussians._features_dc):
        assert torch.isfinite(p).all()
    render_cli.main(["-s", str(src), "-m", str(model), "--quiet"])          # every type (-1)
    base = model / "render" / "train" / f"iteration_{it2 + 24}" / "renders"
    assert sorted(os.listdir(base)) == ["mask_foreground", "mask_other", "orientation_map", "rgb", "rgb_foreground"]
    assert len(os.listdir(base / "rgb")) == 4


def test_scene_reads_the_npz_side_files(tmp_path):
    """Scene (reference scene/__init__.py:103-122): hair_eval_data.npz -> scene.gt with unit directions;
    head_reconstruction_data.npz -> scene.head_reconstruction, its scalp vertices as the model's ref_strand_root (carried
    into the strand model by to_hair_gaussian_model, which cannot orient strands without them)."""
    from tests.test_dataset_io_cpu import _write_capture, _write_side_files
    from scene import Scene
    src, model = tmp_path / "capture", tmp_path / "out"
    _write_capture(src)
    args = _args(src, model)
    plain = Scene(args, shuffle=False)
    assert plain.gt is None and plain.head_reconstruction is None and plain.gaussians.ref_strand_root is None
    _write_side_files(src)
    scene = Scene(args, shuffle=False)
    assert scene.gt.points.shape == (42, 3) and np.allclose(np.linalg.norm(scene.gt.directions, axis=1), 1.0)
    assert scene.gt.points_id_to_strand_id.shape == (42,) and scene.gt_edges.shape == (41, 2)
    assert scene.head_reconstruction.scalp_verts.shape == (30, 3) and scene.head_reconstruction.head_verts.shape == (50, 3)
    assert np.array_equal(scene.gaussians.ref_strand_root, scene.head_reconstruction.scalp_verts)
    from arguments import OptimizationParams
    scene.gaussians.training_setup(OptimizationParams())
    hair = scene.gaussians.to_hair_gaussian_model()
    assert hair.strands_info.n_strands == hair.get_xyz.shape[0] > 0          # (every segment its own strand before merging)
    assert np.array_equal(hair.ref_strand_root, scene.head_reconstruction.scalp_verts)
