"""bench.py --gpus N starts its own N ranks.

CPU: the launcher command (`--dry-launch`), the loud failure on a node with fewer devices than ranks (this container has
none), the refusal of a torchrun world that is not the one asked for.  GPU: two ranks sharing the test box's one GPU
(HGS_BENCH_SHARE_GPU=1, gloo) through `python bench.py --gpus 2` itself -- the logic of the multi-rank line, not a measurement."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = dict(os.environ, **kw)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HGS_BENCH_SHARE_GPU"):
        if k not in kw:
            env.pop(k, None)
    return env


def test_dry_launch_prints_the_torchrun_command():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "7", "--warmup", "3", "--dry-launch"], env=_env(),
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    cmd = rec["launcher"]
    assert rec["gpus"] == 2
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=2" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(BENCH)
    assert cmd[i + 1:] == ["--gpus", "2", "--steps", "7", "--warmup", "3"]      # the caller's arguments, minus --dry-launch


@pytest.mark.skipif(__import__("torch").cuda.device_count() >= 2, reason="a node with two devices runs the real thing")
def test_fewer_devices_than_ranks_is_an_error_not_a_smaller_run():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1"], env=_env(),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "needs 2 visible GPUs" in out.stderr
    assert "n_gpus" not in out.stdout                                           # no line at all


def test_a_world_that_is_not_the_one_asked_for_is_refused():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "2"],
                         env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"),
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE=2" in out.stderr and out.stdout.strip() == ""
    out = subprocess.run([sys.executable, BENCH, "--steps", "2"],               # --gpus defaults to 1
                         env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"),
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and out.stdout.strip() == ""


@pytest.mark.gpu
def test_bench_gpus_2_starts_two_ranks_by_itself():
    """`python bench.py --gpus 2` with no torchrun around it: two ranks (sharing cuda:0 over gloo here), one JSON line from rank
    0 with n_gpus = n_ranks_seen = 2 and two views per optimizer step."""
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--workload", "tiny", "--steps", "8", "--warmup", "2",
                          "--repeats", "1", "--sustained-seconds", "0", "--trained-iters", "0", "--no-cpu-baseline",
                          "--no-kernel-timing"], env=_env(HGS_BENCH_SHARE_GPU="1"), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["n_ranks_seen"] == 2 and rec["ranks_share_one_gpu"] is True
    assert rec["config"]["views_per_optimizer_step"] == 2 and rec["value"] > 0
