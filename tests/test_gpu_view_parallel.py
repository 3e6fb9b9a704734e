"""View-parallel training step on the GPU: two ranks share the one GPU of the test box through the gloo backend (RCCL
needs one GPU per rank), which exercises everything but the transport: per-rank views, the captured pack of the
gradients into the flat exchange buffer, the in-place averaging all-reduce between the two graphs, replicated Adam."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_two_ranks_stay_replicated_and_match_hand_averaged_gradients():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "tests", "_vp_gpu_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "VP_GPU_OK" in out.stdout
