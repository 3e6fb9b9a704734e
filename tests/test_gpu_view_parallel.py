"""View-parallel training step on the GPU: the ranks share the one GPU of the test box through the gloo backend (RCCL
needs one GPU per rank), which exercises everything but the transport: per-rank views, the captured pack of the
gradients into the flat exchange buffer, the in-place all-reduce between the two graphs, replicated Adam -- in the weak
mode (one view per rank and step) and in the strong mode of SURVEY.md 8e (a fixed global batch of views per optimizer
step shared by the ranks, several views per rank inside one captured graph)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _run(nproc, port=None, mode="weak", global_views=None, backend="gloo", steps_per_graph=1):
    from tests.gpu_util import free_port
    port = free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", HGS_VP_MODE=mode, HGS_VP_BACKEND=backend,
               HGS_VP_STEPS_PER_GRAPH=str(steps_per_graph))
    if global_views is not None:
        env["HGS_VP_GLOBAL_VIEWS"] = str(global_views)
    worker = os.path.join(ROOT, "tests", "_vp_gpu_worker.py")
    if nproc == 1:
        env.pop("WORLD_SIZE", None)
        cmd = [sys.executable, worker]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
               "127.0.0.1", "--master-port", str(port), worker]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "VP_GPU_OK" in out.stdout
    if backend == "nccl" and nproc > 1:
        assert "VP_RCCL_GRAD_OK" in out.stdout
    return out.stdout


def _devices():
    import torch
    return torch.cuda.device_count()      # (does not initialise the GPU: the workers are started from a process that never does)


# ---- RCCL between DEVICES: these switch themselves on where the box has more than one GPU (the driver's 8-GPU node); on the
# one-GPU lease they are skipped and the gloo tests below cover everything but the transport.
@pytest.mark.skipif(_devices() < 2, reason="needs 2 GPUs: one rank per device over RCCL")
@pytest.mark.parametrize("steps_per_graph", [1, 4])
def test_rccl_two_devices_weak(steps_per_graph):
    """One view per rank and step on two devices: replicas bit-identical, parameters equal to a single process that averages
    the same views' gradients by hand, the all-reduce captured into the step's graph (four optimizer steps per launch in the
    second case), one averaged exchange within 1e-5 of the hand mean."""
    _run(2, mode="weak", backend="nccl", steps_per_graph=steps_per_graph)


@pytest.mark.skipif(_devices() < 2, reason="needs 2 GPUs: one rank per device over RCCL")
def test_rccl_two_devices_strong():
    """SURVEY.md 8e's protocol on two devices: a global batch of 4 views per optimizer step, two per rank inside one captured
    graph, summed locally, one all-reduce."""
    _run(2, mode="strong", global_views=4, backend="nccl")


@pytest.mark.skipif(_devices() < 8, reason="needs 8 GPUs")
@pytest.mark.parametrize("mode,global_views", [("weak", None), ("strong", 8)])
def test_rccl_eight_devices(mode, global_views):
    _run(8, mode=mode, global_views=global_views, backend="nccl", steps_per_graph=4 if mode == "weak" else 1)


def test_two_ranks_stay_replicated_and_match_hand_averaged_gradients():
    _run(2, 29533)


def test_four_ranks_stay_replicated_and_match_hand_averaged_gradients():
    _run(4, 29534)


@pytest.mark.parametrize("nproc", [1, 2, 4])
def test_strong_mode_equals_single_process_gradient_accumulation(nproc):
    """A global batch of 4 views per optimizer step on 1, 2 and 4 ranks (4, 2, 1 views per rank): the same parameters as
    one process that accumulates the four views' gradients by hand and takes their mean."""
    _run(nproc, 29535 + nproc, mode="strong", global_views=4)


def test_step_graph_with_the_exchange_inside():
    """The RCCL form of the step -- pack, all-reduce and Adam captured INTO the iteration's graph, several optimizer steps
    per launch across ranks (train.GraphedStep with ViewParallel.graph_collective_ok()) -- as far as one GPU can exercise it:
    a 1-rank nccl (= RCCL) group, the world size faked to 2 for the code path, the reduction itself over one rank.  Ten
    optimizer steps (two 4-step launches + two single steps) equal the single-rank run bit for bit.  What one GPU cannot
    show is the transport: DESIGN.md section 7."""
    from tests.gpu_util import free_port
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_graph_worker.py")], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "RCCL_GRAPH_OK" in out.stdout and "collective captured: True steps per graph: 4" in out.stdout


def test_two_ranks_through_the_topology_operators():
    """training() on two ranks (one view per rank and step, graph replays) THROUGH densification, merging and the opacity reset:
    every rank ends with the same segments, endpoints, strands, parameters and Adam moments, bit for bit (the
    operators' statistics are reduced over the ranks, their random draws are seeded alike)."""
    from tests.gpu_util import free_port
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    worker = os.path.join(ROOT, "tests", "_vp_topology_worker.py")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), worker]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "VP_TOPOLOGY_OK" in out.stdout


def test_two_ranks_roll_a_capacity_overflow_back_together():
    """training() on two ranks with a captured binning capacity that is too small: the headroom check takes the maximum over the
    ranks, both return to the same checkpoint, re-capture and run the iterations again -- and end like a run that never
    overflowed, bit for bit, identically on both ranks."""
    from tests.gpu_util import free_port
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    worker = os.path.join(ROOT, "tests", "_vp_rollback_worker.py")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), worker]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "VP_ROLLBACK_OK" in out.stdout

