"""Worker of tests/test_gpu_view_parallel.py::test_step_graph_with_the_exchange_inside (own process: it owns a process group).
RCCL all-reduce inside a captured HIP graph, as far as ONE GPU can show it: a 1-rank nccl (= RCCL) process group, the
probe of train.ViewParallel.graph_collective_ok(), and the multi-rank code path of GraphedStep (pack -> all-reduce -> Adam
inside the step graph, several optimizer steps per launch) driven with the world size faked to 2 -- the collective itself
then reduces over one rank, i.e. is the identity, so the run must equal the plain single-rank run bit for bit."""
import os, sys
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from arguments import OptimizationParams
from diff_gaussian_rasterization import _C as raster
from synthetic import build_workload
from train import GraphedStep, ViewParallel
from utils.general import safe_state

vp = ViewParallel()
vp.world = 2                                   # (fake: exercises the multi-rank branches; the reduction is over 1 rank)
print("probe says the all-reduce can be captured:", vp.graph_collective_ok())
order = [1, 3, 0, 2, 1, 0, 3, 2, 3, 1]
res = {}
for mode in ("single_rank", "in_graph_collective"):
    safe_state(True)
    model, cams, extent = build_workload("tiny", device="cuda", with_targets=True)
    opt = OptimizationParams(); opt.enable_topology = False
    model.training_setup(opt)
    bg = torch.zeros(3, device="cuda")
    v = ViewParallel(enabled=False) if mode == "single_rank" else vp
    gs = GraphedStep(model, cams, opt, bg, extent=extent, vp=v, steps_per_graph=4)
    gs.capture(cams)
    if mode != "single_rank":
        print("collective captured:", gs.collective_captured, "steps per graph:", gs.steps_per_graph)
    gs.step_many([cams[i] for i in order[:4]], 1)
    gs.step_many([cams[i] for i in order[4:8]], 5)
    gs.step(cams[order[8]], 9)
    gs.step(cams[order[9]], 10)
    torch.cuda.synchronize()
    gs.check()
    raster.set_async(False)
    res[mode] = [g["params"][0].detach().clone() for g in model.optimizer.param_groups]
same = all(torch.equal(a, b) for a, b in zip(res["single_rank"], res["in_graph_collective"]))
print("10 optimizer steps through the in-graph exchange == single rank, bit for bit:", same)
if same:
    print("RCCL_GRAPH_OK")
dist.destroy_process_group()
sys.exit(0 if same else 1)
