"""View-parallel gradient / statistics exchange on 2 CPU processes (gloo)."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    for p in (ROOT, os.path.join(ROOT, "hair-gs_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from arguments import OptimizationParams
    from scene.hair_gaussian_model import HairGaussianModel
    from synthetic import strand_polylines
    from train import ViewParallel, ViewSampler
    m = HairGaussianModel.from_strands(strand_polylines(4, 5, seed=0), device="cpu")
    m.training_setup(OptimizationParams())
    vp = ViewParallel()
    assert vp.world == world and vp.rank == rank
    # each rank has a different "view": a different scalar weight on the same differentiable expression
    w = float(rank + 1)
    loss = w * (m.get_xyz ** 2).sum() + w * m.get_opacity.sum() + w * (m._width ** 2).sum() + w * m._features_dc.sum() \
        + 0.0 * m._features_rest.sum() + w * m._mask.sum()
    loss.backward()
    m.xyz_gradient_accum += w
    m.denom += 1
    m.max_radii2D += w
    vp.reduce_stats(m)
    vp.reduce_gradients(m)
    m.optimizer.step()
    picks = [ViewSampler(list(range(10)), seed=3, rank=r, world=world).next() for r in range(world)]
    out[rank] = dict(ep=m._endpoints.detach().clone(), grad_op=m._opacity.grad.clone(), accum=m.xyz_gradient_accum.clone(),
                     radii=m.max_radii2D.clone(), picks=picks)
    dist.destroy_process_group()


def test_two_rank_gradient_and_stats_exchange():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    a, b = out[0], out[1]
    assert torch.equal(a["ep"], b["ep"])                       # replicas stay bit-identical after Adam
    assert torch.equal(a["grad_op"], b["grad_op"])
    assert torch.allclose(a["accum"], torch.full_like(a["accum"], 3.0))      # SUM of 1 and 2
    assert torch.allclose(a["radii"], torch.full_like(a["radii"], 2.0))      # MAX
    assert a["picks"] == b["picks"] and a["picks"][0] != a["picks"][1]       # same shuffle, different views
    # averaged gradient == single-process gradient accumulation over both "views" / 2
    for p in (ROOT, os.path.join(ROOT, "hair-gs_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from scene.hair_gaussian_model import HairGaussianModel
    from synthetic import strand_polylines
    m = HairGaussianModel.from_strands(strand_polylines(4, 5, seed=0), device="cpu")
    tot = sum(w * m.get_opacity.sum() for w in (1.0, 2.0)) / 2
    tot.backward()
    assert torch.allclose(a["grad_op"], m._opacity.grad, rtol=1e-6)


def _worker_strong(rank, world, port, out):
    """Strong-scaling protocol on CPU tensors: a global batch of 4 'views' per step, 2 per rank, gradients summed locally,
    SUM all-reduced, scaled to the mean over the batch (what GraphedStep(views_per_step=2) captures on the GPU)."""
    for p in (ROOT, os.path.join(ROOT, "hair-gs_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from arguments import OptimizationParams
    from scene.hair_gaussian_model import HairGaussianModel
    from synthetic import strand_polylines
    from train import ViewParallel, ViewSampler
    m = HairGaussianModel.from_strands(strand_polylines(4, 5, seed=0), device="cpu")
    m.training_setup(OptimizationParams())
    vp = ViewParallel()
    sampler = ViewSampler(list(range(8)), seed=5, rank=rank, world=world)
    mine = sampler.next_batch(4)
    assert len(mine) == 2
    for view in mine:                       # local accumulation over this rank's views (AccumulateGrad)
        w = float(view + 1)
        (w * (m.get_xyz ** 2).sum() + w * m.get_opacity.sum()).backward()
    vp.pack_gradients(m)
    vp.exchange(average=False)
    torch._foreach_mul_([p.grad for p in vp.params(m) if p.grad is not None], 1.0 / 4)
    out[rank] = dict(mine=mine, grad_op=m._opacity.grad.clone(), grad_ep=m._endpoints.grad.clone())
    dist.destroy_process_group()


def test_strong_mode_global_batch_is_partitioned_and_averaged():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_strong, args=(world, port, out), nprocs=world, join=True)
    a, b = out[0], out[1]
    batch = sorted(a["mine"] + b["mine"])
    assert len(set(batch)) == 4 and not set(a["mine"]) & set(b["mine"])      # one draw, disjoint shares
    assert torch.equal(a["grad_op"], b["grad_op"]) and torch.equal(a["grad_ep"], b["grad_ep"])
    for p in (ROOT, os.path.join(ROOT, "hair-gs_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from scene.hair_gaussian_model import HairGaussianModel
    from synthetic import strand_polylines
    from train import ViewSampler
    assert sorted(ViewSampler(list(range(8)), seed=5).next_batch(4)) == batch   # the 1-rank run draws the same batch
    m = HairGaussianModel.from_strands(strand_polylines(4, 5, seed=0), device="cpu")
    tot = sum(float(v + 1) * ((m.get_xyz ** 2).sum() + m.get_opacity.sum()) for v in batch) / 4
    tot.backward()
    assert torch.allclose(a["grad_op"], m._opacity.grad, rtol=1e-6)
    assert torch.allclose(a["grad_ep"], m._endpoints.grad, rtol=1e-5, atol=1e-9)
