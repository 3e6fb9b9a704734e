"""Host-side helpers of round 5 that need no GPU: the endpoint adjacency tables of the gather-mode backward, the degree
table the topology operators ask in place of torch.unique + torch.isin, the first-occurrence mask of
remove_duplicate_endpoint_rows, and the tile-order view of a capacity-mode binning buffer used by the parity tests."""
import numpy as np
import torch


def test_adjacency_lists_every_role_once():
    from hgs_runtime.strand_step import _adjacency
    pairs = torch.tensor([[0, 1], [1, 2], [2, 3], [5, 4]])          # a chain of three segments and a lone one
    table = _adjacency(pairs.reshape(-1), 2, 6, 2)
    assert table.shape == (6, 2) and table.dtype == torch.int32
    # code = 2 * segment + (which end): endpoint 1 is the second end of segment 0 and the first of segment 1
    assert sorted(table[1].tolist()) == [1, 2] and sorted(table[2].tolist()) == [3, 4]
    assert table[0].tolist() == [0, -1] and table[3].tolist() == [5, -1]
    assert table[5].tolist() == [6, -1] and table[4].tolist() == [7, -1]
    # an endpoint of degree three does not fit a two-slot table
    assert _adjacency(torch.tensor([0, 1, 0, 2, 0, 3]), 2, 4, 2) is None


def test_degree_table_answers_what_unique_and_isin_answered():
    from scene.hair_gaussian_model import HairGaussianModel
    rng = np.random.default_rng(0)
    pts = np.cumsum(rng.normal(size=(7, 6, 3)) * 0.01, axis=1).astype(np.float32)
    m = HairGaussianModel.from_strands(pts, device="cpu")
    m.endpoint_pairs = torch.cat([m.endpoint_pairs, torch.tensor([[3, 40], [41, 41]])])      # a branch and a degenerate segment
    deg = m._endpoint_degree_table()
    u, c = torch.unique(m.endpoint_pairs, return_counts=True)
    assert deg.shape[0] == int(m.endpoint_pairs.max()) + 1
    assert torch.equal(torch.nonzero(deg == 1).squeeze(1), u[c == 1])
    assert torch.equal(deg[m.endpoint_pairs] != 1, torch.isin(m.endpoint_pairs, u[c != 1]))
    assert torch.equal(deg[m.endpoint_pairs] == 1, torch.isin(m.endpoint_pairs, u[c == 1]))
    assert m._endpoint_degree_table() is deg                      # remembered while endpoint_pairs is the same tensor
    m.endpoint_pairs = m.endpoint_pairs[:-1]
    assert m._endpoint_degree_table() is not deg


def test_first_occurrence_mask_equals_the_reference_form():
    from scene.hair_gaussian_model import HairGaussianModel
    m = HairGaussianModel(sh_degree=0, device="cpu")
    rng = np.random.default_rng(1)
    for n in (1, 7, 200):
        rows = torch.as_tensor(rng.integers(0, max(3, n // 2), (n, 2)))
        kept, mask = m.remove_duplicate_endpoint_rows(rows, return_mask=True)
        flat = rows.flatten()
        want = torch.zeros(flat.shape[0], dtype=torch.bool)
        want[m.get_first_occurence_index(flat)] = True                 # the reference's statements (:712-728, :772-784)
        want = want.reshape(-1, 2)
        want = want[:, 0] & want[:, 1]
        assert torch.equal(mask, want) and torch.equal(kept, rows[want])
    empty, mask = m.remove_duplicate_endpoint_rows(torch.zeros((0, 2), dtype=torch.long), return_mask=True)
    assert empty.shape == (0, 2) and mask.shape == (0,)


def test_in_tile_order_relays_an_allocated_layout():
    import importlib.util, os, sys
    # (tests.gpu_util imports the GPU binding at module level; the helper itself is numpy: load it without that import)
    src = open(os.path.join(os.path.dirname(__file__), "gpu_util.py")).read()
    body = src[src.index("def in_tile_order"):src.index("def run_backward")]
    ns = {"np": np}
    exec(body, ns)
    in_tile_order = ns["in_tile_order"]
    # four tiles; the allocation put tile 2's list first, then tile 0's, then tile 3's; tile 1 is empty
    ranges = np.array([[3, 5], [0, 0], [0, 3], [5, 9]], np.uint32)
    point_list = np.array([20, 21, 22, 0, 1, 30, 31, 32, 33], np.uint32)
    keys = point_list.astype(np.uint64) + 100
    r, pl, ks = in_tile_order(dict(ranges=ranges, point_list=point_list, keys_sorted=keys), 9)
    assert r.tolist() == [[0, 2], [0, 0], [2, 5], [5, 9]]
    assert pl.tolist() == [0, 1, 20, 21, 22, 30, 31, 32, 33] and ks.tolist() == (pl.astype(np.uint64) + 100).tolist()
    import pytest
    with pytest.raises(AssertionError):                              # overlapping segments are not a partition
        in_tile_order(dict(ranges=np.array([[0, 3], [2, 5]], np.uint32), point_list=point_list[:5], keys_sorted=keys[:5]), 5)


def test_bucket_capacity_keeps_four_significant_bits():
    """diff_gaussian_rasterization._C.bucket_capacity (round 6): >= n, at most 12.5 % above it, monotone, idempotent -- successive
    captures of a slowly growing model then ask the allocator for the same workspace sizes."""
    from diff_gaussian_rasterization._C import bucket_capacity as b
    prev = 0
    for n in list(range(0, 70)) + [100, 4095, 4096, 4097, 173000, 1 << 20, (1 << 20) + 1, 2437617, (1 << 31) - 5]:
        c = b(n)
        assert c >= n and c <= n + max(n // 8, 1) and b(c) == c and c >= prev
        assert n <= 16 or bin(c).rstrip("0").count("1") <= 4 and len(bin(c).rstrip("0")) - 2 <= 4
        prev = c
    assert len({b(n) for n in range(100000, 112000)}) <= 2


def test_training_event_log_and_visible_gpus_need_no_gpu():
    """train._EventInfo is what the operators take as `training_info`; bench.visible_gpus() counts devices without a HIP call."""
    import importlib.util
    import os
    from train import _EventInfo
    info = _EventInfo()
    info.densification_info["clone"] = 3
    assert info.densification_info == {"clone": 3}
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    n = m.visible_gpus()
    assert n is None or n >= 0
    if not os.path.isdir("/sys/class/kfd"):
        assert n == 0


def test_greedy_partner_filter_equals_the_in_order_walk():
    """HairTopologyMixin._keep_rows_whose_strand_partners_are_free (stage 2 of compute_endpoint_pair_to_merge resolved in vectorised
    rounds) against the reference's walk (scene/hair_gaussian_model.py:1331-1362: a set of blocked ids, rows in order) on random
    tables: strands of two ends each, candidate rows pairing ends of different strands, every id in at most one row, long
    dependency chains included (row k pairs the far end of strand k with the near end of strand k + 1)."""
    from scene.hair_topology import HairTopologyMixin
    rng = np.random.default_rng(5)
    for trial in range(200):
        n_strands = int(rng.integers(2, 400))
        n_ids = 2 * n_strands + int(rng.integers(0, 5))
        ids = rng.permutation(n_ids)[:2 * n_strands]
        partner = -np.ones(n_ids, np.int32)
        partner[ids[0::2]], partner[ids[1::2]] = ids[1::2], ids[0::2]
        if trial % 3 == 0:      # a chain: every row depends on the one in front of it
            cand = np.stack([ids[1:-1:2], ids[2::2]], 1)
        else:
            free = rng.permutation(ids)
            k = int(rng.integers(1, n_strands + 1))
            cand = free[:2 * k].reshape(k, 2)
            cand = cand[partner[cand[:, 0]] != cand[:, 1]]          # (the two ends of one strand are never a candidate row)
        if trial % 2:
            cand = cand[rng.permutation(cand.shape[0])]
        cand = cand.astype(np.int64)
        blocked, keep = set(), []
        for (p, q) in cand.tolist():
            if p in blocked or q in blocked:
                keep.append(False)
                continue
            blocked.add(int(partner[p])); blocked.add(int(partner[q])); keep.append(True)
        got = HairTopologyMixin._keep_rows_whose_strand_partners_are_free(cand, partner)
        np.testing.assert_array_equal(got, np.asarray(keep, bool), err_msg=f"trial {trial}")
