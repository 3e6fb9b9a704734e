"""Stage-III strand model (counterpart of the reference's scene/hair_gaussian_model.py:44-1515).

A Gaussian k is a line segment between two SHARED endpoints (e0, e1) = _endpoints[endpoint_pairs[k]] plus a width:
    mean     = (e0 + e1) / 2                                                    (reference :167-172)
    scale    = (max(|e1-e0|/2 * dist_to_scale_factor, 1e-7), exp(w), exp(w))    (:134-145)
    rotation = quaternion of the rotation x_hat -> (e1-e0), identity if collapsed (:147-165)
Gradients flow through all three back to the shared endpoints (scatter-add of per-segment gradients).

This file holds the rasterizer-facing surface + optimizer plumbing + strand bookkeeping; the topology
operators (split / clone / merge / grow) live in scene/hair_topology.py.
"""

import os

import numpy as np
import torch
from torch import nn

from scene.gaussian_model import GaussianModel
from scene.hair_topology import HairTopologyMixin
from utils.general import get_expon_lr_func, inverse_sigmoid
from utils.transform import calculate_rotation_from_vectors


def walk_chains(pairs, n_ep, id_to_strand, complementary, end_distance):
    """Order every open polyline of the edge table `pairs` ([n, 2] endpoint ids; every id has degree 1 or 2) from one end
    to the other, without walking it edge by edge in Python (a model in training holds 10^5-10^6 segments and this runs at
    every densification / merge): pointer doubling over the 2 n directed arcs.  Arc (u -> v) is followed by the arc that
    leaves v through v's other edge; after ceil(log2(longest chain)) doublings every arc knows the chain end it runs
    into and how many edges lie between.  The chain starts at its end with the SMALLER id (what an id-ordered walk over
    the ends visits first), strands are numbered in the order of their starts, and a strand is reversed when its start is
    farther from the reference roots than its other end (`end_distance(ids) -> distances`); reference
    scene/hair_gaussian_model.py:1410-1498.  Fills id_to_strand / complementary in place and returns the strands flat:
    offsets [S+1], rows [sum n_seg, 2] (cur, next) ids, segment rows [sum n_seg] (rows of `pairs`).  Closed loops have no
    end and are left out, like in the walk."""
    n = pairs.shape[0]
    pairs = np.ascontiguousarray(pairs, dtype=np.int64)
    flat = pairs.reshape(-1)
    # incidence table: endpoint id -> its (at most two) rows
    order = np.argsort(flat, kind="stable")
    ids_sorted = flat[order]
    first = np.r_[True, ids_sorted[1:] != ids_sorted[:-1]]
    inc = -np.ones((n_ep, 2), np.int64)
    inc[ids_sorted[first], 0] = order[first] // 2
    inc[ids_sorted[~first], 1] = order[~first] // 2
    # arcs: 2 r = pairs[r, 0] -> pairs[r, 1], 2 r + 1 the reverse
    rows = np.repeat(np.arange(n, dtype=np.int64), 2)
    head = np.stack([pairs[:, 1], pairs[:, 0]], axis=1).reshape(-1)          # node the arc arrives at
    r0, r1 = inc[head, 0], inc[head, 1]
    nrow = np.where(r0 != rows, r0, r1)                                        # the other edge at the head (-1: chain end)
    terminal = nrow < 0
    nr = np.where(terminal, 0, nrow)
    succ = np.where(terminal, np.arange(2 * n), 2 * nr + (pairs[nr, 0] != head))
    dist = np.where(terminal, 0, 1).astype(np.int64)
    reached = terminal.copy()
    for _ in range(max(1, int(np.ceil(np.log2(max(n, 2)))) + 1)):
        if reached.all():
            break
        dist = dist + np.where(reached, 0, dist[succ])
        reached_next = reached | reached[succ]
        succ = np.where(reached, succ, succ[succ])
        reached = reached_next
    end_of = head[succ]                                                       # chain end in the arc's direction
    ok = reached[0::2] & reached[1::2]                                        # (arcs of closed loops never arrive)
    e_fwd, e_bwd = end_of[0::2], end_of[1::2]                                 # ends beyond pairs[:,1] / beyond pairs[:,0]
    start = np.minimum(e_fwd, e_bwd)
    from_bwd = e_bwd == start                                                 # walking from the start meets pairs[:,0] first
    pos = np.where(from_bwd, dist[1::2], dist[0::2])                          # edges between the start and this edge
    cur = np.where(from_bwd, pairs[:, 0], pairs[:, 1])
    nxt = np.where(from_bwd, pairs[:, 1], pairs[:, 0])
    other = np.maximum(e_fwd, e_bwd)
    sel = np.nonzero(ok)[0]
    if sel.size == 0:
        return np.zeros(1, np.int64), np.zeros((0, 2), np.int64), np.zeros(0, np.int64)
    key = np.lexsort((pos[sel], start[sel]))
    sel = sel[key]
    st = start[sel]
    bounds = np.r_[0, np.nonzero(st[1:] != st[:-1])[0] + 1, st.size]
    starts, others = st[bounds[:-1]], other[sel][bounds[:-1]]
    sid = np.repeat(np.arange(starts.size), np.diff(bounds))
    id_to_strand[cur[sel]] = sid
    id_to_strand[nxt[sel]] = sid
    complementary[starts], complementary[others] = others, starts
    d = np.asarray(end_distance(np.concatenate([starts, others])))
    flip = d[:starts.size] > d[starts.size:]                                  # the far end was first: reverse
    # reversed strands: reverse the order of their edges and swap (cur, next), in bulk; the per-strand arrays are then
    # views into the two flat arrays (models hold 10^5 strands: no per-strand copies)
    lens = np.diff(bounds)
    flip_e = np.repeat(flip, lens)
    j = np.arange(sel.size)
    src = np.where(flip_e, np.repeat(bounds[:-1] + bounds[1:] - 1, lens) - j, j)
    c, x = cur[sel][src], nxt[sel][src]
    seq = np.stack([np.where(flip_e, x, c), np.where(flip_e, c, x)], axis=1)
    rows_sorted = sel[src]
    return bounds.astype(np.int64), seq, rows_sorted


def walk_chains_torch(pairs, n_ep, end_distance, max_len_hint=None):
    """walk_chains on torch tensors (the model's device: the doubling steps are gathers over 2n arcs, a few microseconds
    each on the GPU against milliseconds in numpy).  pairs: [n, 2] int64 tensor; end_distance(ids tensor) -> distances.
    Returns (offsets, rows, segment rows, id_to_strand int32 [n_ep], complementary int32 [n_ep]) as tensors on the device;
    same strands, order and orientation as walk_chains.
    max_len_hint: the caller's guess of the longest strand (segments).  The pointer doubling needs log2(longest strand) steps, not
    log2(n): with a guess the loop runs that many (5 launches per step, ~20 steps for 3 10^5 segments against ~10), and if any
    arc has not reached its end by then -- the guess was short, or the arc lies on a cycle -- the walk is repeated in full."""
    dev = pairs.device
    n = pairs.shape[0]
    i64 = dict(dtype=torch.int64, device=dev)
    id_to_strand = torch.full((n_ep,), -1, dtype=torch.int32, device=dev)
    complementary = torch.full((n_ep,), -1, dtype=torch.int32, device=dev)
    pairs = pairs.to(torch.int64).contiguous()
    flat = pairs.reshape(-1)
    order = torch.argsort(flat, stable=True)
    ids_sorted = flat[order]
    first = torch.ones(flat.shape[0], dtype=torch.bool, device=dev)
    first[1:] = ids_sorted[1:] != ids_sorted[:-1]
    inc = torch.full((n_ep, 2), -1, **i64)
    inc[ids_sorted[first], 0] = order[first] // 2
    inc[ids_sorted[~first], 1] = order[~first] // 2
    rows = torch.arange(n, **i64).repeat_interleave(2)
    head = torch.stack([pairs[:, 1], pairs[:, 0]], dim=1).reshape(-1)
    r0, r1 = inc[head, 0], inc[head, 1]
    nrow = torch.where(r0 != rows, r0, r1)
    terminal = nrow < 0
    nr = torch.where(terminal, torch.zeros_like(nrow), nrow)
    succ = torch.where(terminal, torch.arange(2 * n, **i64), 2 * nr + (pairs[nr, 0] != head).to(torch.int64))
    dist = (~terminal).to(torch.int64)
    reached = terminal.clone()
    full_steps = max(1, int(np.ceil(np.log2(max(n, 2)))) + 1)
    steps = full_steps if max_len_hint is None else min(full_steps, max(1, int(np.ceil(np.log2(max(int(max_len_hint), 2)))) + 1))
    for _ in range(steps):   # fixed count: no host round trip per step
        dist = dist + torch.where(reached, torch.zeros_like(dist), dist[succ])
        reached_next = reached | reached[succ]
        succ = torch.where(reached, succ, succ[succ])
        reached = reached_next
    end_of = head[succ]
    ok = reached[0::2] & reached[1::2]
    if steps < full_steps and int(ok.sum()) != n:      # (the read-back below synchronises anyway)
        return walk_chains_torch(pairs, n_ep, end_distance, None)
    e_fwd, e_bwd = end_of[0::2], end_of[1::2]
    start = torch.minimum(e_fwd, e_bwd)
    from_bwd = e_bwd == start
    pos = torch.where(from_bwd, dist[1::2], dist[0::2])
    cur = torch.where(from_bwd, pairs[:, 0], pairs[:, 1])
    nxt = torch.where(from_bwd, pairs[:, 1], pairs[:, 0])
    other = torch.maximum(e_fwd, e_bwd)
    sel = torch.nonzero(ok).squeeze(1)
    if sel.numel() == 0:
        return (torch.zeros(1, **i64), torch.zeros((0, 2), **i64), torch.zeros(0, **i64), id_to_strand, complementary)
    key = start[sel] * (int(n) + 1) + pos[sel]                                # (start, position) in one key: unique
    sel = sel[torch.argsort(key)]
    st = start[sel]
    cut = torch.nonzero(st[1:] != st[:-1]).squeeze(1) + 1
    bounds = torch.cat([torch.zeros(1, **i64), cut, torch.tensor([st.numel()], **i64)])
    lens = bounds[1:] - bounds[:-1]
    starts, others = st[bounds[:-1]], other[sel][bounds[:-1]]
    sid = torch.arange(starts.numel(), device=dev, dtype=torch.int32).repeat_interleave(lens)
    id_to_strand[cur[sel]] = sid
    id_to_strand[nxt[sel]] = sid
    complementary[starts] = others.to(torch.int32)
    complementary[others] = starts.to(torch.int32)
    d = end_distance(torch.cat([starts, others]))
    flip = d[:starts.numel()] > d[starts.numel():]
    flip_e = flip.repeat_interleave(lens)
    j = torch.arange(sel.numel(), **i64)
    src = torch.where(flip_e, (bounds[:-1] + bounds[1:] - 1).repeat_interleave(lens) - j, j)
    c, x = cur[sel][src], nxt[sel][src]
    seq = torch.stack([torch.where(flip_e, x, c), torch.where(flip_e, c, x)], dim=1)
    return bounds, seq, sel[src], id_to_strand, complementary


def walk_chains_device(pairs, n_ep, end_distance):
    """walk_chains for a segment table on the GPU, by the library's strand walk (include/hgs.h hgs_strand_walk_ends / _fill: one
    lane per strand end follows its chain through a node table -- two launches and a handful of index operations instead of the
    ~120 launches of walk_chains_torch's pointer doubling; 2.7 -> 1 ms on a 4 10^5-segment model, twice per topology event).
    Same returns as walk_chains_torch (same strands, numbering, order and orientation: tests/test_gpu_train.py compares all three
    forms); None if the table is not a set of chains (an endpoint of degree > 2), for the caller to fall back."""
    import hgs_runtime as rt
    dev, n = pairs.device, int(pairs.shape[0])
    pairs = rt.require_gpu_tensor(pairs, "endpoint pairs", torch.int64)
    i32, i64 = dict(dtype=torch.int32, device=dev), dict(dtype=torch.int64, device=dev)
    deg, nodes = torch.empty(n_ep, **i32), torch.empty((n_ep, 4), **i32)
    other, length, flags = torch.empty(n_ep, **i32), torch.empty(n_ep, **i32), torch.empty(1, **i32)
    id_to_strand = torch.full((n_ep,), -1, **i32)
    L = rt.lib()
    with torch.cuda.device(dev):
        rt.check(L.hgs_strand_walk_ends(rt.current_stream(), n, int(n_ep), rt.ptr(pairs), rt.ptr(deg), rt.ptr(nodes), rt.ptr(other),
                                        rt.ptr(length), rt.ptr(flags)))
        # a strand is stored from its end with the SMALLER id, strands in ascending order of that id (walk_chains)
        starts = torch.nonzero(other > torch.arange(n_ep, **i32)).squeeze(1)
        if int(flags.item()) != 0:
            return None
        S = int(starts.numel())
        if S == 0:
            return torch.zeros(1, **i64), torch.zeros((0, 2), **i64), torch.zeros(0, **i64), id_to_strand, other
        others = other[starts].to(torch.int64)
        offsets = torch.zeros(S + 1, **i64)
        torch.cumsum(length[starts], dim=0, out=offsets[1:])
        d = end_distance(torch.cat([starts, others]))
        flip = (d[:S] > d[S:]).to(torch.uint8)          # the far end was first: stored from the other one
        rows, seg_rows = torch.empty((n, 2), **i64), torch.empty(n, **i64)
        rt.check(L.hgs_strand_walk_fill(rt.current_stream(), S, rt.ptr(starts), rt.ptr(offsets), rt.ptr(flip), rt.ptr(nodes),
                                        rt.ptr(rows), rt.ptr(seg_rows), rt.ptr(id_to_strand)))
        total = int(offsets[-1].item())                 # (segments on closed loops are on no strand)
    return offsets, rows[:total], seg_rows[:total], id_to_strand, other


def nearest_distance(points, refs, chunk=65536):
    """Distance of every point to its nearest reference point, float64, brute force in chunks (what the reference asks a
    scipy cKDTree for, :1466-1470; a few thousand roots against 10^5-10^6 strand ends is milliseconds on the GPU).
    torch.cdist in its difference form (no matrix-multiply expansion: exact to rounding like the tree's distances)."""
    refs = refs.to(torch.float64)
    if points.is_cuda and points.dtype == torch.float32 and refs.shape[0] > 0:
        # hgs_nearest_distance_f64: one lane per point, the references through LDS (torch.cdist's float64 kernel needs 8 ms for
        # 6 10^5 ends against 10^3 roots: it was a third of the GPU time of a run with the topology operators)
        import hgs_runtime as rt
        pts = points.contiguous()
        refs_c = refs.to(points.device).contiguous()
        out = torch.empty(pts.shape[0], dtype=torch.float64, device=pts.device)
        with torch.cuda.device(pts.device):
            rt.check(rt.lib().hgs_nearest_distance_f64(rt.current_stream(), pts.shape[0], refs_c.shape[0], rt.ptr(pts), rt.ptr(refs_c),
                                                       rt.ptr(out)))
        return out
    out = torch.empty(points.shape[0], dtype=torch.float64, device=points.device)
    for s in range(0, points.shape[0], chunk):
        p = points[s:s + chunk].to(torch.float64)
        out[s:s + chunk] = torch.cdist(p, refs, compute_mode="donot_use_mm_for_euclid_dist").min(dim=1).values
    return out


class StrandsInfo:
    """Strands of the model (reference scene/hair_gaussian_model.py:1410-1498, same attribute names).  Held flat --
    `flat` = (offsets [S+1], rows [sum n_seg, 2] (cur, next) endpoint ids, root -> tip, segment rows [sum n_seg]) --;
    the reference's per-strand object arrays are built on first access, as views into the flat arrays (a model in training
    holds 10^5 strands and re-derives them at every densification / merge, mostly without looking at single strands)."""

    def __init__(self, offsets, rows, segment_rows, id_to_strand_id, strand_endpoint_id_to_complementary):
        self.offsets, self.rows, self.segment_rows = offsets, rows, segment_rows
        self.id_to_strand_id = id_to_strand_id                      # endpoint id -> strand id (-1 if none)
        self.strand_endpoint_id_to_complementary = strand_endpoint_id_to_complementary  # strand end -> its other end
        self._lists = None

    @property
    def flat(self):
        return self.offsets, self.rows

    @property
    def n_strands(self):
        return len(self.offsets) - 1

    def _materialise(self):
        if self._lists is None:
            n = self.n_strands
            ls, lr = np.empty(n, dtype=object), np.empty(n, dtype=object)
            o = self.offsets
            for i in range(n):
                ls[i], lr[i] = self.rows[o[i]:o[i + 1]], self.segment_rows[o[i]:o[i + 1]]
            self._lists = (ls, lr)
        return self._lists

    @property
    def list_strands(self):              # object array; each [n_seg, 2] endpoint ids, root -> tip
        return self._materialise()[0]

    @property
    def list_strands_segments_id(self):  # object array; each [n_seg] rows of (foreground-filtered) endpoint_pairs
        return self._materialise()[1]


class HairGaussianModel(HairTopologyMixin, GaussianModel):
    _PARAM_ATTRS = (("endpoints", "_endpoints"), ("f_dc", "_features_dc"), ("f_rest", "_features_rest"),
                    ("opacity", "_opacity"), ("mask", "_mask"), ("width", "_width"))
    _POSITION_GROUP = "endpoints"

    def storage_order(self, bits=10):
        """(segment permutation, endpoint permutation) that stores the model strand by strand, root -> tip, the strands along
        a Morton curve through their first vertices; what belongs to no strand (background segments, closed loops, their
        endpoints) follows in its present order.  new[i] = old[perm[i]].  Needs a current strands_info."""
        info = self.strands_info
        P, E = int(self.endpoint_pairs.shape[0]), int(self._endpoints.shape[0])
        offsets, rows = np.asarray(info.offsets, np.int64), np.asarray(info.rows, np.int64).reshape(-1, 2)
        S, total = len(offsets) - 1, rows.shape[0]
        pairs = self.endpoint_pairs.detach().cpu().numpy().astype(np.int64)
        if S == 0 or total == 0:
            return np.arange(P), np.arange(E)
        # strands along the curve
        first = self._endpoints.detach()[torch.as_tensor(rows[offsets[:-1], 0], device=self._endpoints.device)].cpu().numpy()
        lo, hi = first.min(axis=0), first.max(axis=0)
        q = np.clip(((first - lo) / np.maximum(hi - lo, 1e-20) * ((1 << bits) - 1)).astype(np.int64), 0, (1 << bits) - 1)
        code = np.zeros(S, np.int64)
        for b in range(bits):
            for a in range(3):
                code |= ((q[:, a] >> b) & 1) << (3 * b + a)
        order = np.argsort(code, kind="stable")
        lens = np.diff(offsets)[order]
        new_off = np.zeros(S + 1, np.int64)
        np.cumsum(lens, out=new_off[1:])
        sid = np.repeat(np.arange(S), lens)                                  # new strand number of every row
        src = np.repeat(offsets[:-1][order], lens) + (np.arange(total) - np.repeat(new_off[:-1], lens))
        rows_o = rows[src]
        # the row of endpoint_pairs behind every (cur, next) pair of a strand (chains: an unordered pair occurs once)
        key_all = np.minimum(pairs[:, 0], pairs[:, 1]) * E + np.maximum(pairs[:, 0], pairs[:, 1])
        by_key = np.argsort(key_all, kind="stable")
        key_rows = np.minimum(rows_o[:, 0], rows_o[:, 1]) * E + np.maximum(rows_o[:, 0], rows_o[:, 1])
        seg_in = by_key[np.searchsorted(key_all[by_key], key_rows)]
        rest = np.ones(P, bool)
        rest[seg_in] = False
        seg_perm = np.concatenate([seg_in, np.nonzero(rest)[0]])
        # vertices of a strand: `cur` of every row, then `next` of its last row
        ep_in = np.empty(total + S, np.int64)
        ep_in[np.arange(total) + sid] = rows_o[:, 0]
        ep_in[new_off[1:] + np.arange(S)] = rows_o[new_off[1:] - 1, 1]
        rest_e = np.ones(E, bool)
        rest_e[ep_in] = False
        ep_perm = np.concatenate([ep_in, np.nonzero(rest_e)[0]])
        self._order_aux = (order, new_off, rows_o)     # (for the strands-info remap of sort_spatially)
        return seg_perm, ep_perm

    def sort_spatially(self, bits=10):
        """Re-order the storage (parameters, Adam moments, statistics, the id tables) into storage_order().  A strand model
        starts out strand by strand, but clones, splits and merges append what they create at the END of the arrays: after
        a few hundred iterations of densification 256 consecutive Gaussians no longer share a handful of tiles, and the
        binning kernels' block-private tile tables send an atomic per (workgroup, tile) to the same few counter lines
        (measured on a model trained for 1000 iterations, 221 k segments: preprocess 31 us and scatter 70 us per pass
        against 15 / 28 us for 200 k segments in strand order).  Invisible to the reference's semantics: the order of
        segments and the numbering of endpoints carry no meaning (scene/hair_gaussian_model.py:469-622 append and compact
        without one).  Returns (segment permutation, endpoint permutation) as applied, or None if nothing moved."""
        if self.strands_info is None or self.endpoint_pairs.numel() == 0:
            return None
        if self._endpoints.is_cuda and getattr(self, "_strands_dev", None) is not None \
                and self._strands_dev[0].numel() == len(self.strands_info.offsets):
            return self._sort_spatially_device(bits)
        seg_perm, ep_perm = self.storage_order(bits)
        if np.array_equal(seg_perm, np.arange(seg_perm.size)) and np.array_equal(ep_perm, np.arange(ep_perm.size)):
            return None
        dev = self._endpoints.device
        sp, epp = torch.as_tensor(seg_perm, device=dev), torch.as_tensor(ep_perm, device=dev)
        inv_ep = torch.empty_like(epp)
        inv_ep[epp] = torch.arange(epp.numel(), device=dev)
        self._apply_storage_permutation(sp, epp, inv_ep)
        self._strands_dev = None
        # The strands themselves did not change: their bookkeeping is renumbered instead of walked again (at 4 10^5 segments a
        # walk + the root distances cost 75 ms, and this runs at every densification and merge).  In the new numbering a
        # strand's vertices are consecutive ids, root first, and the strands follow each other in storage order -- which is
        # exactly what a fresh compute_strands_info() derives (chains start at their smaller end id, strands are numbered by
        # their starts, the orientation was root -> tip already): tests/test_models_cpu.py compares the two.
        order, new_off, rows_o = self._order_aux
        info = self.strands_info
        inv_ep_h = np.empty(ep_perm.size, np.int64)
        inv_ep_h[ep_perm] = np.arange(ep_perm.size)
        rows_new = inv_ep_h[rows_o]
        S = order.size
        i2s = -np.ones(ep_perm.size, np.int32)
        sid = np.repeat(np.arange(S, dtype=np.int32), np.diff(new_off))
        i2s[rows_new[:, 0]] = sid
        i2s[rows_new[:, 1]] = sid
        comp = -np.ones(ep_perm.size, np.int32)
        starts, ends = rows_new[new_off[:-1], 0], rows_new[new_off[1:] - 1, 1]
        comp[starts], comp[ends] = ends, starts
        self.strands_info = StrandsInfo(new_off.astype(np.int64), rows_new, np.arange(rows_new.shape[0], dtype=np.int64), i2s, comp)
        self._order_aux = None
        return seg_perm, ep_perm

    def _sort_spatially_device(self, bits=10):
        """sort_spatially for a model on the GPU, without leaving it: the same permutations and the same renumbered strand
        bookkeeping as storage_order() + the host code above (tests/test_gpu_train.py compares them), from the device copies of
        the strand tables that compute_strands_info keeps.  (The numpy form took 88 ms at 3 10^5 segments -- most of what a
        densification / merge event cost; this one a few.)"""
        dev = self._endpoints.device
        offsets, rows, seg_full = self._strands_dev
        P, E = int(self.endpoint_pairs.shape[0]), int(self._endpoints.shape[0])
        S, total = int(offsets.numel()) - 1, int(rows.shape[0])
        if S == 0 or total == 0:
            return None
        i64 = dict(dtype=torch.int64, device=dev)
        ep = self._endpoints.detach()
        first = ep[rows[offsets[:-1], 0]]
        lo, hi = first.min(dim=0).values, first.max(dim=0).values
        top = (1 << bits) - 1
        q = ((first - lo) / torch.clamp_min(hi - lo, 1e-20) * top).to(torch.int64).clamp_(0, top)
        bvec = torch.arange(bits, **i64)
        # (bit b of axis a -> bit 3 b + a: one broadcast expression instead of 3 x bits x 4 launches)
        code = ((((q[:, :, None] >> bvec) & 1) << (3 * bvec + torch.arange(3, **i64)[:, None])).sum(dim=(1, 2)))
        order = torch.argsort(code, stable=True)
        lens = (offsets[1:] - offsets[:-1])[order]
        new_off = torch.zeros(S + 1, **i64)
        torch.cumsum(lens, dim=0, out=new_off[1:])
        ar_s, ar_t = torch.arange(S, **i64), torch.arange(total, **i64)
        sid = torch.repeat_interleave(ar_s, lens, output_size=total)
        src = torch.repeat_interleave(offsets[:-1][order], lens, output_size=total) + \
            (ar_t - torch.repeat_interleave(new_off[:-1], lens, output_size=total))
        rows_o = rows[src]
        seg_in = seg_full[src]
        rest = torch.ones(P, dtype=torch.bool, device=dev)
        rest[seg_in] = False
        sp = torch.cat([seg_in, torch.nonzero(rest).squeeze(1)])
        ep_in = torch.empty(total + S, **i64)
        ep_in[ar_t + sid] = rows_o[:, 0]
        ep_in[new_off[1:] + ar_s] = rows_o[new_off[1:] - 1, 1]
        rest_e = torch.ones(E, dtype=torch.bool, device=dev)
        rest_e[ep_in] = False
        epp = torch.cat([ep_in, torch.nonzero(rest_e).squeeze(1)])
        if bool(torch.equal(sp, torch.arange(P, **i64))) and bool(torch.equal(epp, torch.arange(E, **i64))):
            return None
        inv_ep = torch.empty_like(epp)
        inv_ep[epp] = torch.arange(E, **i64)
        self._apply_storage_permutation(sp, epp, inv_ep)
        # the strands' bookkeeping, renumbered (see sort_spatially)
        rows_new = inv_ep[rows_o]
        i2s = torch.full((E,), -1, dtype=torch.int32, device=dev)
        sid32 = sid.to(torch.int32)
        i2s[rows_new[:, 0]] = sid32
        i2s[rows_new[:, 1]] = sid32
        comp = torch.full((E,), -1, dtype=torch.int32, device=dev)
        starts, ends = rows_new[new_off[:-1], 0], rows_new[new_off[1:] - 1, 1]
        comp[starts], comp[ends] = ends.to(torch.int32), starts.to(torch.int32)
        self.strands_info = StrandsInfo(new_off.cpu().numpy(), rows_new.cpu().numpy(), np.arange(total, dtype=np.int64),
                                        i2s.cpu().numpy(), comp.cpu().numpy())
        self._strands_dev = (new_off, rows_new, ar_t)
        self._comp_dev = comp
        return sp.cpu().numpy(), epp.cpu().numpy()

    def _apply_storage_permutation(self, sp, epp, inv_ep):
        """Parameters, Adam moments, statistics and id tables into the order (sp: segments, epp: endpoints); device tensors."""
        dev = self._endpoints.device
        self.endpoint_pairs = inv_ep[self.endpoint_pairs[sp]]
        if torch.is_tensor(self.strand_root_endpoint_idx) and self.strand_root_endpoint_idx.numel():
            self.strand_root_endpoint_idx = torch.sort(inv_ep[self.strand_root_endpoint_idx.to(dev)]).values
        if self.optimizer is not None:
            out = {}
            for g in self.optimizer.param_groups:
                perm = epp if g["name"] == "endpoints" else sp
                out[g["name"]] = self._swap_param(g, g["params"][0].detach()[perm], lambda m, k=perm: m[k])
            self._rebind(out)
        else:
            for name, attr in self._PARAM_ATTRS:
                perm = epp if name == "endpoints" else sp
                setattr(self, attr, nn.Parameter(getattr(self, attr).detach()[perm].requires_grad_(True)))
        for attr in ("xyz_gradient_accum", "denom", "max_radii2D"):
            t = getattr(self, attr, None)
            if torch.is_tensor(t) and t.shape[0] == sp.shape[0]:
                setattr(self, attr, t[sp])
        self._smooth_pairs = None

    _storage_dirty = False
    _strands_dev = None
    _comp_dev = None        # strand end -> the other end of its strand, on the device (int32 [E], -1 elsewhere): with _strands_dev
    _strands_info = None

    @property
    def strands_info(self):
        return self._strands_info

    @strands_info.setter
    def strands_info(self, value):
        # the device copies of the strand tables (`_strands_dev`) and the smoothness pair table built from them describe the
        # strands_info they were made WITH: whoever assigns a new one starts without them (compute_strands_info and
        # sort_spatially put their device tables back right after the assignment) -- a table of equal size from an older
        # topology can then never be taken for the current one
        self._strands_info = value
        self._strands_dev = None
        self._comp_dev = None
        self._smooth_pairs = None

    def _maybe_sort_spatially(self):
        """training_args.spatial_sort (default on) for a model on the GPU, once the operators of an iteration have changed
        the topology (they set _storage_dirty; the training step calls this behind the last of them: one sort per event)."""
        dirty, self._storage_dirty = self._storage_dirty, False
        if dirty and getattr(getattr(self, "training_args", None), "spatial_sort", True) and self._endpoints.is_cuda:
            self.sort_spatially()

    def __init__(self, sh_degree: int = 3, spatial_lr_scale: float = 1.0, device: str = "cuda"):
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        self.ref_strand_root = np.empty(0)
        self.strand_root_endpoint_idx = torch.empty(0)
        self.endpoint_pairs = torch.empty(0)
        for _, attr in self._PARAM_ATTRS:
            setattr(self, attr, torch.empty(0))
        self.max_radii2D = torch.empty(0)
        self.xyz_gradient_accum = torch.empty(0)
        self.denom = torch.empty(0)
        self.optimizer = None
        self.spatial_lr_scale = spatial_lr_scale
        self.device = device
        self.setup_functions()
        self.strands_info = None  # must be refreshed before topology operators run
        self._smooth_pairs = None  # cached index tensor for the smoothness loss (depends on strands_info only)

    def _capture_attrs(self):
        return [(None, a) for a in ("ref_strand_root", "strand_root_endpoint_idx", "endpoint_pairs", "_endpoints",
                                    "_features_dc", "_features_rest", "_opacity", "_mask", "_width")]

    # ---- derived Gaussian parameters -------------------------------------------------------
    # On the GPU the four derived tensors come from ONE fused HIP kernel (hgs_strand_geometry_*).  Like the reference's
    # getters (scene/hair_gaussian_model.py:134-201) they are recomputed from the parameters on EVERY call: nothing is
    # cached on the model, so no write to the parameters -- optimizer step, `.data` mutation, raw-pointer kernel -- can
    # ever be served stale geometry.  A caller that needs several of them (render()) asks `derived_gaussians()` once.
    # `fused_geometry = False` (or CPU tensors) selects the op-by-op PyTorch formulas below, which restate the
    # reference getters and are what the fused kernel is tested against.
    fused_geometry = True

    def _segment_delta(self):
        pairs = self._endpoints[self.endpoint_pairs]
        return pairs, pairs[:, 1] - pairs[:, 0]

    def derived_gaussians(self):
        """(xyz [P,3], scaling [P,3], rotation [P,4], orientation [P,3]) of the current parameters from one launch of the
        fused geometry kernel (differentiable w.r.t. endpoints / width), or None where the op-by-op formulas apply."""
        if not (self.fused_geometry and self._endpoints.is_cuda):
            return None
        from hgs_runtime.fused import strand_geometry
        return strand_geometry(self._endpoints, self._width, self.endpoint_pairs, float(self.dist_to_scale_factor))

    _fused = derived_gaussians

    @property
    def get_scaling(self):
        f = self._fused()
        if f is not None:
            return f[1]
        _, diff = self._segment_delta()
        half_len = torch.norm(diff, p=2, dim=1, keepdim=True) / 2
        scale_x = torch.clamp(half_len * self.dist_to_scale_factor, min=self.min_val)
        scale_yz = self.scaling_activation(self._width.repeat(1, 2))
        return torch.cat((scale_x, scale_yz), dim=1)

    @property
    def get_rotation(self):
        f = self._fused()
        if f is not None:
            return f[2]
        _, v2 = self._segment_delta()
        rotation = torch.zeros((v2.shape[0], 4), dtype=torch.float, device=v2.device)
        rotation[:, 0] = 1.0
        valid = torch.norm(v2, p=2, dim=1) > self.min_val  # collapsed segments keep the identity
        x_hat = torch.zeros_like(v2[valid])
        x_hat[:, 0] = 1.0
        rotation[valid] = calculate_rotation_from_vectors(x_hat, v2[valid], representation="quat")
        return rotation

    @property
    def get_xyz(self):
        f = self._fused()
        if f is not None:
            return f[0]
        return torch.mean(self._endpoints[self.endpoint_pairs], dim=1)

    @property
    def get_orientation(self):
        """World-space unit direction of every segment; x_hat for collapsed ones (reference :188-201)."""
        f = self._fused()
        if f is not None:
            return f[3]
        _, d = self._segment_delta()
        norm = torch.norm(d, p=2, dim=1, keepdim=True)
        ok = (norm >= self.min_val).squeeze(1)
        out = torch.zeros_like(d)
        out[:, 0] = 1.0
        out[ok] = d[ok] / norm[ok]
        return out

    def get_covariance(self, scaling_modifier=0.5):
        return self.covariance_activation(self.get_scaling, scaling_modifier, self.get_rotation)

    def _num_primitives(self):
        return self.endpoint_pairs.shape[0]

    # ---- on-disk format (reference :292-466), scene/ply_io.py ----
    def construct_list_of_attributes(self):
        from scene.ply_io import hair_attributes
        return hair_attributes(self)

    def save_ply(self, path):
        from scene.ply_io import save_hair_ply
        save_hair_ply(self, path)

    def load_ply(self, path):
        from scene.ply_io import load_hair_ply
        load_hair_ply(self, path)

    def create_from_pcd(self, pcd):
        raise NotImplementedError("This method is only intended for Gaussian Model")

    # ---- optimizer ---------------------------------------------------------------------------
    def _group_lrs(self, ta):
        return {"endpoints": ta.position_lr_init * self.spatial_lr_scale, "f_dc": ta.feature_lr,
                "f_rest": ta.feature_lr / 20.0, "opacity": ta.opacity_lr, "mask": ta.mask_lr, "width": ta.scaling_lr}

    def training_setup(self, training_args, sort=True):
        """6 Adam groups + endpoint lr schedule + merge distance/angle schedules + max segment length
        (reference :212-283)."""
        n = self._num_primitives()
        self.max_radii2D = torch.zeros((n,), device=self.device)
        self.xyz_gradient_accum = torch.zeros((n, 1), device=self.device)
        self.denom = torch.zeros((n, 1), device=self.device)
        lrs = self._group_lrs(training_args)
        groups = [{"params": [getattr(self, attr)], "lr": lrs[name], "name": name} for name, attr in self._PARAM_ATTRS]
        self.optimizer = self._make_optimizer(groups)
        ta = training_args

        def sched(a, b):
            return get_expon_lr_func(lr_init=a, lr_final=b, lr_delay_mult=ta.position_lr_delay_mult,
                                     max_steps=ta.position_lr_max_steps)
        self.endpoints_scheduler = sched(ta.position_lr_init * self.spatial_lr_scale,
                                         ta.position_lr_final * self.spatial_lr_scale)
        self.xyz_scheduler_args = self.endpoints_scheduler
        self.merge_dist_th, self.merge_dist_th_scheduler = ta.merge_dist_th_init, sched(ta.merge_dist_th_init,
                                                                                         ta.merge_dist_th_final)
        self.merge_angle_th, self.merge_angle_th_scheduler = ta.merge_angle_th_init, sched(ta.merge_angle_th_init,
                                                                                            ta.merge_angle_th_final)
        self.set_pval(ta.pval)
        self.training_args = ta
        # longest allowed segment = foreground bounding-box diagonal / num_points_strand.  Evaluated without autograd:
        # the reference leaves a graph hanging off this tensor (it indexes the parameter with grad enabled), which
        # keeps the parameter's AccumulateGrad node alive on the creating stream and breaks later graph capture.
        with torch.no_grad():
            fg = (self.get_mask >= self.foreground_binarization_th).squeeze(1)
            used = torch.zeros(self._endpoints.shape[0], dtype=torch.bool, device=self.device)
            used[self.endpoint_pairs[fg].flatten()] = True
            pts = self._endpoints[used]
            self.max_segment_length = torch.norm(pts.max(dim=0).values - pts.min(dim=0).values) / ta.num_points_strand

    def update_learning_rate(self, iteration):
        for group in self.optimizer.param_groups:
            if group["name"] == "endpoints":
                group["lr"] = self.endpoints_scheduler(iteration)
        self.merge_dist_th = self.merge_dist_th_scheduler(iteration)
        self.merge_angle_th = self.merge_angle_th_scheduler(iteration)

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

    def reset_opacity(self):
        new = inverse_sigmoid(torch.min(self.get_opacity, torch.ones_like(self.get_opacity) * 0.01))
        self._rebind(self.replace_tensor_to_optimizer(new, "opacity"))

    # ---- segment-level surgery (reference :469-622) ------------------------------------------
    def _replace_tensor_in_optimizer(self, tensor, name):
        return self.replace_tensor_to_optimizer(tensor, name)

    def cat_segments(self, new_endpoint_pairs, new_endpoints, new_features_dc, new_features_rest, new_opacities,
                     new_masks, new_widths):
        self.endpoint_pairs = torch.cat([self.endpoint_pairs, new_endpoint_pairs], dim=0)
        self._rebind(self.cat_tensors_to_optimizer({
            "endpoints": new_endpoints, "f_dc": new_features_dc, "f_rest": new_features_rest, "opacity": new_opacities,
            "mask": new_masks, "width": new_widths}))
        self._reset_stats()
        self._smooth_pairs = None

    def prune_segments(self, segments_prune_mask):
        """Drop segments; endpoints no segment references any more are dropped too and ids are compacted."""
        # Round 5: every row selection below goes through ONE index list per mask (nonzero: the call's two host
        # synchronisations) and index_select -- the same rows in the same order as boolean-mask indexing, which costs a
        # nonzero kernel and a synchronisation per tensor: 18 of them per call (parameter, both Adam moments, six groups),
        # 0.7 ms of a call that the operators make ~8 times per densification event (tools/dev/soak_profile.py).
        pairs, roots, seg_idx, ep_idx = self._prune_plan(self.endpoint_pairs, self.strand_root_endpoint_idx,
                                                         self._endpoints.shape[0], segments_prune_mask)
        self.endpoint_pairs, self.strand_root_endpoint_idx = pairs, roots
        self._apply_row_selection(seg_idx, ep_idx)

    def _prune_plan(self, pairs, roots, n_endpoints, segments_prune_mask, keeps_every_segment=False):
        """The index side of prune_segments, on values instead of the model: (pairs with the pruned rows gone and the ids
        compacted, root ids renumbered, index list of the kept segment rows or None if all are kept, index list of the kept
        endpoints or None).  No parameter is touched: a caller that prunes several times in a row (merge_collapsed_segments: two
        calls per round, five rounds on a trained model) composes the index lists and re-creates the tensors ONCE."""
        seg_idx = None
        if not keeps_every_segment:
            seg_keep = ~segments_prune_mask
            seg_idx = torch.nonzero(seg_keep).squeeze(1)
            if seg_idx.shape[0] == seg_keep.shape[0]:
                seg_idx = None
            else:
                pairs = pairs.index_select(0, seg_idx)
        ep_keep = torch.zeros(n_endpoints, dtype=torch.bool, device=self.device)
        ep_keep[pairs.flatten()] = True
        ep_idx = torch.nonzero(ep_keep).squeeze(1)
        if ep_idx.shape[0] == ep_keep.shape[0]:
            ep_idx = None
        else:
            remap = torch.cumsum(ep_keep.to(torch.long), dim=0) - 1  # old id -> new id for kept endpoints
            pairs = remap[pairs]
            if torch.is_tensor(roots) and roots.numel():
                roots = remap[roots]
        return pairs, roots, seg_idx, ep_idx

    def _apply_row_selection(self, seg_idx, ep_idx):
        """Re-create the per-segment tensors (parameters, Adam moments, statistics) as their rows seg_idx and the endpoints as
        their rows ep_idx (None: all rows, the tensor stays as it is)."""
        # (a call that keeps every segment -- the reference's id compaction, twice per round of merge_collapsed_segments -- or
        # every endpoint leaves those tensors as they are: a masked copy of all rows is the same rows)
        out = {}
        for g in self.optimizer.param_groups:
            idx = ep_idx if g["name"] == "endpoints" else seg_idx
            if idx is None:
                g["params"][0].grad = None      # (a re-created parameter has no gradient: it skips this iteration's Adam step)
                continue
            out[g["name"]] = self._swap_param(g, g["params"][0].detach().index_select(0, idx), lambda m, k=idx: m.index_select(0, k))
        if out:
            self._rebind(out)
        if seg_idx is not None:
            self.xyz_gradient_accum = self.xyz_gradient_accum.index_select(0, seg_idx)
            self.denom = self.denom.index_select(0, seg_idx)
            self.max_radii2D = self.max_radii2D.index_select(0, seg_idx)
        self._smooth_pairs = None

    # ---- strand bookkeeping ----------------------------------------------------------------------
    def update_strand_root(self, dist_th: float = 1e-2):
        """Endpoints that are the nearest neighbour of a reference root within sqrt(dist_th) become strand roots
        (reference :1373-1399 uses pytorch3d.knn_points whose `dist` is the SQUARED distance)."""
        if self.ref_strand_root is None or self.ref_strand_root.shape[0] == 0:
            return
        roots = torch.from_numpy(np.asarray(self.ref_strand_root)).to(self.device).to(self._endpoints.dtype)
        ep = self._endpoints.detach()
        sel = torch.zeros(ep.shape[0], dtype=torch.bool, device=self.device)
        for s in range(0, roots.shape[0], 4096):
            d2 = torch.cdist(roots[s:s + 4096], ep).pow(2)
            best, arg = d2.min(dim=1)
            sel[arg[best <= dist_th]] = True
        self.strand_root_endpoint_idx = torch.nonzero(sel).squeeze(1)
        print(f"Identified {int(sel.sum())} endpoints as strand roots")

    def compute_strands_info(self, only_foreground: bool = True):
        """Walk every open polyline of `endpoint_pairs` from one end to the other and orient it root -> tip by the
        distance of its two ends to the nearest reference root (reference :1410-1498).  Assumes well-formed
        chains (every endpoint id appears once or twice, no cycles)."""
        if self.ref_strand_root is None or np.asarray(self.ref_strand_root).shape[0] == 0:
            raise ValueError("ref_strand_root is not set")
        if self._endpoints.is_cuda:
            # on the device; only the results come to the host
            pairs_t = self.endpoint_pairs
            ep = self._endpoints.detach()
            roots = torch.as_tensor(np.asarray(self.ref_strand_root), device=ep.device)
            chunk = max(1024, (1 << 27) // max(1, roots.shape[0]))       # <= 1 GiB of float64 distances at a time
            fg_rows = None
            if only_foreground:
                fg_rows = torch.nonzero(self.compute_foreground_mask()).squeeze(1)
                pairs_t = self.endpoint_pairs[fg_rows]
            # (longest strand so far, with room for what one operator can do to it -- a merge joins at most three strands end to
            # end, a split doubles a segment: walk_chains_torch checks the guess and falls back)
            prev = getattr(self, "_longest_strand", None)
            end_distance = lambda ids: nearest_distance(ep[ids], roots, chunk)
            walked = walk_chains_device(pairs_t, ep.shape[0], end_distance) if os.environ.get("HGS_STRAND_WALK", "device") == "device" else None
            if walked is None:      # (a table that is not a set of chains, or the torch form asked for: HGS_STRAND_WALK=torch)
                walked = walk_chains_torch(pairs_t, ep.shape[0], end_distance, max_len_hint=None if prev is None else max(64, 4 * prev))
            offsets, rows, seg_rows, i2s, comp = walked
            self.strands_info = StrandsInfo(offsets.cpu().numpy(), rows.cpu().numpy(), seg_rows.cpu().numpy(),
                                            i2s.cpu().numpy(), comp.cpu().numpy())
            off_h = self.strands_info.offsets
            self._longest_strand = int((off_h[1:] - off_h[:-1]).max()) if len(off_h) > 1 else None
            # device copies for sort_spatially (offsets, rows, the row of endpoint_pairs behind every strand row)
            self._strands_dev = (offsets, rows, seg_rows if fg_rows is None else fg_rows[seg_rows])
            self._comp_dev = comp
            self._smooth_pairs = None
            return
        self._strands_dev = None
        from scipy.spatial import cKDTree
        tree = cKDTree(np.asarray(self.ref_strand_root))
        endpoints = self._endpoints.detach().cpu().numpy()
        pairs = self.endpoint_pairs.cpu().numpy()
        if only_foreground:
            m = self.compute_foreground_mask().cpu().numpy()
            pairs = pairs[m]
        n_ep = endpoints.shape[0]
        id_to_strand = -np.ones(n_ep, np.int32)
        complementary = -np.ones(n_ep, np.int32)
        if pairs.shape[0] == 0:
            self.strands_info = StrandsInfo(np.zeros(1, np.int64), np.zeros((0, 2), np.int64), np.zeros(0, np.int64),
                                            id_to_strand, complementary)
            self._smooth_pairs = None
            return
        offsets, rows, seg_rows = walk_chains(pairs, n_ep, id_to_strand, complementary,
                                              lambda ends: tree.query(endpoints[ends], k=1, workers=-1)[0])
        self.strands_info = StrandsInfo(offsets, rows, seg_rows, id_to_strand, complementary)
        self._smooth_pairs = None

    def smoothness_index_pairs(self):
        """[pairs, 2, 2] endpoint-id tensor of consecutive segments of every strand, cached on the device until the
        topology changes.  (The reference rebuilds it on the CPU through Cython every iteration, losses.py:193-199.)"""
        if self._smooth_pairs is None:
            dev_tables = getattr(self, "_strands_dev", None)
            if self._endpoints.is_cuda and dev_tables is not None and dev_tables[0].numel() == len(self.strands_info.offsets) \
                    and dev_tables[1].shape[0] == len(self.strands_info.rows):
                # on the device, from the strand tables compute_strands_info / sort_spatially left there: every row of a strand
                # but its last one, with its successor -- c_utils.filter_strand_segments_flat's pairs in its order
                # (tests/test_gpu_train.py::test_device_smoothness_pairs_equal_the_native_filter) without the 1.1 ms upload of
                # a 10 MB host table per topology event (round 5: tools/dev/soak_profile.py)
                off, rows = dev_tables[0], dev_tables[1].reshape(-1, 2)
                total = rows.shape[0]
                if total == 0:
                    self._smooth_pairs = torch.empty((0, 2, 2), dtype=torch.long, device=self.device)
                else:
                    keep = torch.ones(total, dtype=torch.bool, device=self.device)
                    last = off[1:][off[1:] > off[:-1]] - 1
                    keep[last] = False
                    first = torch.nonzero(keep).squeeze(1)
                    self._smooth_pairs = torch.stack((rows.index_select(0, first), rows.index_select(0, first + 1)), dim=1).to(torch.long).contiguous()
            else:
                from c_utils import filter_strand_segments_flat
                idx = filter_strand_segments_flat(*self.strands_info.flat)
                self._smooth_pairs = torch.as_tensor(np.asarray(idx), device=self.device, dtype=torch.long)
        return self._smooth_pairs

    # ---- construction from explicit polylines (synthetic scenes; the reference builds strands through merge.py) --
    @classmethod
    def from_strands(cls, strand_points, width=1e-4, opacity=None, mask_prob=0.9, colors=None, sh_degree=0,
                     spatial_lr_scale=1.0, device="cuda", ref_strand_root=None):
        """strand_points: [S, V, 3] polylines; consecutive vertices share an endpoint (E = S*V, P = S*(V-1))."""
        sp = torch.as_tensor(strand_points, dtype=torch.float32)
        S, V, _ = sp.shape
        m = cls(sh_degree=sh_degree, spatial_lr_scale=spatial_lr_scale, device=device)
        ids = torch.arange(S * V).reshape(S, V)
        pairs = torch.stack((ids[:, :-1].reshape(-1), ids[:, 1:].reshape(-1)), dim=1)
        P = pairs.shape[0]
        n_coef = (sh_degree + 1) ** 2
        if colors is None:
            colors = torch.full((P, 3), 0.5)
        from utils.sh import RGB2SH
        f_dc = RGB2SH(torch.as_tensor(colors, dtype=torch.float32)).reshape(P, 1, 3)
        if opacity is None:
            opacity = torch.full((P, 1), 0.8)
        opacity = torch.as_tensor(opacity, dtype=torch.float32).reshape(P, 1)
        m._endpoints = nn.Parameter(sp.reshape(-1, 3).to(device).contiguous().requires_grad_(True))
        m.endpoint_pairs = pairs.to(device)
        m._features_dc = nn.Parameter(f_dc.to(device).contiguous().requires_grad_(True))
        m._features_rest = nn.Parameter(torch.zeros((P, n_coef - 1, 3), device=device).requires_grad_(True))
        m._opacity = nn.Parameter(inverse_sigmoid(opacity).to(device).requires_grad_(True))
        m._mask = nn.Parameter(inverse_sigmoid(torch.full((P, 1), float(mask_prob))).to(device).requires_grad_(True))
        m._width = nn.Parameter(torch.full((P, 1), float(np.log(width)), device=device).requires_grad_(True))
        m.ref_strand_root = (sp[:, 0].numpy() if ref_strand_root is None else np.asarray(ref_strand_root))
        m.strand_root_endpoint_idx = ids[:, 0].to(device)
        return m
