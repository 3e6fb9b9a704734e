"""Strand topology operators of the Stage-III model (counterparts of the reference's
scene/hair_gaussian_model.py:619-706 merge_endpoint_pairs, :712-784 index helpers, :788-1077 densification =
clone / split / merge-collapsed / prune, :1079-1096 merging, :1205-1362 compute_endpoint_pair_to_merge, :1500-1515
clean_gaussians).  Mixed into HairGaussianModel.  They run every `densification_interval` / `merge_interval` = 100
iterations on replicated state.  The reference loops over CUDA tensors element by element (:1246-1253); here the candidate
search is one query (kd-tree on the CPU, hgs_radius_pairs on the GPU, where the candidates also stay on the device up to the
first-occurrence rule) and the greedy selections, sequential in the reference, are resolved in vectorised rounds that keep its
order (compute_endpoint_pair_to_merge)."""
import os

import numpy as np
import torch


def _info_dict(info):
    """The strategies take the reference's `training_info` (an object with a `densification_info` dict, utils/logging.py) or
    the dict itself."""
    return getattr(info, "densification_info", info)


class HairTopologyMixin:
    # ---- index helpers -------------------------------------------------------------------------------------------
    def _endpoint_degree_table(self):
        """deg[id] = number of segments that reference endpoint id, for every id below max(id) + 1 (0: not referenced) -- what
        the reference reads off torch.unique(endpoint_pairs, return_counts=True) + torch.isin (`u[c == 1]` are the ids of degree 1,
        `isin(x, u[c != 1])` is deg[x] != 1 for ids that occur).  One bincount, and a gather per question instead of a sort-based
        membership test over 6 x 10^5 ids (torch.isin: 0.4 ms a call, four to six calls per topology event); remembered for as
        long as `endpoint_pairs` is the same tensor object (every topology change assigns a new one)."""
        cached = getattr(self, "_degree_cache", None)
        if cached is None or cached[0] is not self.endpoint_pairs:
            flat = self.endpoint_pairs.reshape(-1)
            deg = torch.bincount(flat) if flat.numel() else torch.zeros(0, dtype=torch.long, device=self.device)
            cached = self._degree_cache = (self.endpoint_pairs, deg)
        return cached[1]

    def get_first_occurence_index(self, tensor):
        """Index of the first occurrence of every unique value (reference :772-784)."""
        uniq, inv = torch.unique(tensor, return_inverse=True, sorted=False, dim=0)
        perm = torch.arange(inv.shape[0], dtype=inv.dtype, device=tensor.device)
        inv, perm = inv.flip([0]), perm.flip([0])
        return inv.new_empty(uniq.shape[0]).scatter_(0, inv, perm)

    def remove_duplicate_endpoint_rows(self, index_pairs, return_mask=False):
        """Keep the rows whose two ids both occur there for the first time in row-major order (reference :712-728)."""
        flat = index_pairs.flatten()
        if flat.numel():
            # position of every id's first occurrence by a scatter-min (order-independent: deterministic), instead of
            # get_first_occurence_index's unique(dim=0) + flips: mask[i] <=> flat[i] occurs at i for the first time
            pos = torch.arange(flat.shape[0], dtype=torch.long, device=flat.device)
            first = torch.full((int(flat.max()) + 1,), flat.shape[0], dtype=torch.long, device=flat.device)
            first.scatter_reduce_(0, flat, pos, reduce="amin", include_self=True)
            mask = first[flat] == pos
        else:
            mask = torch.zeros(0, dtype=torch.bool, device=self.device)
        mask = mask.reshape(-1, 2)
        mask = mask[:, 0] & mask[:, 1]
        return (index_pairs[mask], mask) if return_mask else index_pairs[mask]

    def get_endpoint_pairs_row_indices(self, endpoint_id, exclude_segments=None):
        """Row of `endpoint_pairs` holding each id (the last one if it occurs twice; -1 if none) (reference :730-752)."""
        mapping = -torch.ones(int(self.endpoint_pairs.max()) + 1, dtype=torch.long, device=self.device)
        rows = torch.arange(self.endpoint_pairs.shape[0], device=self.device)
        pairs = self.endpoint_pairs
        if exclude_segments is not None:
            pairs, rows = pairs[~exclude_segments], rows[~exclude_segments]
        mapping[pairs[:, 0]] = rows
        mapping[pairs[:, 1]] = rows
        return mapping[endpoint_id]

    def get_complementary_endpoint_idx(self, endpoint_id, exclude_segments=None):
        rows = self.get_endpoint_pairs_row_indices(endpoint_id, exclude_segments)
        sel = self.endpoint_pairs[rows]
        return torch.where(sel[:, 1] == endpoint_id, sel[:, 0], sel[:, 1]), rows

    # ---- densification (reference :788-1077) -------------------------------------------------------------------------
    def densification(self, extent, max_screen_size, training_info=None):
        grads = self.xyz_gradient_accum / self.denom
        grads = torch.where(grads.isnan(), torch.zeros_like(grads), grads)      # grads[grads.isnan()] = 0.0 without its host sync
        # (without a training_info nobody reads the operators' counters: None spares them a synchronisation each)
        info = training_info.densification_info if training_info is not None else None
        self.clone_strategy(grads, extent, info)
        self.split_strategy(grads, extent, info)
        self.merge_collapsed_segments(info)
        self.prune_strategy(extent, max_screen_size, info, avoid_connected=True)
        self.compute_strands_info()
        self._storage_dirty = True        # (what the operators created sits at the end of the arrays: sort_spatially, once the
                                          #  iteration's operators are through -- train._training_step)

    def _segment_lengths(self):
        seg = self._endpoints[self.endpoint_pairs]
        return torch.norm(seg[:, 1] - seg[:, 0], p=2, dim=1)

    def clone_strategy(self, grads, scene_extent, info=None):
        """High view-space gradient + small extent -> duplicate the segment as a new, disconnected one (:915-967)."""
        info = _info_dict(info)
        ta = self.training_args
        sel = (torch.norm(grads, dim=-1) >= ta.densify_grad_threshold) & (
            torch.max(self.get_scaling, dim=1).values <= ta.percent_dense * scene_extent)
        si = torch.nonzero(sel).squeeze(1)      # (one index list for the seven row selections: same rows as [sel], one sync)
        take = lambda t: t.detach().index_select(0, si)
        new_ep = self._endpoints.detach()[take(self.endpoint_pairs)].flatten(0, 1)
        ids = torch.arange(new_ep.shape[0], device=self.device) + self.endpoint_pairs.max() + 1
        if info is not None:
            info["clone"] = int(si.shape[0])
        self.cat_segments(ids.reshape(-1, 2), new_ep, take(self._features_dc), take(self._features_rest),
                          take(self._opacity), take(self._mask), take(self._width))

    def split_strategy(self, grads, scene_extent, info=None):
        """High gradient + large extent, or longer than max_segment_length (foreground only) -> cut at the midpoint
        into two connected segments sharing a new endpoint (:828-913)."""
        info = _info_dict(info)
        ta = self.training_args
        n0 = self.endpoint_pairs.shape[0]
        padded = torch.zeros((n0,), device=self.device)
        padded[: grads.shape[0]] = grads.squeeze()
        sel = (padded >= ta.densify_grad_threshold) & (
            torch.max(self.get_scaling, dim=1).values > ta.percent_dense * scene_extent)
        sel = sel | (self._segment_lengths() >= self.max_segment_length)
        sel = sel & (self.get_mask > self.foreground_binarization_th).squeeze(1)
        si = torch.nonzero(sel).squeeze(1)      # (one index list for the seven row selections: same rows as [sel], one sync)
        k = int(si.shape[0])
        take = lambda t: t.detach().index_select(0, si)
        mid = take(self.get_xyz)
        ids = torch.arange(k, device=self.device) + 1 + torch.max(self.endpoint_pairs)
        orig = take(self.endpoint_pairs)
        new_pairs = torch.cat([torch.stack([orig[:, 0], ids], 1), torch.stack([ids, orig[:, 1]], 1)], dim=0)
        self.cat_segments(new_pairs, mid, take(self._features_dc).repeat(2, 1, 1),
                          take(self._features_rest).repeat(2, 1, 1), take(self._opacity).repeat(2, 1),
                          take(self._mask).repeat(2, 1), take(self._width).repeat(2, 1))
        if info is not None:
            info["split"] = k
        self.prune_segments(torch.cat((sel, torch.zeros(2 * k, device=self.device, dtype=torch.bool))))

    def merge_collapsed_segments(self, info=None):
        """Segments that collapsed to a point or left the foreground, and whose both ends are interior joints, are
        removed by fusing their two endpoints; repeated until nothing merges (:969-1018)."""
        # The reference prunes twice per round (the merged segments; then, with an all-False mask, the endpoint ids that lost their
        # last reference), i.e. re-creates every parameter, both Adam moments and the statistics -- ~20 tensors -- ten times on a
        # trained model (five rounds).  What a round DECIDES depends on the index structures only -- the pairs, which segment rows and
        # endpoint ids are left, the (unchanged) positions, opacities and masks behind them -- so the rounds run on those
        # (_prune_plan: the same statements on values), the row selections compose, and the tensors are re-created ONCE at the end:
        # the same rows in the same order, 3 ms less per densification event (tools/dev/operator_syncs.py).
        info = _info_dict(info)
        total = 0
        pairs, roots = self.endpoint_pairs, self.strand_root_endpoint_idx
        ep_all, fg_all = self._endpoints.detach(), self.compute_foreground_mask()
        seg_sel, ep_sel = None, None          # current segment row / endpoint id -> the model's (None: identity)
        compose = lambda sel, idx: sel if idx is None else (idx if sel is None else sel.index_select(0, idx))
        while True:
            ep = ep_all if ep_sel is None else ep_all.index_select(0, ep_sel)
            seg = ep[pairs]
            collapsed = torch.norm(seg[:, 1] - seg[:, 0], p=2, dim=1) < self.min_val          # (_segment_lengths)
            mask = collapsed | ~(fg_all if seg_sel is None else fg_all.index_select(0, seg_sel))
            cand = pairs[mask]
            if cand.shape[0] == 0:        # nothing collapsed, nothing in the background: the round that finds nothing (below)
                break
            flat = pairs.reshape(-1)
            deg = torch.bincount(flat) if flat.numel() else torch.zeros(0, dtype=torch.long, device=self.device)   # (_endpoint_degree_table)
            both_interior = torch.all(deg[cand] != 1, dim=1)     # (isin(cand, u[c != 1]))
            mask[mask.clone()] = both_interior
            to_merge = cand[both_interior]
            to_merge, keep = self.remove_duplicate_endpoint_rows(to_merge, return_mask=True)
            mask[mask.clone()] = keep
            if to_merge.shape[0] == 0:
                # the round that finds nothing (every call ends with one): `mask` is all False by now, so the reference's two
                # prune_segments calls keep every segment and every endpoint
                break
            n_ep = ep.shape[0]
            pairs, roots, seg_idx, ep_idx = self._prune_plan(pairs, roots, n_ep, mask)
            seg_sel, ep_sel = compose(seg_sel, seg_idx), compose(ep_sel, ep_idx)
            n_ep = n_ep if ep_idx is None else ep_idx.shape[0]
            mapping = torch.arange(n_ep, device=self.device)     # (ids are compact after a prune: max + 1 = their number)
            mapping[to_merge[:, 1]] = to_merge[:, 0]
            pairs = mapping[pairs]
            # compacts the endpoint table (drops the ids that just lost their last reference)
            pairs, roots, _, ep_idx = self._prune_plan(pairs, roots, n_ep, None, keeps_every_segment=True)
            ep_sel = compose(ep_sel, ep_idx)
            total += int(to_merge.shape[0])
        self.endpoint_pairs, self.strand_root_endpoint_idx = pairs, roots
        # all the reference's calls leave behind when they re-create nothing is what _apply_row_selection(None, None) leaves:
        # parameters without a gradient for this iteration's Adam step
        self._apply_row_selection(seg_sel, ep_sel)
        if info is not None:
            info["merge_collapsed"] = total

    def prune_strategy(self, extent, max_screen_size, info=None, avoid_connected=False):
        """Drop collapsed / transparent / oversized segments; with avoid_connected only strand-end or background
        segments may go, so strands are never cut in the middle (:1020-1077)."""
        info = _info_dict(info)
        prune = self._segment_lengths() < self.min_val
        if info is not None:
            info["prune_collapsed"] = int(prune.sum())
        low = (self.get_opacity < self.opacity_th).squeeze(1)
        if info is not None:
            info["prune_low_opacity"] = int(low.sum())
        prune = prune | low
        if max_screen_size and extent != 0.0:
            big = self.get_scaling.max(dim=1).values > 0.1 * extent
            if info is not None:
                info["prune_big_ws"] = int(big.sum())
            prune = prune | big
        if avoid_connected and prune.sum() != 0:
            is_end = torch.any(self._endpoint_degree_table()[self.endpoint_pairs] == 1, dim=1)   # (isin(pairs, u[c == 1]))
            allowed = is_end | (self.get_mask < self.foreground_binarization_th).squeeze(1)
            if info is not None:
                info["prune_avoided"] = int(prune.sum() - (prune & allowed).sum())
            prune = prune & allowed
        n = int(prune.sum())
        if info is not None:
            info["prune_total"] = n
        if 0 < n < self._opacity.shape[0]:
            self.prune_segments(prune)

    def clean_gaussians(self, avoid_connected=True):
        """Remove background / transparent segments (only strand-end ones when avoid_connected) (:1500-1515)."""
        prune = ~self.compute_foreground_mask()
        if avoid_connected:
            is_end = torch.any(self._endpoint_degree_table()[self.endpoint_pairs[prune]] == 1, dim=1)
            prune[prune.clone()] = is_end
        self.prune_segments(prune)

    # ---- merging (reference :1079-1096, :1205-1362, :619-706) ----------------------------------------------------------
    def merging(self, training_info=None, strands_info_is_current=False):
        """strands_info_is_current: the caller ran densification() (which ends with compute_strands_info) right before,
        with no optimizer step or topology change since -- the reference recomputes regardless (:1079-1096)."""
        if not strands_info_is_current or self.strands_info is None:
            self.compute_strands_info()
        pairs = self.compute_endpoint_pair_to_merge()
        if training_info is not None:
            training_info.densification_info["merge"] = int(pairs.shape[0])
        if pairs.shape[0] == 0:
            return                      # nothing merged: the strands are what they were
        self.merge_endpoint_pairs(pairs)
        self.compute_strands_info()
        self._storage_dirty = True

    def growing(self, training_info=None, **_):
        raise NotImplementedError("the reference's growing() cannot run either (cat_segments is called without "
                                  "`new_masks`, hair_gaussian_model.py:1187-1194) and its interval is 100000 iterations")

    def compute_endpoint_pair_to_merge(self, chunk_size=-1, max_num_nn=-1):
        """Greedy one-to-one matching of nearby strand ends (root/tip) that face each other: candidates = foreground
        strand ends within merge_dist_th of each other, not the two ends of one strand, whose outgoing directions are
        opposite within merge_angle_th; sorted by distance; a pair is kept if neither id was used before and neither
        sits on a strand whose OTHER end was already merged in this round.  Returns an [N,2] id tensor."""
        from scipy.spatial import cKDTree
        dir_th = np.cos(np.deg2rad(self.merge_angle_th))
        deg = self._endpoint_degree_table()
        ends = torch.nonzero(deg == 1).squeeze(1)            # ascending ids, like unique()'s (ids[counts == 1])
        is_fg = torch.zeros(deg.shape[0], dtype=torch.bool, device=self.device)
        is_fg[self.endpoint_pairs[self.compute_foreground_mask()].flatten()] = True
        ends = ends[is_fg[ends]]                             # (ends[isin(ends, ids of the foreground segments)])
        empty = torch.zeros((0, 2), dtype=torch.long, device=self.device)
        if ends.numel() < 2:
            return empty
        comp, _ = self.get_complementary_endpoint_idx(ends)
        pos_t = self._endpoints[ends].detach()
        dirs_t = self._endpoints[comp].detach() - pos_t
        dirs_t = dirs_t / torch.norm(dirs_t, dim=1, keepdim=True)
        partner = self.strands_info.strand_endpoint_id_to_complementary  # other end of the same strand, per id
        comp_dev = getattr(self, "_comp_dev", None)
        if pos_t.is_cuda and max_num_nn <= 0 and comp_dev is not None and comp_dev.shape[0] == self._endpoints.shape[0] \
                and os.environ.get("HGS_MERGE_SEARCH", "device") == "device":       # (HGS_MERGE_SEARCH=host: the form below, for A/B)
            # everything up to stage 1 on the device: late in Stage III a frame has 5 10^5 strand ends and 3 10^5 candidate pairs,
            # and the host form below -- norm, lexsort, unique over all of them -- took 95 of an event's 125 ms
            cand = self._merge_candidates_device(pos_t, dirs_t, ends, comp_dev, float(dir_th))
            if cand.shape[0] == 0:
                return empty
        else:
            pos, dirs, ends_np = pos_t.cpu().numpy(), dirs_t.cpu().numpy(), ends.cpu().numpy()
            if pos_t.is_cuda:
                # candidate search on the GPU (hgs_radius_pairs: distance + direction test, brute force over the strand ends)
                a, b = self._radius_pairs_gpu(pos_t, dirs_t, float(self.merge_dist_th), float(dir_th),
                                              bool(self.training_args.bidirectional_merge))
                ok = partner[ends_np[a]] != ends_np[b]
            else:
                pairs = cKDTree(pos).query_pairs(r=float(self.merge_dist_th), output_type="ndarray")
                if pairs.shape[0] == 0:
                    return empty
                a, b = pairs[:, 0], pairs[:, 1]
                ok = partner[ends_np[a]] != ends_np[b]
                dot = -(dirs[a] * dirs[b]).sum(1)                     # directions must be opposite
                if self.training_args.bidirectional_merge:
                    dot = np.abs(dot)
                ok &= dot >= dir_th
            a, b = a[ok], b[ok]
            if a.shape[0] == 0:
                return empty
            dist = np.linalg.norm(pos[a] - pos[b], axis=1)
            order = np.lexsort((b, a, dist))          # deterministic whatever order the candidates were found in
            a, b, dist = a[order], b[order], dist[order]
            if max_num_nn > 0:  # cap candidates per point, nearest first
                order = np.argsort(dist, kind="stable")
                a, b, dist = a[order], b[order], dist[order]
                seen = {}
                keep = np.ones(len(a), bool)
                for i, (x, y) in enumerate(zip(a, b)):
                    if seen.get(x, 0) >= max_num_nn or seen.get(y, 0) >= max_num_nn:
                        keep[i] = False
                    else:
                        seen[x] = seen.get(x, 0) + 1
                        seen[y] = seen.get(y, 0) + 1
                a, b, dist = a[keep], b[keep], dist[keep]
            order = np.argsort(dist, kind="stable")
            cand = np.stack([ends_np[a[order]], ends_np[b[order]]], 1)
            # stage 1 (reference remove_duplicate_endpoint_rows): both ids must occur here for the first time in the
            # distance-sorted candidate list -- ids of rejected rows count as seen too, so the test does not depend on what was
            # accepted: a row survives iff it holds the first occurrence of both its ids (one np.unique instead of a Python loop
            # over every candidate: thousands per merge event, a few hundred survivors)
            flat = cand.reshape(-1)
            _, first_at = np.unique(flat, return_index=True)
            is_first = np.zeros(flat.shape[0], bool)
            is_first[first_at] = True
            cand = cand[is_first.reshape(-1, 2).all(axis=1)]
        # stage 2 (reference remove_complementary_rows): never merge both ends of one strand in the same round.  The reference walks
        # the rows in order with a set of blocked ids -- the strand partners of the ids of every row it has kept -- and skips a row that
        # holds a blocked id.  Id p is blocked exactly when the row that holds partner(p) came earlier and was kept (partner is an
        # involution on strand ends; after stage 1 an id sits in at most one row): every row depends on at most two EARLIER rows,
        # and the walk is resolved in a few vectorised rounds -- a row is kept once all the earlier rows it depends on are skipped,
        # skipped once one of them is kept.  (The loop over thousands of Python lists also fed the cyclic garbage collector: a
        # Stage-III event's 10^4 temporaries were promoted to its oldest generation, and five full collections of ~100 ms each hit
        # every 2000 iterations: tools/dev/gc_pauses.py.)
        out = cand[self._keep_rows_whose_strand_partners_are_free(cand, partner)]
        if out.shape[0] == 0:
            return empty
        return torch.as_tensor(np.ascontiguousarray(out, dtype=np.int64), device=self.device)

    @staticmethod
    def _keep_rows_whose_strand_partners_are_free(cand, partner):
        """Boolean mask over the rows of `cand` ([m, 2] ids, every id in at most one row): the rows the reference's in-order walk
        keeps (remove_complementary_rows: skip a row that holds the strand partner of an id of a row kept before).
        tests/test_host_helpers_cpu.py compares it with that walk on random tables."""
        m = cand.shape[0]
        partner_of = partner[cand]                # (other end of the same strand, per surviving id)
        row_of = np.full(partner.shape[0], -1, np.int64)
        rows = np.arange(m, dtype=np.int64)
        row_of[cand[:, 0]] = rows
        row_of[cand[:, 1]] = rows
        dep = np.where(partner_of >= 0, row_of[np.maximum(partner_of, 0)], -1)
        has = (dep >= 0) & (dep < rows[:, None])  # only rows in front of this one can have blocked it
        dep = np.where(has, dep, 0)
        state = np.zeros(m, np.int8)              # 0 undecided, 1 kept, 2 skipped
        for _ in range(64):
            und = state == 0
            if not und.any():
                return state == 1
            st = np.where(has, state[dep], 2)
            blocked = (st == 1).any(axis=1)
            free = (st == 2).all(axis=1)
            state[und & blocked] = 2
            state[und & ~blocked & free] = 1
        # a dependency chain longer than the rounds above (rows that follow each other end to end IN the table's order: a round
        # settles one link of it): the rest by the walk itself, over the undecided rows only
        for r in np.nonzero(state == 0)[0].tolist():
            d0, d1 = (int(dep[r, 0]) if has[r, 0] else -1), (int(dep[r, 1]) if has[r, 1] else -1)
            state[r] = 2 if (d0 >= 0 and state[d0] == 1) or (d1 >= 0 and state[d1] == 1) else 1
        return state == 1

    def _merge_candidates_device(self, pos, dirs, ends, comp_dev, dir_th):
        """The candidate rows of compute_endpoint_pair_to_merge after its stage 1, [K, 2] endpoint ids on the host, from device
        operations only: hgs_radius_pairs, the same-strand filter (comp_dev: strand end -> other end), the distances with numpy's
        expression and association (sqrt((dx dx + dy dy) + dz dz) in float32: the same bits as np.linalg.norm over three
        components), the order (distance, a, b) by two sorts (a sort on the unique key a n + b, then a stable one on the distance
        = np.lexsort((b, a, dist))), and the first-occurrence rule by a scatter-min of the row positions (order-independent, like
        remove_duplicate_endpoint_rows).  tests/test_gpu_knn.py compares the merged pairs with the CPU model's."""
        import hgs_runtime as rt
        n, dev = pos.shape[0], pos.device
        pos = rt.require_gpu_tensor(pos, "positions", torch.float32)
        dirs = rt.require_gpu_tensor(dirs, "directions", torch.float32)
        order = torch.argsort(pos[:, 0])
        ps, ds = pos[order].contiguous(), dirs[order].contiguous()
        cap = max(4 * n, 1024)
        while True:
            pairs = torch.empty((cap, 2), dtype=torch.int32, device=dev)
            dist_k = torch.empty((cap,), dtype=torch.float32, device=dev)
            count = torch.zeros(1, dtype=torch.int32, device=dev)
            with torch.cuda.device(dev):
                rt.check(rt.lib().hgs_radius_pairs(rt.current_stream(), n, rt.ptr(ps), rt.ptr(ds), float(self.merge_dist_th), float(dir_th),
                                                   int(bool(self.training_args.bidirectional_merge)), cap, rt.ptr(pairs), rt.ptr(dist_k),
                                                   rt.ptr(count), 1))
            found = int(count.item())
            if found <= cap:
                break
            cap = found
        empty = np.zeros((0, 2), np.int64)
        if found == 0:
            return empty
        ia, ib = order[pairs[:found, 0].long()], order[pairs[:found, 1].long()]      # back to the caller's indices
        a, b = torch.minimum(ia, ib), torch.maximum(ia, ib)
        ea, eb = ends[a], ends[b]
        ok = comp_dev[ea].long() != eb                   # not the two ends of one strand
        a, b, ea, eb = a[ok], b[ok], ea[ok], eb[ok]
        if a.numel() == 0:
            return empty
        d = pos[a] - pos[b]
        dist = torch.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])
        o1 = torch.argsort(a * n + b)
        o = o1[torch.argsort(dist[o1], stable=True)]
        cand = torch.stack([ea[o], eb[o]], dim=1)
        flat = cand.reshape(-1)
        at = torch.arange(flat.shape[0], dtype=torch.long, device=dev)
        first = torch.full((int(self._endpoints.shape[0]),), flat.shape[0], dtype=torch.long, device=dev)
        first.scatter_reduce_(0, flat, at, reduce="amin", include_self=True)
        keep = (first[flat] == at).reshape(-1, 2).all(dim=1)
        return cand[keep].cpu().numpy()

    @staticmethod
    def _radius_pairs_gpu(pos, dirs, radius, min_cos, bidirectional):
        """(a, b) index arrays (a < b) of the strand ends within `radius` whose directions oppose within the angle."""
        import hgs_runtime as rt
        pos = rt.require_gpu_tensor(pos, "positions", torch.float32)
        dirs = rt.require_gpu_tensor(dirs, "directions", torch.float32)
        n, dev = pos.shape[0], pos.device
        order = torch.argsort(pos[:, 0])                 # ascending x: lets the search stop a radius beyond each block
        pos, dirs = pos[order].contiguous(), dirs[order].contiguous()
        order = order.cpu().numpy()
        cap = max(4 * n, 1024)
        while True:
            pairs = torch.empty((cap, 2), dtype=torch.int32, device=dev)
            dist = torch.empty((cap,), dtype=torch.float32, device=dev)
            count = torch.zeros(1, dtype=torch.int32, device=dev)
            with torch.cuda.device(dev):
                rt.check(rt.lib().hgs_radius_pairs(rt.current_stream(), n, rt.ptr(pos), rt.ptr(dirs), float(radius), float(min_cos),
                                                   int(bidirectional), cap, rt.ptr(pairs), rt.ptr(dist), rt.ptr(count), 1))
            found = int(count.item())
            if found <= cap:
                p = order[pairs[:found].cpu().numpy().astype(np.int64)]      # back to the caller's indices, a < b
                return p.min(axis=1), p.max(axis=1)
            cap = found      # dense cluster of strand ends: run again with room for every pair

    def merge_endpoint_pairs(self, endpoint_pair_index):
        """Fuse each pair of strand ends into ONE new endpoint at their midpoint: the two end segments are re-created
        attached to it (attributes cloned), the old ones and the two old endpoints disappear (reference :619-706)."""
        if endpoint_pair_index.shape[0] == 0:
            return
        pos = self._endpoints[endpoint_pair_index].detach()
        c1, r1 = self.get_complementary_endpoint_idx(endpoint_pair_index[:, 0])
        c2, r2 = self.get_complementary_endpoint_idx(endpoint_pair_index[:, 1])
        new_ep = 0.5 * pos[:, 1] + 0.5 * pos[:, 0]
        new_ids = torch.arange(new_ep.shape[0], device=self.device) + self.endpoint_pairs.max() + 1
        remap = torch.arange(self._endpoints.shape[0], device=self.device)
        remap[endpoint_pair_index[:, 0]] = new_ids
        remap[endpoint_pair_index[:, 1]] = new_ids
        new_pairs = torch.cat((torch.stack((remap[c1], new_ids), 1), torch.stack((new_ids, remap[c2]), 1)), dim=0)
        rows = torch.cat((r1, r2))
        self.cat_segments(new_pairs, new_ep, self._features_dc[rows].detach(), self._features_rest[rows].detach(),
                          self._opacity[rows].detach(), self._mask[rows].detach(), self._width[rows].detach())
        prune = torch.zeros(self.endpoint_pairs.shape[0], device=self.device, dtype=torch.bool)
        prune[r1] = True
        prune[r2] = True
        self.prune_segments(prune)
