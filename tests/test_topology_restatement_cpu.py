"""The strand topology operators against (1) a scalar restatement of the reference's statements and (2) the reference's own
code, run on the CPU of the authoring container (the fixture's tests are at the end of this file).

scene/hair_topology.py runs clone / split / merge_collapsed / prune (reference scene/hair_gaussian_model.py:788-1077) as
vectorised tensor code with id remapping, endpoint compaction and optimizer-state surgery.  (1) restates the same statements
a SECOND time, in the plainest form available -- Python lists and loops over one segment at a time, one function per
reference statement block, each citing its lines -- and demands that random models come out of the vectorised operators
IDENTICAL to the restatement: the same endpoint_pairs (ids and order), endpoint positions, per-segment attributes, Adam
moments and statistics.  Element-wise quantities the selections read (scales, opacity, mask, segment lengths) are taken from
the model's getters, which have their own tests; what is pinned here is the topology arithmetic.  (2) holds what the
REFERENCE's HairGaussianModel (device="cpu") made of the very same random models (tests/golden/make_ref_topology_pins.py):
reference run == restatement == product.
"""
import copy

import numpy as np
import pytest
import torch

from arguments import OptimizationParams


# ---- the restatement: a model is a dict of python lists ---------------------------------------------------------------------
SEG_KEYS = ("f_dc", "f_rest", "opacity", "mask", "width")        # per-segment parameter groups
ALL_KEYS = ("endpoints",) + SEG_KEYS


def snapshot(m):
    """The builder's model -> lists: pairs [[a, b]], per-group rows of (value, exp_avg, exp_avg_sq), statistics."""
    st = {"pairs": [[int(a), int(b)] for a, b in m.endpoint_pairs.tolist()]}
    for g in m.optimizer.param_groups:
        p = g["params"][0]
        s = m.optimizer.state.get(p, {})
        ea = s.get("exp_avg", torch.zeros_like(p)).detach()
        eq = s.get("exp_avg_sq", torch.zeros_like(p)).detach()
        st[g["name"]] = [(p.detach()[i].clone(), ea[i].clone(), eq[i].clone()) for i in range(p.shape[0])]
    st["grad_accum"] = [float(x) for x in m.xyz_gradient_accum.reshape(-1).tolist()]
    st["denom"] = [float(x) for x in m.denom.reshape(-1).tolist()]
    st["max_radii2D"] = [float(x) for x in m.max_radii2D.reshape(-1).tolist()]
    return st


def r_prune_segments(st, prune):
    """prune_segments (reference :575-617): drop the flagged segments, then every endpoint no remaining segment references,
    renumbering the kept endpoints in ascending order of their old ids; optimizer rows follow their parameters."""
    keep = [i for i, p in enumerate(prune) if not p]
    st["pairs"] = [st["pairs"][i] for i in keep]
    used = sorted({e for pr in st["pairs"] for e in pr})
    new_id = {old: new for new, old in enumerate(used)}
    st["pairs"] = [[new_id[a], new_id[b]] for a, b in st["pairs"]]
    st["endpoints"] = [st["endpoints"][e] for e in used]
    for k in SEG_KEYS:
        st[k] = [st[k][i] for i in keep]
    for k in ("grad_accum", "denom", "max_radii2D"):
        st[k] = [st[k][i] for i in keep]


def r_cat_segments(st, new_pairs, new_endpoints, new_rows):
    """cat_segments (reference :536-573): append endpoints and segments; new rows start with zero Adam moments; the
    densification statistics are reset for ALL segments (:561-571)."""
    st["pairs"] += [list(p) for p in new_pairs]
    st["endpoints"] += [(v.clone(), torch.zeros_like(v), torch.zeros_like(v)) for v in new_endpoints]
    for k in SEG_KEYS:
        st[k] += [(v.clone(), torch.zeros_like(v), torch.zeros_like(v)) for v in new_rows[k]]
    n = len(st["pairs"])
    st["grad_accum"], st["denom"], st["max_radii2D"] = [0.0] * n, [0.0] * n, [0.0] * n


def r_clone(st, grads, max_scale, thr_grad, thr_size):
    """clone_strategy (reference :915-967)."""
    sel = [i for i in range(len(st["pairs"])) if grads[i] >= thr_grad and max_scale[i] <= thr_size]
    nxt = max(max(p) for p in st["pairs"]) + 1
    new_pairs, new_eps, rows = [], [], {k: [] for k in SEG_KEYS}
    for j, i in enumerate(sel):
        a, b = st["pairs"][i]
        new_eps += [st["endpoints"][a][0], st["endpoints"][b][0]]
        new_pairs.append([nxt + 2 * j, nxt + 2 * j + 1])
        for k in SEG_KEYS:
            rows[k].append(st[k][i][0])
    r_cat_segments(st, new_pairs, new_eps, rows)
    return len(sel)


def r_split(st, grads, max_scale, seg_len, mask_prob, centre, thr_grad, thr_size, max_len, fg_th):
    """split_strategy (reference :828-913): grads are padded with zeros for the segments clone appended (:836-838)."""
    n = len(st["pairs"])
    g = list(grads) + [0.0] * (n - len(grads))
    sel = [i for i in range(n) if ((g[i] >= thr_grad and max_scale[i] > thr_size) or seg_len[i] >= max_len) and mask_prob[i] > fg_th]
    nxt = max(max(p) for p in st["pairs"]) + 1
    first, second, mids, rows = [], [], [], {k: [] for k in SEG_KEYS}
    for j, i in enumerate(sel):
        a, b = st["pairs"][i]
        first.append([a, nxt + j])
        second.append([nxt + j, b])
        mids.append(centre[i])
    for k in SEG_KEYS:
        rows[k] = [st[k][i][0] for i in sel] * 2          # .repeat(2, ...): the selected rows, then the selected rows again
    r_cat_segments(st, first + second, mids, rows)
    prune = [False] * len(st["pairs"])
    for i in sel:
        prune[i] = True
    r_prune_segments(st, prune)
    return len(sel)


def r_remove_duplicate_rows(rows):
    """remove_duplicate_endpoint_rows (reference :712-728): a row survives iff BOTH its ids occur there for the first time
    in row-major order of the flattened list."""
    seen, first = set(), []
    for a, b in rows:
        fa = a not in seen
        seen.add(a)
        fb = b not in seen
        seen.add(b)
        first.append(fa and fb)
    return first


def r_merge_collapsed(st, seg_len_of, fg_of, min_val):
    """merge_collapsed_segments (reference :969-1018), one round per loop trip; `seg_len_of` / `fg_of` evaluate the current
    model (lengths from the endpoint positions, foreground from opacity and mask)."""
    total = 0
    while True:
        n = len(st["pairs"])
        lens, fg = seg_len_of(st), fg_of(st)
        cand = [i for i in range(n) if lens[i] < min_val or not fg[i]]
        count = {}
        for a, b in st["pairs"]:
            count[a] = count.get(a, 0) + 1
            count[b] = count.get(b, 0) + 1
        cand = [i for i in cand if count[st["pairs"][i][0]] != 1 and count[st["pairs"][i][1]] != 1]   # both ends interior joints
        keep = r_remove_duplicate_rows([st["pairs"][i] for i in cand])
        cand = [i for i, k in zip(cand, keep) if k]
        to_merge = [list(st["pairs"][i]) for i in cand]                    # ids BEFORE the prune below (:1000-1004)
        prune = [False] * n
        for i in cand:
            prune[i] = True
        r_prune_segments(st, prune)
        if to_merge:                                                       # ... applied to the ids AFTER it
            top = max(max(p) for p in st["pairs"]) + 1
            mapping = list(range(top))
            for a, b in to_merge:
                if b < top:
                    mapping[b] = a
                else:                                                      # (an index past the table: the reference would raise)
                    raise IndexError("merge pair beyond the compacted ids")
            st["pairs"] = [[mapping[a], mapping[b]] for a, b in st["pairs"]]
        r_prune_segments(st, [False] * len(st["pairs"]))
        total += len(to_merge)
        if not to_merge:
            return total


def r_prune(st, seg_len, opacity, max_scale, mask_prob, min_val, op_th, fg_th, extent, max_screen_size, avoid_connected):
    """prune_strategy (reference :1020-1077)."""
    n = len(st["pairs"])
    prune = [seg_len[i] < min_val or opacity[i] < op_th for i in range(n)]
    if max_screen_size and extent != 0.0:
        prune = [prune[i] or max_scale[i] > 0.1 * extent for i in range(n)]
    if avoid_connected and any(prune):
        count = {}
        for a, b in st["pairs"]:
            count[a] = count.get(a, 0) + 1
            count[b] = count.get(b, 0) + 1
        for i in range(n):
            a, b = st["pairs"][i]
            is_end = count[a] == 1 or count[b] == 1
            if not (is_end or mask_prob[i] < fg_th):
                prune[i] = False
    k = sum(prune)
    if 0 < k < n:
        r_prune_segments(st, prune)
    return k


# ---- element-wise inputs of the selections, from a restated model -----------------------------------------------------------
def _rebuild(m0, st):
    """A model object carrying the restated state (for the getters: scales, opacity, mask, lengths)."""
    m = copy.deepcopy(m0)
    m.endpoint_pairs = torch.tensor(st["pairs"], dtype=torch.long).reshape(-1, 2)
    vals = {k: torch.stack([r[0] for r in st[k]]) for k in ALL_KEYS}
    m._endpoints = torch.nn.Parameter(vals["endpoints"])
    m._features_dc, m._features_rest = torch.nn.Parameter(vals["f_dc"]), torch.nn.Parameter(vals["f_rest"])
    m._opacity, m._mask, m._width = torch.nn.Parameter(vals["opacity"]), torch.nn.Parameter(vals["mask"]), torch.nn.Parameter(vals["width"])
    return m


def _elementwise(m0, st):
    m = _rebuild(m0, st)
    with torch.no_grad():
        return dict(max_scale=m.get_scaling.max(dim=1).values.tolist(), seg_len=m._segment_lengths().tolist(),
                    opacity=m.get_opacity.reshape(-1).tolist(), mask_prob=m.get_mask.reshape(-1).tolist(),
                    centre=[c.clone() for c in m.get_xyz], fg=m.compute_foreground_mask().tolist())


def r_densification(m0, st, extent, max_screen_size):
    """densification (reference :788-817): clone, split, merge_collapsed, prune, on the restated state."""
    ta = m0.training_args
    # :802-803: accum / denom with IEEE semantics (x / 0 = inf for x > 0, 0 / 0 = nan), nan -> 0
    grads = [(a / d) if d != 0 else (float("inf") if a > 0 else float("nan")) for a, d in zip(st["grad_accum"], st["denom"])]
    grads = [0.0 if g != g else g for g in grads]
    thr_size = ta.percent_dense * extent
    ew = _elementwise(m0, st)
    info = {"clone": r_clone(st, grads, ew["max_scale"], ta.densify_grad_threshold, thr_size)}
    ew = _elementwise(m0, st)
    info["split"] = r_split(st, grads, ew["max_scale"], ew["seg_len"], ew["mask_prob"], ew["centre"], ta.densify_grad_threshold,
                            thr_size, float(m0.max_segment_length), m0.foreground_binarization_th)
    info["merge_collapsed"] = r_merge_collapsed(st, lambda s: _elementwise(m0, s)["seg_len"], lambda s: _elementwise(m0, s)["fg"], m0.min_val)
    ew = _elementwise(m0, st)
    info["prune_total"] = r_prune(st, ew["seg_len"], ew["opacity"], ew["max_scale"], ew["mask_prob"], m0.min_val, m0.opacity_th,
                                  m0.foreground_binarization_th, extent, max_screen_size, True)
    return info


# ---- random models -----------------------------------------------------------------------------------------------------------
def _random_model(seed, device="cpu"):
    from scene.hair_gaussian_model import HairGaussianModel
    from synthetic import strand_polylines
    rng = np.random.default_rng(seed)
    S, V = int(rng.integers(4, 9)), int(rng.integers(5, 10))
    m = HairGaussianModel.from_strands(strand_polylines(S, V, seed=seed), device=device)
    opt = OptimizationParams()
    m.training_setup(opt)
    P = m.endpoint_pairs.shape[0]
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        # distinct attributes per segment, so that a row attached to the wrong segment shows
        m._features_dc.add_(torch.randn(m._features_dc.shape, generator=g).to(device) * 0.1)
        m._width.add_(torch.randn(m._width.shape, generator=g).to(device) * 0.3)
        m._opacity.copy_(torch.randn((P, 1), generator=g).to(device) * 2.0)
        low = torch.rand(P, generator=g) < 0.15
        m._opacity[low.to(device)] = -8.0                                  # transparent: prune / background candidates
        bg = torch.rand(P, generator=g) < 0.15
        m._mask[bg.to(device)] = -3.0                                      # background: merge-collapsed / prune candidates
        # collapse some interior segments: the second endpoint onto the first
        for i in torch.nonzero(torch.rand(P, generator=g) < 0.12).flatten().tolist():
            a, b = m.endpoint_pairs[i].tolist()
            m._endpoints[b] = m._endpoints[a]
    # Adam moments that tell rows apart, statistics with a spread of gradients
    for gp in m.optimizer.param_groups:
        p = gp["params"][0]
        m.optimizer.state[p] = {"step": torch.tensor(3.0), "exp_avg": torch.randn(p.shape, generator=g).to(device),
                                "exp_avg_sq": torch.rand(p.shape, generator=g).to(device)}
    m.denom = torch.randint(0, 3, (P, 1), generator=g).float().to(device)
    m.xyz_gradient_accum = (torch.rand((P, 1), generator=g) * 6e-4).to(device) * m.denom.clamp(min=1)
    m.max_radii2D = torch.rand(P, generator=g).to(device) * 30
    m.compute_strands_info()
    return m


def _assert_same(m, st, what):
    got = snapshot(m)
    assert got["pairs"] == st["pairs"], what
    for k in ALL_KEYS:
        assert len(got[k]) == len(st[k]), (what, k)
        for i, (a, b) in enumerate(zip(got[k], st[k])):
            for j, name in enumerate(("value", "exp_avg", "exp_avg_sq")):
                assert torch.equal(a[j].cpu(), b[j].cpu()), (what, k, i, name)
    for k in ("grad_accum", "denom", "max_radii2D"):
        assert got[k] == st[k], (what, k)


@pytest.mark.parametrize("seed", range(12))
@pytest.mark.parametrize("extent", [1e-3, 0.05, 50.0])    # tiny: every gradient splits; mid: mixed; huge: every gradient clones
def test_densification_equals_the_scalar_restatement(seed, extent):
    m = _random_model(seed)
    st = snapshot(m)
    m0 = copy.deepcopy(m)
    info_r = r_densification(m0, st, extent, None)

    class Info:                      # (the operators fill a dict the way the reference fills training_info.densification_info)
        densification_info = {}
    info = Info()
    info.densification_info = {}
    m.densification(extent, None, info)
    assert {k: info.densification_info[k] for k in info_r} == info_r
    _assert_same(m, st, f"densification seed {seed} extent {extent}")


@pytest.mark.parametrize("seed", range(8))
def test_each_operator_alone_equals_its_restatement(seed):
    """clone, split, merge_collapsed and prune one at a time on the same random model (the screen-size branch of prune on)."""
    extent = 0.02
    for op in ("clone", "split", "merge_collapsed", "prune"):
        m = _random_model(100 + seed)
        st = snapshot(m)
        m0 = copy.deepcopy(m)
        ta = m.training_args
        grads_t = m.xyz_gradient_accum / m.denom
        grads_t[grads_t.isnan()] = 0.0
        grads = [float(x) for x in grads_t.reshape(-1).tolist()]
        ew = _elementwise(m0, st)
        if op == "clone":
            r_clone(st, grads, ew["max_scale"], ta.densify_grad_threshold, ta.percent_dense * extent)
            m.clone_strategy(grads_t, extent, {})
        elif op == "split":
            r_split(st, grads, ew["max_scale"], ew["seg_len"], ew["mask_prob"], ew["centre"], ta.densify_grad_threshold,
                    ta.percent_dense * extent, float(m.max_segment_length), m.foreground_binarization_th)
            m.split_strategy(grads_t, extent, {})
        elif op == "merge_collapsed":
            r_merge_collapsed(st, lambda s: _elementwise(m0, s)["seg_len"], lambda s: _elementwise(m0, s)["fg"], m.min_val)
            m.merge_collapsed_segments({})
        else:
            r_prune(st, ew["seg_len"], ew["opacity"], ew["max_scale"], ew["mask_prob"], m.min_val, m.opacity_th,
                    m.foreground_binarization_th, extent, 20, True)
            m.prune_strategy(extent, 20, {}, avoid_connected=True)
        _assert_same(m, st, f"{op} seed {seed}")


# ---- merging: the greedy matching of strand ends (reference :1205-1362) ---------------------------------------------------------
def r_pairs_to_merge(m):
    """compute_endpoint_pair_to_merge restated with loops: every strand end of a foreground segment, its direction towards
    the other endpoint of its segment; candidates = the OTHER ends within merge_dist_th (brute force instead of the kd-tree)
    that are neither itself nor the other end of its own strand and whose direction opposes its own within merge_angle_th;
    both directed rows (i, j) and (j, i) as the reference emits them; ascending distance; a row survives
    remove_duplicate_endpoint_rows iff both ids are new in row-major order, then remove_complementary_rows walks the
    survivors and drops a row touching an end whose strand's other end was merged before it."""
    pairs = m.endpoint_pairs.tolist()
    ep = m._endpoints.detach().double().numpy()
    count = {}
    for a, b in pairs:
        count[a] = count.get(a, 0) + 1
        count[b] = count.get(b, 0) + 1
    fg = m.compute_foreground_mask().tolist()
    fg_ids = {e for (a, b), f in zip(pairs, fg) if f for e in (a, b)}
    ends = sorted(e for e, c in count.items() if c == 1 and e in fg_ids)
    other = {}                                    # end -> the other endpoint of ITS segment (the last row holding it: :730-752)
    for a, b in pairs:
        other[a], other[b] = b, a
    partner = m.strands_info.strand_endpoint_id_to_complementary
    dirs = {}
    for e in ends:
        d = ep[other[e]] - ep[e]
        dirs[e] = d / np.linalg.norm(d)
    th, cos_th = float(m.merge_dist_th), float(np.cos(np.deg2rad(m.merge_angle_th)))
    rows = []
    for i in ends:
        for j in ends:
            if j == i or j == int(partner[i]):
                continue
            dist = float(np.linalg.norm(ep[i] - ep[j]))
            if dist > th:
                continue
            dot = float(-(dirs[i] @ dirs[j]))
            if m.training_args.bidirectional_merge:
                dot = abs(dot)
            if dot >= cos_th:
                rows.append((dist, i, j))
    rows.sort(key=lambda r: r[0])
    keep = r_remove_duplicate_rows([(i, j) for _, i, j in rows])
    rows = [(i, j) for (_, i, j), k in zip(rows, keep) if k]
    disabled, out = set(), []
    for i, j in rows:
        if i in disabled or j in disabled:
            continue
        disabled.add(int(partner[i]))
        disabled.add(int(partner[j]))
        out.append(frozenset((i, j)))
    return set(out)


def _cut_model(seed, bidirectional):
    """Long curves cut into pieces whose ends sit 0-3 mm apart (the threshold starts at 2 mm) with jittered directions and some
    background pieces, in shuffled order: what Stage II (merge.py) starts from."""
    from scene.hair_gaussian_model import HairGaussianModel
    from synthetic import strand_polylines
    rng = np.random.default_rng(seed)
    curves = strand_polylines(5, 41, seed=seed)                       # [5, 41, 3], 2.5 mm steps
    pieces = []
    for c in curves:
        for k in range(5):
            seg = c[8 * k: 8 * k + 9].copy()                          # 9 vertices; neighbours share a vertex ...
            seg += rng.normal(scale=4e-4, size=(1, 3))                # ... until each piece is shifted by ~0.7 mm
            seg[-1] += rng.normal(scale=3e-4, size=3)                 # and its ends bent a little
            seg[0] += rng.normal(scale=3e-4, size=3)
            pieces.append(seg)
    order = rng.permutation(len(pieces))
    pts = np.stack([pieces[i] for i in order]).astype(np.float32)
    m = HairGaussianModel.from_strands(pts, device="cpu", ref_strand_root=curves[:, 0])
    opt = OptimizationParams()
    opt.bidirectional_merge = bidirectional
    m.training_setup(opt)
    with torch.no_grad():
        P = m.endpoint_pairs.shape[0]
        bg = torch.from_numpy(rng.random(P) < 0.08)
        m._mask[bg] = -3.0
    g = torch.Generator().manual_seed(seed)
    for gp in m.optimizer.param_groups:                               # Adam moments that tell rows apart
        p = gp["params"][0]
        m.optimizer.state[p] = {"step": torch.tensor(3.0), "exp_avg": torch.randn(p.shape, generator=g), "exp_avg_sq": torch.rand(p.shape, generator=g)}
    m.compute_strands_info()
    return m


@pytest.mark.parametrize("seed", range(10))
@pytest.mark.parametrize("bidirectional", [False, True])
def test_pairs_to_merge_equal_the_scalar_restatement(seed, bidirectional):
    """The vectorised search (kd-tree pairs, lexsort, one greedy loop) selects exactly the restatement's pairs on cut-up curves."""
    m = _cut_model(seed, bidirectional)
    got = m.compute_endpoint_pair_to_merge()
    got = {frozenset(int(x) for x in row) for row in got.tolist()}
    want = r_pairs_to_merge(m)
    assert got == want and len(want) >= 5


# ---- the reference's own operators, executed on CPU (tests/golden/make_ref_topology_pins.py) ------------------------------------
# The fixture holds, for the same random models as above, what /root/reference's HairGaussianModel (device="cpu") made of them:
# the third leg -- reference run == scalar restatement == vectorised product.
import os

_PINS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_topology_pins.npz")
_GROUPS = ("endpoints", "f_dc", "f_rest", "opacity", "mask", "width")


@pytest.fixture(scope="module")
def ref_pins():
    assert os.path.exists(_PINS), "tests/golden/ref_topology_pins.npz is part of the repository"
    return np.load(_PINS)


def _model_for_pin(pins, seed):
    """This repository's random model of `seed`, checked to BE the state the reference was given."""
    m = _random_model(seed)
    k = f"s{seed}_"
    # training_setup once more, on the FINAL random state (the reference's ran on it too: max_segment_length, :268-283, is
    # derived there from the foreground endpoints), then the moments and statistics again
    keep = {g["name"]: dict(m.optimizer.state[g["params"][0]]) for g in m.optimizer.param_groups}
    stats = (m.xyz_gradient_accum, m.denom, m.max_radii2D)
    m.training_setup(OptimizationParams())
    for g in m.optimizer.param_groups:
        m.optimizer.state[g["params"][0]] = keep[g["name"]]
    m.xyz_gradient_accum, m.denom, m.max_radii2D = stats
    m.compute_strands_info()
    assert np.array_equal(m.endpoint_pairs.numpy(), pins[k + "pairs"])
    for g in m.optimizer.param_groups:
        p = g["params"][0]
        assert np.array_equal(p.detach().numpy(), pins[k + g["name"]]), g["name"]
        assert np.array_equal(m.optimizer.state[p]["exp_avg"].numpy(), pins[k + g["name"] + "_exp_avg"])
    assert np.array_equal(m.xyz_gradient_accum.numpy(), pins[k + "grad_accum"]) and np.array_equal(m.denom.numpy(), pins[k + "denom"])
    # derived by training_setup on both sides from the same endpoints: the reference's own value
    if (k + "ref_max_segment_length") in pins.files:
        assert float(m.max_segment_length) == float(pins[k + "ref_max_segment_length"])
    return m


def _assert_equals_pin(m, pins, key, what):
    assert np.array_equal(m.endpoint_pairs.numpy(), pins[key + "pairs"]), (what, "endpoint_pairs")
    for g in m.optimizer.param_groups:
        p = g["params"][0]
        st = m.optimizer.state.get(p, {})
        for name, got in ((g["name"], p.detach()), (g["name"] + "_exp_avg", st.get("exp_avg", torch.zeros_like(p))),
                          (g["name"] + "_exp_avg_sq", st.get("exp_avg_sq", torch.zeros_like(p)))):
            want = pins[key + name]
            assert tuple(got.shape) == want.shape, (what, name, tuple(got.shape), want.shape)
            assert np.array_equal(got.detach().numpy(), want), (what, name)
    for name, got in (("grad_accum", m.xyz_gradient_accum), ("denom", m.denom), ("max_radii2D", m.max_radii2D)):
        assert np.array_equal(got.numpy().reshape(-1), pins[key + name].reshape(-1)), (what, name)


def _assert_strands_equal_pin(si, pins, key, what):
    off = pins[key + "strand_offsets"]
    assert len(si.list_strands) == len(off) - 1, (what, "number of strands")
    pts, ids = pins[key + "strand_points"], pins[key + "strand_segment_ids"]
    for s in range(len(off) - 1):
        assert np.array_equal(np.asarray(si.list_strands[s]).reshape(-1, 2), pts[off[s]:off[s + 1]]), (what, "strand", s)
        assert np.array_equal(np.asarray(si.list_strands_segments_id[s]).reshape(-1), ids[off[s]:off[s + 1]]), (what, "segment ids", s)
    assert np.array_equal(np.asarray(si.id_to_strand_id), pins[key + "id_to_strand_id"]), (what, "id_to_strand_id")
    assert np.array_equal(np.asarray(si.strand_endpoint_id_to_complementary), pins[key + "complementary"]), (what, "complementary")


def _assert_equivalent_pin(m, pins, key, what):
    """After merge_endpoint_pairs the reference's model is determined only up to (a) the order of the two segments a merge
    creates within the appended block and (b) their orientation: both follow from which endpoint of a pair is column 0, and
    the reference gets that from torch.sort's handling of EXACT ties (every candidate is found twice, once from each end, at
    the same distance; scene/hair_gaussian_model.py:1334-1340 sorts them with the default, unstable sort and keeps the first).
    Compared: endpoints (ids, positions, moments) as they are; segments as a set keyed by their unordered endpoint pair with
    every attribute and moment; statistics."""
    def rows(pairs, groups):
        out = {}
        for i, (a, b) in enumerate(pairs.tolist()):
            k = (min(a, b), max(a, b))
            assert k not in out, (what, "two segments on one endpoint pair", k)
            out[k] = tuple(g[i].tobytes() for g in groups)
        return out
    names = [g["name"] for g in m.optimizer.param_groups if g["name"] != "endpoints"]
    got_groups, want_groups = [], []
    for g in m.optimizer.param_groups:
        p = g["params"][0]
        st = m.optimizer.state.get(p, {})
        trio = (p.detach().numpy(), st.get("exp_avg", torch.zeros_like(p)).numpy(), st.get("exp_avg_sq", torch.zeros_like(p)).numpy())
        want = (pins[key + g["name"]], pins[key + g["name"] + "_exp_avg"], pins[key + g["name"] + "_exp_avg_sq"])
        if g["name"] == "endpoints":
            for a, b, n in zip(trio, want, ("value", "exp_avg", "exp_avg_sq")):
                assert np.array_equal(a, b), (what, "endpoints", n)
        else:
            got_groups += list(trio)
            want_groups += list(want)
    assert rows(m.endpoint_pairs.numpy(), got_groups) == rows(pins[key + "pairs"], want_groups), (what, "segments", names)
    for name, got in (("grad_accum", m.xyz_gradient_accum), ("denom", m.denom), ("max_radii2D", m.max_radii2D)):
        assert np.array_equal(np.sort(got.numpy().reshape(-1)), np.sort(pins[key + name].reshape(-1))), (what, name)


def _assert_strands_equivalent_pin(m, pins, key, what):
    si = m.strands_info
    off = pins[key + "strand_offsets"]
    assert len(si.list_strands) == len(off) - 1, (what, "number of strands")
    pts, ids = pins[key + "strand_points"], pins[key + "strand_segment_ids"]
    mine_pairs = m.endpoint_pairs.numpy()
    want_pairs = pins[key + "pairs"]
    fg = m.compute_foreground_mask().numpy()                          # list_strands_segments_id index the foreground rows (:1421-1424)
    mine_fg = mine_pairs[fg]
    for s in range(len(off) - 1):
        assert np.array_equal(np.asarray(si.list_strands[s]).reshape(-1, 2), pts[off[s]:off[s + 1]]), (what, "strand", s)
        # the segment ROWS differ by the permutation above: the rows named must hold the strand's own point pairs
        rows = np.asarray(si.list_strands_segments_id[s]).reshape(-1)
        assert np.array_equal(np.sort(mine_fg[rows], axis=1), np.sort(pts[off[s]:off[s + 1]], axis=1)), (what, "segment rows", s)
    assert np.array_equal(np.asarray(si.id_to_strand_id), pins[key + "id_to_strand_id"]), (what, "id_to_strand_id")
    assert np.array_equal(np.asarray(si.strand_endpoint_id_to_complementary), pins[key + "complementary"]), (what, "complementary")


class _PinInfo:
    def __init__(self):
        self.densification_info = {}


@pytest.mark.parametrize("seed", range(12))
@pytest.mark.parametrize("xi", range(3))
def test_densification_equals_the_reference_run(ref_pins, seed, xi):
    """densification (reference :788-817) as the reference itself executed it, device="cpu": same segments in the same order,
    same endpoint ids and positions, same attributes, same Adam moments and statistics, same counters, same strands_info."""
    extent = float(ref_pins["meta_dens_extents"][xi])
    m = _model_for_pin(ref_pins, seed)
    info = _PinInfo()
    m.densification(extent, None, info)
    key = f"dens_s{seed}_x{xi}_"
    _assert_equals_pin(m, ref_pins, key, f"densification seed {seed} extent {extent}")
    names = [str(n) for n in ref_pins["meta_dens_info_names"]]
    want = {n: int(v) for n, v in zip(names, ref_pins[key + "info"]) if v >= 0}
    assert {n: info.densification_info[n] for n in want} == want
    _assert_strands_equal_pin(m.strands_info, ref_pins, key, f"strands_info after densification seed {seed}")


@pytest.mark.parametrize("seed", [100 + s for s in range(8)])
@pytest.mark.parametrize("op", ["clone", "split", "merge_collapsed", "prune", "prune_free", "clean", "clean_all", "stats"])
def test_each_operator_alone_equals_the_reference_run(ref_pins, seed, op):
    extent = 0.02
    m = _model_for_pin(ref_pins, seed)
    grads = m.xyz_gradient_accum / m.denom
    grads[grads.isnan()] = 0.0
    info = _PinInfo()
    if op == "clone":
        m.clone_strategy(grads, extent, info)
    elif op == "split":
        m.split_strategy(grads, extent, info)
    elif op == "merge_collapsed":
        m.merge_collapsed_segments(info)
    elif op == "prune":
        m.prune_strategy(extent, 20, info, avoid_connected=True)
    elif op == "prune_free":
        m.prune_strategy(extent, 20, info, avoid_connected=False)
    elif op == "clean":
        m.clean_gaussians()
    elif op == "clean_all":
        m.clean_gaussians(avoid_connected=False)
    else:
        vs = torch.zeros((m.endpoint_pairs.shape[0], 3), requires_grad=True)
        vs.grad = torch.from_numpy(ref_pins[f"op_stats_s{seed}_in_vs_grad"].copy())
        radii = torch.from_numpy(ref_pins[f"op_stats_s{seed}_in_radii"].copy())
        m.update_densification_stats(vs, radii, radii > 0)
    key = f"op_{op}_s{seed}_"
    _assert_equals_pin(m, ref_pins, key, f"{op} seed {seed}")
    names = [str(n) for n in ref_pins["meta_dens_info_names"]]
    want = {n: int(v) for n, v in zip(names, ref_pins[key + "info"]) if v >= 0}
    assert {n: info.densification_info[n] for n in want} == want


def _cut_model_for_pin(pins, tag):
    seed, bidir = int(tag[1:]) // 2, bool(int(tag[1:]) % 2)
    m = _cut_model(seed, bidir)
    k = f"{tag}_"
    # training_setup once more on the final state (max_segment_length, :268-283, comes from the FOREGROUND endpoints and the
    # background pieces were marked after it), as the reference's ran
    keep = {g["name"]: dict(m.optimizer.state[g["params"][0]]) for g in m.optimizer.param_groups}
    opt = OptimizationParams()
    opt.bidirectional_merge = bidir
    m.training_setup(opt)
    for g in m.optimizer.param_groups:
        m.optimizer.state[g["params"][0]] = keep[g["name"]]
    m.compute_strands_info()
    assert np.array_equal(m.endpoint_pairs.numpy(), pins[k + "pairs"]) and np.array_equal(m._endpoints.detach().numpy(), pins[k + "endpoints"])
    assert np.array_equal(m._mask.detach().numpy(), pins[k + "mask"])
    assert float(m.max_segment_length) == float(pins[k + "ref_max_segment_length"])
    return m


@pytest.mark.parametrize("case", range(12))
def test_merging_equals_the_reference_run(ref_pins, case):
    """compute_strands_info (:1410-1496), compute_endpoint_pair_to_merge (:1205-1362, both settings of bidirectional_merge),
    merging (:1079-1096), reset_opacity (:1364-1371) and the Stage-II loop of merge.py:113-177 to its fixed point, as the
    reference executed them on cut-up curves (the pieces Stage II starts from)."""
    from merge import merge_rounds
    tag = f"c{case}"
    key = f"merge_{tag}_"
    m = _cut_model_for_pin(ref_pins, tag)
    _assert_strands_equal_pin(m.strands_info, ref_pins, key + "before_", f"strands_info {tag}")
    assert float(m.merge_dist_th) == float(ref_pins[key + "merge_dist_th"]) and float(m.merge_angle_th) == float(ref_pins[key + "merge_angle_th"])
    pairs = m.compute_endpoint_pair_to_merge()
    want = ref_pins[key + "pairs_to_merge"]
    assert want.shape[0] >= 5
    # the same pairs in the same (distance) order; which endpoint of a pair is column 0 is not a property of the reference
    # (see _assert_equivalent_pin)
    assert np.array_equal(np.sort(pairs.numpy().reshape(-1, 2), axis=1), np.sort(want, axis=1)), "pairs to merge"
    info = _PinInfo()
    m.merging(info)
    assert info.densification_info["merge"] == int(ref_pins[key + "info_merge"])
    _assert_equivalent_pin(m, ref_pins, key + "after_", f"merging {tag}")
    _assert_strands_equivalent_pin(m, ref_pins, key + "after_", f"strands_info after merging {tag}")
    m.reset_opacity()
    assert np.array_equal(m._opacity.detach().numpy(), ref_pins[key + "reset_opacity"])
    st = m.optimizer.state.get(m._opacity, {})
    assert np.array_equal(st.get("exp_avg", torch.zeros_like(m._opacity)).numpy(), ref_pins[key + "reset_opacity_exp_avg"])
    assert np.array_equal(st.get("exp_avg_sq", torch.zeros_like(m._opacity)).numpy(), ref_pins[key + "reset_opacity_exp_avg_sq"])
    # Stage II to its fixed point
    ml = _cut_model_for_pin(ref_pins, tag)
    per_round = []
    rounds = merge_rounds(ml, 50, log=lambda msg: per_round.append(int(msg.split("merged ")[1].split(" ")[0])))
    want_rounds = [int(v) for v in ref_pins[key + "loop_pairs_per_round"]]
    assert want_rounds[-1] == 0 and len(want_rounds) >= 2
    assert per_round == want_rounds[:-1] and rounds == len(want_rounds) - 1
    _assert_equivalent_pin(ml, ref_pins, key + "loop_", f"Stage-II loop {tag}")
    _assert_strands_equivalent_pin(ml, ref_pins, key + "loop_", f"strands_info after the Stage-II loop {tag}")
