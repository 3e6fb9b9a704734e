"""Forward-only renders of one model from many views -- the loop of the reference's render.py:57-62 (`render(view,
gaussians, background)["render"]` per view under no_grad; SURVEY.md 3b, the unit behind "render ms/view") -- with the
whole forward of a view as ONE captured HIP graph:

    view select + clearing of the image buffer's counters   (riders of the next launch: hgs_runtime.strand_step.ViewTable)
    parameters -> Gaussians                                  (hgs_hair_params_forward / hgs_cloud_params_forward)
    preprocess, binning, sort, blend                         (hgs_forward_preprocess + hgs_forward_render, capacity mode)

A view switch re-points the graph's first node (no launch, no copies); the image is the one render() returns, bit for bit
(tests/test_gpu_train.py::test_frame_renderer_equals_render).  render() itself stays the drop-in: ~20 host-side tensor
operations per call keep it at about twice the kernels' time; this is the path for callers that render many views of a model
that does not change in between (a viewer, render.py's loop, the evaluation of a checkpoint)."""
import ctypes as C

import torch

import hgs_runtime as rt
from hgs_runtime.strand_step import ViewTable


class FrameRenderer:
    """renderer = FrameRenderer(gaussians, cameras, bg);  out = renderer.render(i)  ->  {"render": [3,H,W], "radii": [P]}

    The returned tensors are the graph's own output buffers: valid until the next render() of this renderer (clone to keep).
    The model may change VALUES between calls (training steps); after anything that re-creates its tensors or changes the
    SH degree (topology operators, oneupSHdegree, load_ply) the next render() notices and captures again.
    check=True (default) waits for the frame and validates it (binning capacity; an overflowing frame is rendered again with a
    larger capacity); check=False only enqueues -- call validate() before trusting the frames since the last check."""

    def __init__(self, gaussians, cameras, bg, use_graph=True, slack=1.5, frames_per_launch=1):
        from scene.hair_gaussian_model import HairGaussianModel
        self.g = gaussians
        self.hair = isinstance(gaussians, HairGaussianModel)
        self.views = cameras if isinstance(cameras, ViewTable) else ViewTable(cameras, targets=False)
        dev = self.views.device
        self.bg = rt.require_gpu_tensor(bg.to(dev), "bg", torch.float32).clone()   # (a constant of the captured graph)
        self.use_graph, self.slack = bool(use_graph), float(slack)
        # frames_per_launch = K > 1: render_batch() replays a second graph that holds K frames with K sets of output buffers
        # (a graph launch costs ~8 us of idle GPU whatever it holds: K frames per launch pay it once)
        self.K = max(1, int(frames_per_launch))
        self._many = None
        self.empty = torch.empty(0, device=dev)
        self._graph = self._binding = self._key = self._out = None
        self._pending = []          # graph frames enqueued since the last validation
        self._max_R = torch.zeros(1, dtype=torch.int32, device=dev)   # sticky maximum of num_rendered of THIS renderer's frames
        self._last_R = 0
        self.captures = 0

    # ---- one frame on the current stream, current slot view -------------------------------------------------------
    def _model_key(self):
        g = self.g
        ts = (g._endpoints, g._width, g._opacity, g._mask, g._features_dc, g._features_rest, g.endpoint_pairs) if self.hair \
            else (g._xyz, g._scaling, g._rotation, g._opacity, g._mask, g._features_dc, g._features_rest)
        return tuple((t.data_ptr(), tuple(t.shape)) for t in ts) + (int(g.active_sh_degree),)

    def _frame(self):
        """One frame of the current slot view on the current stream.  The Gaussians are what render() hands to the rasterizer,
        bit for bit: strand geometry and opacity from the kernel behind HairGaussianModel.derived_gaussians (which also
        carries the view select as a rider), a cloud's from the model's own getters (captured like any other launch)."""
        from diff_gaussian_rasterization import _C as raster
        g, vt, L = self.g, self.views, rt.lib()
        dev = vt.device
        f32 = dict(dtype=torch.float32, device=dev)
        with torch.no_grad(), torch.cuda.device(dev):
            if self.hair and g.endpoint_pairs.shape[0] > 0:
                fu = rt.StrandFusion()
                pairs = rt.require_gpu_tensor(g.endpoint_pairs, "endpoint_pairs", torch.int64)
                P = pairs.shape[0]
                xyz, scale, quat = torch.empty((P, 3), **f32), torch.empty((P, 3), **f32), torch.empty((P, 4), **f32)
                # the kernel's sigmoid, 1 / (1 + expf(-x)), gives torch.sigmoid's bits (tests/test_gpu_frames.py pins that on
                # 10^6 values): the opacity comes out of the same launch instead of one of its own
                opacity = torch.empty((P, 1), **f32)

                def fill(fused):     # (see hgs_runtime.strand_step._StrandIteration: same protocol)
                    if fused and not vt.counts_clean:
                        vt.flush_prologue()
                    vt.fill_prologue(fu, behind_counts=fused)
                hair = raster.HairSource(g._endpoints, pairs, g._width, g.dist_to_scale_factor, g._opacity, g._mask, fu, fill)
            else:
                vt.flush_prologue()
                xyz, scale, quat, opacity, hair = g.get_xyz, g.get_scaling, g.get_rotation, g.get_opacity, None
            # (get_features is cat(dc, rest): with no higher-order coefficients the DC tensor itself is that array)
            shs = g._features_dc if g._features_rest.shape[1] == 0 else g.get_features
            own_image = vt.take_image()
            out = raster.rasterize_gaussians_prezeroed(
                self.bg, xyz, self.empty, opacity, scale, quat, 1.0, self.empty, vt.viewmatrix, vt.projmatrix,
                vt.tanfovx, vt.tanfovy, vt.H, vt.W, shs, int(g.active_sh_degree), vt.campos, own_image, self._max_R, hair)
            if own_image is not None and xyz.shape[0] > 0:
                vt.counts_clean = raster._state["last_counts_clean"]
        self._last_R = int(out[0])
        return {"render": out[1], "radii": out[2]}

    # ---- capture / replay ---------------------------------------------------------------------------------------------
    def _read_max(self):
        worst = int(self._max_R.item()) & 0xFFFFFFFF       # (.item() waits for the frames on this stream)
        self._max_R.zero_()
        if worst == 0xFFFFFFFF:   # include/hgs.h HGS_WAIT_TIMED_OUT
            raise rt.HgsError("a raster pass gave up an inter-workgroup wait (status word 8): its frame is invalid")
        return worst

    def _capture(self):
        from diff_gaussian_rasterization import _C as raster
        vt, st = self.views, raster._state
        saved = {k: st[k] for k in ("async", "slack", "dirty", "cap_used")}
        st["async"], st["slack"] = True, self.slack         # capacity mode for the passes issued below (nothing blocks)
        try:
            s = self._stream = torch.cuda.Stream(device=vt.device)
            s.wait_stream(torch.cuda.current_stream(vt.device))
            with torch.cuda.stream(s):
                # a spread of views, eagerly: allocator warm-up, and a capacity that covers the busiest of them (a busier
                # one is caught by validate() and captured again)
                for v in sorted({(k * vt.n) // 16 for k in range(16)}):
                    vt.prologue(v, ride=True)
                    self._frame()
                    st["cap"] = max(st["cap"], raster.bucket_capacity(int(self._read_max() * self.slack) + 4096))
                ga = torch.cuda.CUDAGraph(keep_graph=True)
                with torch.cuda.graph(ga, stream=s):
                    vt.prologue(0, ride=True)
                    self._out = self._frame()
                ga.instantiate()
                self._binding = vt.graph_bind(ga)
                self._many = None
                if self.K > 1:
                    gk, outs = torch.cuda.CUDAGraph(keep_graph=True), []
                    with torch.cuda.graph(gk, pool=ga.pool(), stream=s):
                        for j in range(self.K):
                            vt.prologue(j % vt.n, lr=float(j), ride=True)      # lr = j: the tag graph_bind sorts by
                            outs.append(self._frame())
                    gk.instantiate()
                    self._many = (gk, vt.graph_bind(gk, self.K), outs)
            torch.cuda.current_stream(vt.device).wait_stream(s)
        finally:
            st.update(saved)
        self._graph, self._cap = ga, self._last_R
        self._max_R.zero_()                              # (the capture itself launched nothing)
        self._key = self._model_key()
        self.captures += 1

    def _enqueue(self, view):
        vt = self.views
        if not 0 <= int(view) < vt.n:
            raise rt.HgsError(f"view {view} outside the table (0..{vt.n - 1})")
        if not self.use_graph or (self.g.endpoint_pairs if self.hair else self.g._xyz).shape[0] == 0:
            from diff_gaussian_rasterization import _C as raster
            was, raster._state["async"] = raster._state["async"], False   # eager: the blocking mode, exact buffer sizes
            try:
                vt.prologue(int(view), ride=True)
                self._out = self._frame()
            finally:
                raster._state["async"] = was
            return
        if self._graph is None or self._key != self._model_key():
            self._capture()
        vt.graph_set(self._binding, int(view))
        vt.ensure_counts_clean()
        self._graph.replay()
        self._pending.append(int(view))

    def validate(self):
        """Wait for the frames enqueued since the last validation and compare the largest instance count among them with
        the capacity the graph was captured for.  Returns the views to render again ([] if every frame is good): which
        frame overflowed is not recorded, so all of them are suspect; the capacity has been raised and the next render()
        captures anew."""
        from diff_gaussian_rasterization import _C as raster
        pending, self._pending = self._pending, []
        if not pending:
            return []
        worst = self._read_max()
        if worst > self._cap:
            raster._state["cap"] = max(raster._state["cap"], raster.bucket_capacity(int(worst * self.slack) + 4096))
            self._graph = None
            return pending
        return []

    def render_batch(self, views, check=True):
        """len(views) == frames_per_launch frames by ONE graph launch; returns one {"render", "radii"} per view (the batch
        graph's own buffers: valid until the next render_batch())."""
        views = [int(v) for v in views]
        if not self.use_graph or self.K == 1 or len(views) != self.K or (self.g.endpoint_pairs if self.hair else self.g._xyz).shape[0] == 0:
            return [{k: t.clone() for k, t in self.render(v, check).items()} for v in views]
        if any(not 0 <= v < self.views.n for v in views):
            raise rt.HgsError(f"views {views}: outside the table (0..{self.views.n - 1})")
        if check and self._pending and self.validate():
            raise rt.HgsError("frames enqueued with check=False overflowed the binning capacity: call validate() first")
        for _attempt in range(4):
            if self._graph is None or self._key != self._model_key():
                self._capture()
            gk, binding, outs = self._many
            for j, v in enumerate(views):
                self.views.graph_set(binding, v, k=j)
            self.views.ensure_counts_clean()
            gk.replay()
            self._pending += views
            if not check or not self.validate():
                return outs
        raise RuntimeError("rasterizer capacity kept overflowing")

    def render(self, view, check=True):
        if check and self._pending and self.validate():
            raise rt.HgsError("frames enqueued with check=False overflowed the binning capacity: call validate() and render "
                              "the views it returns again before a checked render")
        for _attempt in range(4):
            self._enqueue(view)
            if not check or not self.validate():
                return self._out
        raise RuntimeError("rasterizer capacity kept overflowing")
