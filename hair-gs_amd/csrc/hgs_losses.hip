// hgs_losses.hip -- fused image loss: L1 + SSIM (11x11 Gaussian window, sigma 1.5, zero padding) forward and
// backward.  Replaces the five grouped 11x11 F.conv2d calls (+ their autograd) of the reference's
// loss/losses.py:43-84 `ssim` and :16-17 `l1_loss`, which on MI355X cost ~16 ms per 1080p iteration through
// MIOpen vs ~0.1 ms here.
//
// The window is separable (the reference builds it as an outer product, losses.py:34-41), so each 16x16 pixel
// block loads a 26x26 halo tile into LDS once, filters rows into LDS, then columns from LDS:
//   forward : mu1, mu2, E[x1^2], E[x2^2], E[x1 x2] -> ssim map -> per-block partial sums (ssim, |x1-x2|)
//             and the three maps a = dS/dmu1, b = dS/dE11, c = dS/dE12 (S at the window centre);
//   backward: dL/dx1(q) = g_ssim * [conv(a) + 2 x1(q) conv(b) + x2(q) conv(c)] + g_l1 * sign(x1 - x2)
// (the window is symmetric, so the adjoint of the filter is the filter itself; zero padding on both sides).
#include "hgs_common.h"
#include "hgs_head_tail.h"

namespace {

#define LT 32              // output pixels per block side (256 threads, 4 pixels per thread and pass)
#define HALO 5             // window radius
#define TILE (LT + 2 * HALO)
#define HP (LT + 1)

struct SsimWin { float w[11]; };

// development aid (tools/ssim_trace.py; build with -DHGS_SSIM_TRACE=1): per workgroup of the SSIM kernels, the time spent
// in each phase of its blocks -- 16 words per workgroup: [0] blocks, [1] filtered blocks, [2..9] phase sums (10 ns ticks),
// [10] first tick, [11] last tick
#ifndef HGS_SSIM_TRACE
#define HGS_SSIM_TRACE 0
#endif
#if HGS_SSIM_TRACE
__device__ unsigned long long* g_ssim_trace[2] = {nullptr, nullptr};
struct SsimTrace {
  unsigned long long* buf; unsigned long long last, acc[8], first; unsigned nb = 0, nf = 0;
  __device__ SsimTrace(int which) : buf(g_ssim_trace[which] ? g_ssim_trace[which] + 16 * (size_t)blockIdx.x : nullptr) {
    for (int i = 0; i < 8; i++) acc[i] = 0;
    first = last = __builtin_amdgcn_s_memrealtime();
  }
  __device__ void phase(int k) { const unsigned long long n = __builtin_amdgcn_s_memrealtime(); acc[k] += n - last; last = n; }
  __device__ void block(bool filtered) { nb++; nf += filtered ? 1u : 0u; }
  __device__ ~SsimTrace() {
    if (buf && threadIdx.x == 0) {
      buf[0] = nb; buf[1] = nf;
      for (int i = 0; i < 8; i++) buf[2 + i] = acc[i];
      buf[10] = first; buf[11] = __builtin_amdgcn_s_memrealtime();
    }
  }
};
#else
struct SsimTrace {
  __device__ SsimTrace(int) {}
  __device__ void phase(int) {}
  __device__ void block(bool) {}
};
#endif

__device__ __forceinline__ float block_sum(float v, float* red4) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red4[0] + red4[1]) + (red4[2] + red4[3]);
}
// the block's totals of two per-thread values (all threads return them); red: 8 floats that no thread touches again
// before its next barrier
__device__ __forceinline__ void block_sum2(float& a, float& b, float* red) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d, 64); b += __shfl_xor(b, d, 64); }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = a; red[4 + (threadIdx.x >> 6)] = b; }
  __syncthreads();
  a = (red[0] + red[1]) + (red[2] + red[3]);
  b = (red[4] + red[5]) + (red[6] + red[7]);
}

// Halo tile load.  Fast path (W % 4 == 0): the tile is widened to x in [bx0-8, bx0+LT+8) so that every row is 12
// aligned float4 chunks, each wholly inside or wholly outside the image -> 504 independent 16-B loads per plane, all in
// flight at once (the scalar form exposed ~7 dependent load latencies per block).  LDS column c <-> image x = bx0-8+c.
#define XOFF 8
#define TW (LT + 2 * XOFF)          // 48 columns staged
#define TPW (TW + 1)
template <int NP, typename PtrOf>
__device__ __forceinline__ void load_tiles(float (*t)[TILE][TPW], int H, int W, int bx0, int by0, PtrOf plane_ptr) {
  if ((W & 3) == 0) {
    for (int i = threadIdx.x; i < TILE * (TW / 4); i += 256) {
      const int r = i / (TW / 4), c4 = (i - r * (TW / 4)) * 4;
      const int y = by0 + r - HALO, x = bx0 - XOFF + c4;
      const bool in = (unsigned)y < (unsigned)H && x >= 0 && x < W;
#pragma unroll
      for (int p = 0; p < NP; p++) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in) v = hgs_load4(plane_ptr(p) + (size_t)y * W + x);
        t[p][r][c4] = v.x; t[p][r][c4 + 1] = v.y; t[p][r][c4 + 2] = v.z; t[p][r][c4 + 3] = v.w;
      }
    }
  } else {
    for (int i = threadIdx.x; i < TILE * TW; i += 256) {
      const int r = i / TW, c = i - r * TW;
      const int y = by0 + r - HALO, x = bx0 - XOFF + c;
      const bool in = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
#pragma unroll
      for (int p = 0; p < NP; p++) t[p][r][c] = in ? plane_ptr(p)[(size_t)y * W + x] : 0.f;
    }
  }
}

// Register-blocked separable filter.  Row pass: one work item = 4 consecutive outputs of one tile row (14 LDS reads
// per input image feed 4 x 11 taps); column pass: one thread = 4 consecutive output rows of one column (14 reads per
// filtered quantity).  ~29 LDS reads per output pixel instead of ~100 for the one-pixel-per-thread form.
// One row-pass work item: NO consecutive outputs of tile row r starting at column x0 (NO + 10 LDS reads per input image
// feed NO x 11 taps).
template <int NQ, int NO, typename F>
__device__ __forceinline__ void row_item(const SsimWin& win, float (*hz)[TILE][HP], F load, int r, int x0) {
  float acc[NQ][NO];
#pragma unroll
  for (int q = 0; q < NQ; q++)
#pragma unroll
    for (int o = 0; o < NO; o++) acc[q][o] = 0.f;
#pragma unroll
  for (int k = 0; k < NO + 10; k++) {
    float val[NQ];
    load(r, x0 + k, val);
#pragma unroll
    for (int o = 0; o < NO; o++) {
      const int tap = k - o;
      if (tap >= 0 && tap < 11) {
#pragma unroll
        for (int q = 0; q < NQ; q++) acc[q][o] += win.w[tap] * val[q];
      }
    }
  }
#pragma unroll
  for (int q = 0; q < NQ; q++)
#pragma unroll
    for (int o = 0; o < NO; o++) hz[q][r][x0 + o] = acc[q][o];
}
// The tile has TILE x LT / 4 = 336 four-output items for 256 threads.  Dealt item by item, 80 threads did two and the
// rest waited (the pass took 88 taps x NQ per thread for 57.75 on average); now every thread takes one four-output item and
// the remaining 80 are halved into 160 two-output items: 66 taps x NQ on the critical path, 26 LDS reads instead of 28.
template <int NQ, typename F>
__device__ __forceinline__ void row_pass(const SsimWin& win, float (*hz)[TILE][HP], F load) {
  constexpr int ITEMS = TILE * (LT / 4);
  static_assert(ITEMS > 256 && 2 * (ITEMS - 256) <= 256, "row-pass split assumes 256 < items <= 384");
  {
    const int it = threadIdx.x;
    const int r = it / (LT / 4), x0 = (it - r * (LT / 4)) * 4;
    row_item<NQ, 4>(win, hz, load, r, x0);
  }
  if (threadIdx.x < 2 * (ITEMS - 256)) {
    const int it = 256 + (threadIdx.x >> 1);
    const int r = it / (LT / 4), x0 = (it - r * (LT / 4)) * 4 + 2 * (threadIdx.x & 1);
    row_item<NQ, 2>(win, hz, load, r, x0);
  }
}
// The same row pass with its results kept in registers until the caller has passed a barrier (`row_store` then writes them):
// for kernels whose filtered planes ALIAS the input tile in LDS.
template <int NQ, int NO, typename F>
__device__ __forceinline__ void row_item_regs(const SsimWin& win, F load, int r, int x0, float (*acc)[NO]) {
#pragma unroll
  for (int q = 0; q < NQ; q++)
#pragma unroll
    for (int o = 0; o < NO; o++) acc[q][o] = 0.f;
#pragma unroll
  for (int k = 0; k < NO + 10; k++) {
    float val[NQ];
    load(r, x0 + k, val);
#pragma unroll
    for (int o = 0; o < NO; o++) {
      const int tap = k - o;
      if (tap >= 0 && tap < 11) {
#pragma unroll
        for (int q = 0; q < NQ; q++) acc[q][o] += win.w[tap] * val[q];
      }
    }
  }
}
template <int NQ> struct RowRegs { float a4[NQ][4], a2[NQ][2]; };
template <int NQ, typename F>
__device__ __forceinline__ void row_pass_regs(const SsimWin& win, F load, RowRegs<NQ>& rr) {
  constexpr int ITEMS = TILE * (LT / 4);
  {
    const int it = threadIdx.x;
    const int r = it / (LT / 4), x0 = (it - r * (LT / 4)) * 4;
    row_item_regs<NQ, 4>(win, load, r, x0, rr.a4);
  }
  if (threadIdx.x < 2 * (ITEMS - 256)) {
    const int it = 256 + (threadIdx.x >> 1);
    const int r = it / (LT / 4), x0 = (it - r * (LT / 4)) * 4 + 2 * (threadIdx.x & 1);
    row_item_regs<NQ, 2>(win, load, r, x0, rr.a2);
  }
}
template <int NQ>
__device__ __forceinline__ void row_store(float (*hz)[TILE][HP], const RowRegs<NQ>& rr) {
  constexpr int ITEMS = TILE * (LT / 4);
  {
    const int it = threadIdx.x;
    const int r = it / (LT / 4), x0 = (it - r * (LT / 4)) * 4;
#pragma unroll
    for (int q = 0; q < NQ; q++)
#pragma unroll
      for (int o = 0; o < 4; o++) hz[q][r][x0 + o] = rr.a4[q][o];
  }
  if (threadIdx.x < 2 * (ITEMS - 256)) {
    const int it = 256 + (threadIdx.x >> 1);
    const int r = it / (LT / 4), x0 = (it - r * (LT / 4)) * 4 + 2 * (threadIdx.x & 1);
#pragma unroll
    for (int q = 0; q < NQ; q++)
#pragma unroll
      for (int o = 0; o < 2; o++) hz[q][r][x0 + o] = rr.a2[q][o];
  }
}
template <int NQ>
__device__ __forceinline__ void col_pass(const SsimWin& win, float (*hz)[TILE][HP], int lx, int y0, float (*out)[4]) {
#pragma unroll
  for (int q = 0; q < NQ; q++)
#pragma unroll
    for (int o = 0; o < 4; o++) out[q][o] = 0.f;
#pragma unroll
  for (int k = 0; k < 14; k++) {
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const float v = hz[q][y0 + k][lx];
#pragma unroll
      for (int o = 0; o < 4; o++) {
        const int tap = k - o;
        if (tap >= 0 && tap < 11) out[q][o] += win.w[tap] * v;
      }
    }
  }
}

// Workgroup -> image block mapping, persistent workgroups.
//  * Workgroups are dealt round-robin to the 8 XCDs (each with its own L2) and neighbouring image blocks share 10-pixel
//    halos: with the natural mapping every halo is fetched from HBM by two or three different L2s (measured: 255 MB of
//    fabric traffic for 125 MB of algorithmic bytes).  XCD x owns the contiguous run of logical blocks
//    [x*chunk, (x+1)*chunk) -- a band of block rows of one channel -- so halos hit in L2 (measured after: 120 MB).
//  * Each workgroup walks its XCD's run with stride (workgroups per XCD) and fetches the NEXT block's halo tile into
//    registers while it filters the current one, so the HBM round trip is paid once per workgroup, not once per block
//    (3 blocks of 44 KB LDS fit a CU: too few to hide it by occupancy alone).
struct SsimGrid { int nbx, nby, C, chunk, total, sub; };
struct SsimBlock { int c, bx0, by0, logical; };
__device__ __forceinline__ void ssim_block_decode(const SsimGrid& gd, int logical, SsimBlock& o) {
  o.logical = logical;
  const int per = gd.nbx * gd.nby;
  o.c = logical / per;
  const int r = logical - o.c * per;
  const int by = r / gd.nbx;
  o.by0 = by * LT;
  o.bx0 = (r - by * gd.nbx) * LT;
}
__device__ __forceinline__ bool ssim_block(const SsimGrid& gd, int j, SsimBlock& o) {
  // the logical sequence is cut into SSIM_SUBBANDS contiguous sub-bands dealt round-robin to the XCDs: neighbours still
  // share an L2, and an XCD's share mixes parts of the frame (all-zero blocks are cheap: a single band per XCD left
  // the XCDs that own the middle of the frame as stragglers)
  const int q = j / gd.sub;
  o.logical = (q * 8 + (int)(blockIdx.x & 7)) * gd.sub + (j - q * gd.sub);
  if (j >= gd.chunk || o.logical >= gd.total) return false;
  const int per = gd.nbx * gd.nby;
  o.c = o.logical / per;
  const int r = o.logical - o.c * per;
  const int by = r / gd.nbx;
  o.by0 = by * LT;
  o.bx0 = (r - by * gd.nbx) * LT;
  return true;
}
#ifndef SSIM_SUBBANDS
#define SSIM_SUBBANDS 32
#endif
static inline SsimGrid ssim_grid(int C, int H, int W) {
  SsimGrid gd;
  gd.nbx = (W + LT - 1) / LT; gd.nby = (H + LT - 1) / LT; gd.C = C;
  gd.total = gd.nbx * gd.nby * C;
  gd.sub = (gd.total + SSIM_SUBBANDS - 1) / SSIM_SUBBANDS;
  gd.chunk = gd.sub * (SSIM_SUBBANDS / 8);
  return gd;
}
// persistent workgroups per XCD = 32 CUs x resident workgroups (forward: 38.7 KB LDS, register allocation held to 128
// VGPRs by amdgpu_waves_per_eu(4, 4) -> 4; at 131 it drops to 3 and the kernel takes 54 instead of 42 us; backward: 41 KB, 163 -> 3)
// Round 3: the forward's filtered planes alias its input tile in LDS (22.5 KB instead of 38.9) and, with the SLP vectorizer
// off, it needs 94 VGPRs: FIVE workgroups per CU instead of four (same box: 37.4 -> 35.5 us at north_star, 38.2 -> 35.4 at
// C3; aliasing alone, at four, changes nothing; six would need 80 VGPRs).  |x1 - x2| of a thread's own pixels is read before
// the tile is overwritten.
#ifndef HGS_SSIM_FWD_WAVES
#define HGS_SSIM_FWD_WAVES 5
#endif
#ifndef HGS_SSIM_FWD_ALIAS
#define HGS_SSIM_FWD_ALIAS 1
#endif
#define SSIM_FWD_WG_PER_XCD (32 * HGS_SSIM_FWD_WAVES)
#ifndef HGS_SSIM_BWD_WAVES
#define HGS_SSIM_BWD_WAVES 4
#endif
#define SSIM_BWD_WG_PER_XCD (32 * HGS_SSIM_BWD_WAVES)
static inline unsigned ssim_grid_size(const SsimGrid& gd, int per_xcd) { return 8u * (unsigned)(gd.chunk < per_xcd ? gd.chunk : per_xcd); }

// register staging of one halo tile (fast path, W % 4 == 0): 504 float4 per plane = 2 per thread
#define ST_F4 (TILE * (TW / 4))
// Every chunk is loaded UNCONDITIONALLY from an address clamped into the image, straight into its registers, and zeroed
// when it is written to LDS if it lay outside: a load under `if (inside)` merged with a zero made the compiler wait for
// the data where the load was ISSUED (to copy it into the merged registers), i.e. the prefetch was not one.
template <int NP> struct TileStage { float4 v[NP][2]; unsigned in; };   // in: bit u = chunk u lies inside the image
template <int NP, typename PtrOf>
__device__ __forceinline__ void stage_load(TileStage<NP>& st, int H, int W, int bx0, int by0, PtrOf plane_ptr) {
  st.in = 0u;
#pragma unroll
  for (int u = 0; u < 2; u++) {
    const int i = threadIdx.x + 256 * u;
    const int r = i / (TW / 4), c4 = (i - r * (TW / 4)) * 4;
    const int y = by0 + r - HALO, x = bx0 - XOFF + c4;
    const bool in = i < ST_F4 && (unsigned)y < (unsigned)H && x >= 0 && x < W;
    st.in |= in ? 1u << u : 0u;
    const size_t at = (size_t)min(max(y, 0), H - 1) * W + min(max(x, 0), W - 4);
#pragma unroll
    for (int p = 0; p < NP; p++) st.v[p][u] = hgs_load4(plane_ptr(p) + at);
  }
}
template <int NP>
__device__ __forceinline__ void stage_store(const TileStage<NP>& st, float (*t)[TILE][TPW]) {
#pragma unroll
  for (int u = 0; u < 2; u++) {
    const int i = threadIdx.x + 256 * u;
    if (i < ST_F4) {
      const int r = i / (TW / 4), c4 = (i - r * (TW / 4)) * 4;
      const bool in = (st.in >> u) & 1u;
#pragma unroll
      for (int p = 0; p < NP; p++) {
        const float4 v = st.v[p][u];
        t[p][r][c4] = in ? v.x : 0.f; t[p][r][c4 + 1] = in ? v.y : 0.f; t[p][r][c4 + 2] = in ? v.z : 0.f; t[p][r][c4 + 3] = in ? v.w : 0.f;
      }
    }
  }
}

// The three derivative maps of a pixel whose window is exactly zero in both images (mu = E = 0), evaluated with the very
// operations of the forward epilogue on run-time values (no constant folding of the hardware reciprocal: bit-identical).
__device__ __forceinline__ void ssim_zero_maps(float* m) {
  float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
  asm volatile("" : "+v"(C1), "+v"(C2));
  const float iB1 = __builtin_amdgcn_rcpf(C1), iB2 = __builtin_amdgcn_rcpf(C2);
  const float inv = iB1 * iB2;
  const float S = (C1 * C2) * inv;
  m[0] = 0.f; m[1] = -S * iB2; m[2] = 2.f * C1 * inv;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HGS_SSIM_FWD_WAVES, HGS_SSIM_FWD_WAVES))) void ssim_l1_fwd_kernel(int H, int W, SsimGrid gd, SsimWin win, const float* __restrict__ img1,
                                                          const float* __restrict__ img2_, const HgsViewTargets* __restrict__ tgt,
                                                          float* __restrict__ dmap, float* __restrict__ partials,
                                                          unsigned char* __restrict__ zero_flags) {
  // zero_flags (may be NULL): flag per block = "both images are exactly zero on the block's whole halo tile".  Hair
  // renders and their targets are black outside the hair: such a block's filtered maps are all zero, so the two filter
  // passes are skipped (the epilogue below then evaluates the same expressions on zeros, bit for bit what the full path
  // would produce), and the backward skips blocks whose 3x3 neighbourhood is flagged (its result is exactly zero).
#if HGS_SSIM_FWD_ALIAS
  __shared__ float smem[4 * TILE * HP];   // the tile (2 planes of TILE x TPW) and, behind a barrier, the 4 filtered planes
  static_assert(4 * TILE * HP >= 2 * TILE * TPW, "filtered planes cover the tile");
  float (*t)[TILE][TPW] = (float (*)[TILE][TPW])smem;
  float (*hz)[TILE][HP] = (float (*)[TILE][HP])smem;
#else
  __shared__ float t[2][TILE][TPW];
  __shared__ float hz[4][TILE][HP];   // mu1, mu2, E[x1^2 + x2^2], E[x1 x2]: S only needs the SUM of the two variances
#endif
  __shared__ float red[8];
  const size_t plane = (size_t)H * W, cp = (size_t)gd.C * plane;
  const HGS_GLOBAL float* im1 = hgs_global(img1);
  const HGS_GLOBAL float* im2 = tgt ? hgs_global(tgt->image) : hgs_global(img2_);   // per-view target read through the device-resident slot
  const bool fast = (W & 3) == 0;
  const int nwg = gridDim.x >> 3;
  int j = blockIdx.x >> 3;
  SsimBlock bk, nx;
  bool have = ssim_block(gd, j, bk);
  TileStage<2> st;
  if (fast && have) stage_load<2>(st, H, W, bk.bx0, bk.by0, [&](int p) { return (p == 0 ? im1 : im2) + bk.c * plane; });
  const int lx = threadIdx.x & (LT - 1), y0 = (threadIdx.x >> 5) * 4;
  SsimTrace tr(0);
  while (have) {
    tr.phase(7);
    int nz = 1;                                   // does this thread's share of the tile hold a non-zero value?
#if HGS_SSIM_FWD_ALIAS
    __syncthreads();                              // (the previous block's column pass / L1 reads are through)
#endif
    if (fast) {
      stage_store<2>(st, t);
      if (zero_flags) {
        nz = 0;
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
          for (int u = 0; u < 2; u++)
            if ((st.in >> u) & 1u) nz |= (st.v[p][u].x != 0.f) | (st.v[p][u].y != 0.f) | (st.v[p][u].z != 0.f) | (st.v[p][u].w != 0.f);
      }
    } else {
      load_tiles<2>(t, H, W, bk.bx0, bk.by0, [&](int p) { return (p == 0 ? im1 : im2) + bk.c * plane; });
    }
    const int any_nz = __syncthreads_or(nz);      // (also the barrier that publishes the tile)
    tr.phase(0);
    tr.block(any_nz != 0);
    j += nwg;
    const bool have_next = ssim_block(gd, j, nx);
    if (fast && have_next) stage_load<2>(st, H, W, nx.bx0, nx.by0, [&](int p) { return (p == 0 ? im1 : im2) + nx.c * plane; });
    float f[4][4];
#if HGS_SSIM_FWD_ALIAS
    float l1c[4];                                 // |x1 - x2| of this thread's pixels: read before the tile is overwritten
#pragma unroll
    for (int o = 0; o < 4; o++) l1c[o] = fabsf(t[0][y0 + o + HALO][lx + XOFF] - t[1][y0 + o + HALO][lx + XOFF]);
#endif
    if (any_nz) {
#if HGS_SSIM_FWD_ALIAS
      {
        RowRegs<4> rr;
        row_pass_regs<4>(win, [&](int r, int x, float* v) {
          const float a = t[0][r][x + XOFF - HALO], b = t[1][r][x + XOFF - HALO];
          v[0] = a; v[1] = b; v[2] = a * a + b * b; v[3] = a * b;
        }, rr);
        __syncthreads();
        row_store<4>(hz, rr);
      }
#else
      row_pass<4>(win, hz, [&](int r, int x, float* v) {   // x = tile column of the tap: image x = bx0 - HALO + x
        const float a = t[0][r][x + XOFF - HALO], b = t[1][r][x + XOFF - HALO];
        v[0] = a; v[1] = b; v[2] = a * a + b * b; v[3] = a * b;
      });
#endif
      __syncthreads();
      tr.phase(1);
      col_pass<4>(win, hz, lx, y0, f);
      tr.phase(2);
    } else {
#pragma unroll
      for (int q = 0; q < 4; q++)
#pragma unroll
        for (int o = 0; o < 4; o++) f[q][o] = 0.f;
    }
    if (zero_flags && threadIdx.x == 0) zero_flags[bk.logical] = any_nz ? 0 : 1;
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    const int px = bk.bx0 + lx;
    float ssim_v = 0.f, l1_v = 0.f;
#pragma unroll
    for (int o = 0; o < 4; o++) {
      const int py = bk.by0 + y0 + o;
      if (px < W && py < H) {
        const float mu1 = f[0][o], mu2 = f[1][o];
        const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
        const float s_sum = f[2][o] - (mu1_sq + mu2_sq), s12 = f[3][o] - mu12;    // sigma1^2 + sigma2^2, sigma12
        const float A1 = 2.f * mu12 + C1, A2 = 2.f * s12 + C2, B1 = mu1_sq + mu2_sq + C1, B2 = s_sum + C2;
        // B1 >= C1 > 0; B2 = C2 + (window variances) > 0 up to rounding: hardware reciprocals (1 ulp) instead of four
        // IEEE division sequences per pixel
        const float iB1 = __builtin_amdgcn_rcpf(B1), iB2 = __builtin_amdgcn_rcpf(B2);
        const float inv = iB1 * iB2;
        const float S = (A1 * A2) * inv;                                 // losses.py:71-73
        ssim_v += S;
#if HGS_SSIM_FWD_ALIAS
        l1_v += l1c[o];
#else
        l1_v += fabsf(t[0][y0 + o + HALO][lx + XOFF] - t[1][y0 + o + HALO][lx + XOFF]);
#endif
        if (any_nz) {   // (a flagged block's maps are the constants ssim_zero_maps(): the backward substitutes them)
          const size_t oo = bk.c * plane + (size_t)py * W + px;
          dmap[oo] = 2.f * mu2 * (A2 - A1) * inv - S * (2.f * mu1 * iB1 - 2.f * mu1 * iB2);   // dS/dmu1 at fixed E11, E12
          dmap[cp + oo] = -S * iB2;                                                           // dS/dE11
          dmap[2 * cp + oo] = 2.f * A1 * inv;                                                 // dS/dE12
        }
      }
    }
    tr.phase(3);
    block_sum2(ssim_v, l1_v, red);
    if (threadIdx.x == 0) {
      partials[2 * (size_t)bk.logical] = ssim_v;
      partials[2 * (size_t)bk.logical + 1] = l1_v;
    }
    // (no barrier here: t was last read before block_sum2's barriers, hz before them too, and red is rewritten only after
    // the two barriers of the next block's tile and row pass, which a thread still reading it has not reached)
    tr.phase(4);
    bk = nx;
    have = have_next;
  }
}

// ------------------------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HGS_SSIM_BWD_WAVES, HGS_SSIM_BWD_WAVES))) void ssim_l1_bwd_kernel(int H, int W, SsimGrid gd, SsimWin win, const float* __restrict__ img1,
                                                          const float* __restrict__ img2_, const HgsViewTargets* __restrict__ tgt,
                                                          const float* __restrict__ dmap,
                                                          const float* __restrict__ g_ssim_mean,
                                                          const float* __restrict__ g_l1_mean, const float* __restrict__ go,
                                                          float* __restrict__ dimg1, float* __restrict__ zero_buf,
                                                          int zero_n, const int* __restrict__ lists,
                                                          const unsigned char* __restrict__ zero_flags) {
  // (the loss head's endpoint-gradient buffer is cleared here, in passing: saves a launch before the smoothness scatter)
  for (int i = blockIdx.x * 256 + threadIdx.x; i < zero_n; i += gridDim.x * 256) zero_buf[i] = 0.f;
#ifndef HGS_SSIM_BWD_ALIAS
#define HGS_SSIM_BWD_ALIAS 1
#endif
#if HGS_SSIM_BWD_ALIAS
  // The filtered planes live in the SAME LDS as the input tile (24.7 KB instead of 24.7 + 16.6): a fourth workgroup fits a
  // CU.  The row pass keeps its results in registers until every thread has read the tile (one barrier), and the next tile
  // is written only when every thread is through with the column pass (one more).
  __shared__ float smem[3 * TILE * TPW];
  float (*t)[TILE][TPW] = (float (*)[TILE][TPW])smem;
  float (*hz)[TILE][HP] = (float (*)[TILE][HP])smem;
#else
  __shared__ float t[3][TILE][TPW];
  __shared__ float hz[3][TILE][HP];
#endif
  const HGS_GLOBAL float* im1 = hgs_global(img1);
  const HGS_GLOBAL float* im2 = tgt ? hgs_global(tgt->image) : hgs_global(img2_);
  const HGS_GLOBAL float* dmg = hgs_global(dmap);
  const size_t plane = (size_t)H * W, cp = (size_t)gd.C * plane;
  const float n = 1.f / (float)((size_t)gd.C * plane);
  const float up = go ? *go : 1.f;                 // upstream dL/dtotal of the loss head (NULL: 1)
  const float gs = *g_ssim_mean * up * n, gl = *g_l1_mean * up * n;
  const bool fast = (W & 3) == 0;
  const int nwg = gridDim.x >> 3;
  const int lx = threadIdx.x & (LT - 1), y0 = (threadIdx.x >> 5) * 4;
  int j = blockIdx.x >> 3;
  SsimBlock bk, nx;
  // `lists` (the loss head's forward built it, build_block_lists): the blocks whose gradient is not identically zero, in
  // logical order, and the rest.  Each XCD takes an equal contiguous share of the first list (the halo sharing in its L2
  // is kept, and the work is balanced however the hair sits in the frame); the second list is only zero-filled.
  const int xcd = blockIdx.x & 7;
  int lo = 0, hi = 0;
  const int* work = nullptr;
  if (lists) {
    const int n_work = lists[0], n_skip = lists[1];
    work = lists + 4;
    const int* skipped = work + gd.total;
    lo = (int)(((long long)n_work * xcd) >> 3);
    hi = (int)(((long long)n_work * (xcd + 1)) >> 3);
    for (int q = blockIdx.x; q < n_skip; q += gridDim.x) {
      SsimBlock z;
      ssim_block_decode(gd, skipped[q], z);
      const int px = z.bx0 + lx;
#pragma unroll
      for (int o = 0; o < 4; o++) {
        const int py = z.by0 + y0 + o;
        if (px < W && py < H) dimg1[z.c * plane + (size_t)py * W + px] = 0.f;
      }
    }
  }
  // The ids of this workgroup's next 64 blocks sit one per lane (one vector load per 64 blocks): an id fetched per block is
  // wave-uniform, so the compiler moves it to a scalar register at once -- a full memory round trip, and a wait for every
  // load in flight, in the middle of each block.
  const int jstep = nwg, j0 = j;
  int idv = 0, idbase = 0;                        // idv: id of block number idbase + lane of this workgroup
  auto refill = [&](int k0) {
    idbase = k0;
    const int jj = j0 + (k0 + (int)(threadIdx.x & 63)) * jstep;
    idv = (work && lo + jj < hi) ? work[lo + jj] : 0;
  };
  int kblk = 0;                                   // number of the current block of this workgroup
  auto block_at = [&](int jj, int k, SsimBlock& o) {
    if (!work) return ssim_block(gd, jj, o);
    if (lo + jj >= hi) return false;
    if (k - idbase >= 64) refill(k);
    ssim_block_decode(gd, __builtin_amdgcn_readlane(idv, k - idbase), o);
    return true;
  };
  refill(0);
  bool have = block_at(j, 0, bk);
  TileStage<3> st;
  float x1[4], x2[4];                              // the block's own pixels of both images
  auto centre = [&](const SsimBlock& q) {
#pragma unroll
    for (int o = 0; o < 4; o++) {
      const int px = q.bx0 + lx, py = q.by0 + y0 + o;
      x1[o] = 0.f; x2[o] = 0.f;
      if (px < W && py < H) { const size_t oo = q.c * plane + (size_t)py * W + px; x1[o] = im1[oo]; x2[o] = im2[oo]; }
    }
  };
  // the forward wrote no maps for the blocks it flagged all-zero: their constants are put in place of the loads
  float zm[3];
  ssim_zero_maps(zm);
  // (the flag bytes travel with the tile and are applied when it is written to LDS: a load that waits for its flag put one
  // memory round trip per block on the critical path; the maps of a flagged block are loaded and discarded)
  unsigned char fl[2] = {0, 0};
  auto stage = [&](const SsimBlock& q) {
    stage_load<3>(st, H, W, q.bx0, q.by0, [&](int p) { return dmg + p * cp + q.c * plane; });
    if (!zero_flags) return;
    const int per = gd.nbx * gd.nby;
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int i = threadIdx.x + 256 * u;
      const int r = i / (TW / 4), c4 = (i - r * (TW / 4)) * 4;
      const int y = q.by0 + r - HALO, x = q.bx0 - XOFF + c4;
      fl[u] = zero_flags[q.c * per + (min(max(y, 0), H - 1) / LT) * gd.nbx + min(max(x, 0), W - 4) / LT];   // (unconditional, like the tile)
    }
  };
  auto apply_flags = [&]() {
#pragma unroll
    for (int u = 0; u < 2; u++)
      if (fl[u] && ((st.in >> u) & 1u)) {
#pragma unroll
        for (int p = 0; p < 3; p++) st.v[p][u] = make_float4(zm[p], zm[p], zm[p], zm[p]);
      }
  };
  if (have && fast) stage(bk);
  SsimTrace tr(1);
  while (have) {
    tr.phase(7);
    tr.block(true);
#if HGS_SSIM_BWD_ALIAS
    __syncthreads();                               // (the previous block's column pass has read the aliased planes)
#endif
    if (fast) { apply_flags(); stage_store<3>(st, t); }
    else load_tiles<3>(t, H, W, bk.bx0, bk.by0, [&](int p) { return dmg + p * cp + bk.c * plane; });
    __syncthreads();
    tr.phase(0);
    // this block's own pixels: needed in the epilogue, two filter passes from here (fetched a block ahead they were
    // loop-carried registers, and the compiler parked a wait for every load in flight in front of the row pass)
    centre(bk);
    j += nwg;
    kblk++;
    const bool have_next = block_at(j, kblk, nx);
    if (have_next && fast) stage(nx);
#if HGS_SSIM_BWD_ALIAS
    {
      RowRegs<3> rr;
      row_pass_regs<3>(win, [&](int r, int x, float* v) {
        v[0] = t[0][r][x + XOFF - HALO]; v[1] = t[1][r][x + XOFF - HALO]; v[2] = t[2][r][x + XOFF - HALO];
      }, rr);
      __syncthreads();                             // every thread has read the tile: its LDS becomes the filtered planes
      row_store<3>(hz, rr);
    }
#else
    row_pass<3>(win, hz, [&](int r, int x, float* v) {
      v[0] = t[0][r][x + XOFF - HALO]; v[1] = t[1][r][x + XOFF - HALO]; v[2] = t[2][r][x + XOFF - HALO];
    });
#endif
    __syncthreads();
    tr.phase(1);
    float f[3][4];
    col_pass<3>(win, hz, lx, y0, f);
    tr.phase(2);
    const int px = bk.bx0 + lx;
#pragma unroll
    for (int o = 0; o < 4; o++) {
      const int py = bk.by0 + y0 + o;
      if (px < W && py < H) {
        const size_t oo = bk.c * plane + (size_t)py * W + px;
        const float d = x1[o] - x2[o];
        const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        dimg1[oo] = gs * (f[0][o] + 2.f * x1[o] * f[1][o] + x2[o] * f[2][o]) + gl * sgn;
      }
    }
    // (no barrier here: t is read only before the row-pass barrier, and hz is rewritten only after the next block's tile
    // barrier, which a thread still in this block's column pass has not reached)
    tr.phase(4);
    bk = nx;
    have = have_next;
  }
}

// ---- orientation loss (reference loss/losses.py:224-289) ---------------------------------------------------------
// per pixel: world-space direction image -> view space (x,y) -> unit 2-vector -> angle in [0,pi) w.r.t. the image
// y axis -> bidirectional difference to the GT angle, confidence-weighted, averaged over the mask.
struct OriParams { const float* view; float bg0, bg1, bg2; float min_val; int has_mask; };

__device__ __forceinline__ bool ori_pixel(const OriParams& p, float o0, float o1, float o2, float& px, float& py, float& r,
                                          float& n, float& x, float& y, float& yq, float& theta) {
  const float* v = p.view;                      // world_view_transform, row-major 4x4 (device, wave-uniform)
  px = o0 * v[0] + o1 * v[4] + o2 * v[8];      // (flat @ world_view[:3,:3])[:, :2]
  py = o0 * v[1] + o1 * v[5] + o2 * v[9];
  r = sqrtf(px * px + py * py);
  n = r + p.min_val;
  const float in = __builtin_amdgcn_rcpf(n);    // (hardware reciprocals, 1 ulp, here and in the gradient: the per-pixel kernel is
  x = px * in;                                  //  bound by its vector instructions -- 357 per wavefront, 63 % of the pipe -- and an
  y = py * in;                                  //  IEEE division is ten of them)
  yq = y < p.min_val ? y + p.min_val : y;
  theta = atan2f(x, yq);
  if (theta < 0.f) theta += 3.14159265358979323846f;
  return true;
}

__global__ __launch_bounds__(256) void ori_fwd_kernel(int N, OriParams p, const float* __restrict__ omap,
                                                      const float* __restrict__ gt, const float* __restrict__ conf,
                                                      const uint8_t* __restrict__ mask, float* __restrict__ partials) {
  __shared__ float red[4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  float s = 0.f, cnt = 0.f;
  if (i < N) {
    const float o0 = omap[i], o1 = omap[(size_t)N + i], o2 = omap[2 * (size_t)N + i];
    const bool m = p.has_mask ? mask[i] != 0 : (o0 != p.bg0 || o1 != p.bg1 || o2 != p.bg2);
    if (m) {
      float px, py, r, n, x, y, yq, th;
      ori_pixel(p, o0, o1, o2, px, py, r, n, x, y, yq, th);
      const float hp = 1.57079632679489661923f;
      const float diff = hp - fabsf(fabsf(th - gt[i]) - hp);
      s = diff * conf[i];
      cnt = 1.f;
    }
  }
  const float bs = block_sum(s, red);
  const float bc = block_sum(cnt, red);
  if (threadIdx.x == 0) { partials[2 * blockIdx.x] = bs; partials[2 * blockIdx.x + 1] = bc; }
}

__global__ __launch_bounds__(256) void ori_bwd_kernel(int N, OriParams p, const float* __restrict__ omap,
                                                      const float* __restrict__ gt, const float* __restrict__ conf,
                                                      const uint8_t* __restrict__ mask, const float* __restrict__ g_loss,
                                                      const float* __restrict__ mask_count, float* __restrict__ d_omap) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const float o0 = omap[i], o1 = omap[(size_t)N + i], o2 = omap[2 * (size_t)N + i];
  const bool m = p.has_mask ? mask[i] != 0 : (o0 != p.bg0 || o1 != p.bg1 || o2 != p.bg2);
  float g0 = 0.f, g1 = 0.f, g2 = 0.f;
  if (m) {
    float px, py, r, n, x, y, yq, th;
    ori_pixel(p, o0, o1, o2, px, py, r, n, x, y, yq, th);
    {
      const float hp = 1.57079632679489661923f;
      const float e = th - gt[i];
      const float u = fabsf(e) - hp;
      const float sg = (u > 0.f ? 1.f : (u < 0.f ? -1.f : 0.f)) * (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f));
      const float dth = -sg * conf[i] * (*g_loss) / (*mask_count);       // dL/dtheta
      const float den = x * x + yq * yq;
      const float dx = dth * (yq / den), dy = dth * (-x / den);           // atan2(x, yq)
      // x = px/n, y = py/n, n = r + eps
      // r = 0 (a masked pixel nothing was blended into): torch's norm has the subgradient 0 there, the direct 1 / n path stays --
      // the reference's gradient at such a pixel is ~conf / (count eps^2), and so is this one (tests/test_ref_loss_pins.py)
      const float inv_n = 1.f / n, inv_n2 = inv_n * inv_n, ir = r > 0.f ? 1.f / r : 0.f;
      const float dn = -(dx * px + dy * py) * inv_n2;
      const float dpx = dx * inv_n + dn * px * ir, dpy = dy * inv_n + dn * py * ir;
      const float* v = p.view;
      g0 = dpx * v[0] + dpy * v[1]; g1 = dpx * v[4] + dpy * v[5]; g2 = dpx * v[8] + dpy * v[9];
    }
  }
  d_omap[i] = g0; d_omap[(size_t)N + i] = g1; d_omap[2 * (size_t)N + i] = g2;
}


// ---- loss head (hgs_loss_head_*): the per-pixel terms that are not SSIM/L1, and the final reduction ---------------
// pix kernels: binary cross-entropy with logits of the blended mask channel against the view's float mask
// (loss/losses.py:240-248, F.binary_cross_entropy_with_logits, mean over H*W) and the orientation term above, in
// one pass over the pixels; targets come from the device-resident HgsViewTargets.
struct HeadFlags { int bce, ori; };

// gradient of the orientation term w.r.t. the direction image at one masked pixel, `scale` = dL/d(term) / mask count
__device__ __forceinline__ void ori_pixel_grad(const OriParams& p, float px, float py, float r, float n, float x, float yq,
                                               float th, float gt, float conf, float scale, float& g0, float& g1, float& g2) {
  const float hp = 1.57079632679489661923f;
  const float e = th - gt;
  const float u = fabsf(e) - hp;
  const float sg = (u > 0.f ? 1.f : (u < 0.f ? -1.f : 0.f)) * (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f));
  const float dth = -sg * conf * scale;                               // dL/dtheta
  const float iden = __builtin_amdgcn_rcpf(x * x + yq * yq);
  const float dx = dth * (yq * iden), dy = dth * (-x * iden);         // atan2(x, yq)
  // x = px/n, y = py/n, n = r + eps.  r = 0 (a masked pixel nothing was blended into): torch's norm has the subgradient 0
  // there and the direct 1 / n path stays -- the reference's gradient at such a pixel is ~conf / (count eps^2), and so is this one
  const float inv_n = __builtin_amdgcn_rcpf(n), inv_n2 = inv_n * inv_n, ir = r > 0.f ? __builtin_amdgcn_rcpf(r) : 0.f;
  const float dn = -(dx * px + dy * py) * inv_n2;
  const float dpx = dx * inv_n + dn * px * ir, dpy = dy * inv_n + dn * py * ir;
  const float* v = p.view;
  g0 = dpx * v[0] + dpy * v[1]; g1 = dpx * v[4] + dpy * v[5]; g2 = dpx * v[8] + dpy * v[9];
}

// d_unit != NULL: the gradient planes for an upstream gradient of 1 are written in the same pass (g_mask = l_mask/HW,
// g_ori = l_orientation; the orientation term is normalised by tgt->mask_count, known before the pass)
struct HeadReduce { int nb_ssim, nb_pix, nb_smooth; float inv_chw, inv_hw; float l_dssim, l_mask, l_ori, l_smooth; int bce, ori; };

// The head's reduction.  Part 1 -- everything that does not depend on the per-pixel kernel's own partials -- runs in
// workgroup 0 of pix_fwd_kernel (the SSIM forward and the smoothness forward are earlier launches); the tail (the sums
// over the per-pixel partials, hgs_head_tail.h) needs a later launch: head_tail_kernel, or a spare workgroup of the
// parameter backward (HgsHeadParams.defer_tail).  Round 2's single finalize launch cost 8.4 us of the iteration.
__device__ __forceinline__ void head_part1(const HeadReduce& h, const float* __restrict__ p_ssim,
                                           const float* __restrict__ p_smooth, float* __restrict__ out) {
  __shared__ float red4[4][4];
  float sv[4] = {0.f, 0.f, 0.f, 0.f};              // SSIM sum, L1 sum, smoothness sum, smoothness count
  // Twelve 8-byte loads per thread in flight per round trip, issued from clamped indices before the first use (a 1080p
  // frame: 24 per thread, two trips; the smoothness partials: one).  This workgroup runs while 8100 pixel workgroups saturate
  // HBM -- every dependent round trip costs it several us, and the launch lasts as long as its slowest workgroup.  The order
  // of the additions per thread is unchanged (bit-identical sums).
  const float2* ps = (const float2*)p_ssim;
  const float2* pm = (const float2*)p_smooth;
  float2 w[4];
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int i = u * 256 + (int)threadIdx.x;
    w[u] = h.nb_smooth > 0 ? pm[min(i, h.nb_smooth - 1)] : make_float2(0.f, 0.f);
  }
  for (int base = 0; base < h.nb_ssim; base += 12 * 256) {
    float2 v[12];
#pragma unroll
    for (int u = 0; u < 12; u++) v[u] = ps[min(base + u * 256 + (int)threadIdx.x, h.nb_ssim - 1)];
#pragma unroll
    for (int u = 0; u < 12; u++)
      if (base + u * 256 + (int)threadIdx.x < h.nb_ssim) { sv[0] += v[u].x; sv[1] += v[u].y; }
  }
#pragma unroll
  for (int u = 0; u < 4; u++)
    if (u * 256 + (int)threadIdx.x < h.nb_smooth) { sv[2] += w[u].x; sv[3] += w[u].y; }
  for (int i = 4 * 256 + (int)threadIdx.x; i < h.nb_smooth; i += 256) { sv[2] += pm[i].x; sv[3] += pm[i].y; }
#pragma unroll
  for (int q = 0; q < 4; q++)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sv[q] += __shfl_xor(sv[q], d, 64);
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int q = 0; q < 4; q++) red4[q][threadIdx.x >> 6] = sv[q];
  __syncthreads();
  if (threadIdx.x != 0) return;
#pragma unroll
  for (int q = 0; q < 4; q++) sv[q] = (red4[q][0] + red4[q][1]) + (red4[q][2] + red4[q][3]);
  const float ssim_s = sv[0], l1_s = sv[1], sm_s = sv[2], sm_c = sv[3];
  const float l1 = l1_s * h.inv_chw, dssim = 1.f - ssim_s * h.inv_chw;
  const float w_l1 = fmaxf(0.f, 1.f - h.l_dssim);
  out[HGS_HEAD_TOTAL] = out[HGS_HEAD_TOTAL_FWD] = fmaf(h.l_dssim, dssim, w_l1 * l1);      // (the tail adds the other terms)
  out[HGS_HEAD_L1] = l1; out[HGS_HEAD_DSSIM] = dssim;
  out[HGS_HEAD_SMOOTH] = h.nb_smooth > 0 ? sm_s / fmaxf(sm_c, 1.f) : 0.f;
  out[HGS_HEAD_SMOOTH_COUNT] = sm_c;
  out[HGS_HEAD_G_SSIM] = -h.l_dssim; out[HGS_HEAD_G_L1] = w_l1;
  out[HGS_HEAD_G_MASK] = h.bce ? h.l_mask * h.inv_hw : 0.f;
  out[HGS_HEAD_G_ORI] = h.ori ? h.l_ori : 0.f;
  out[HGS_HEAD_G_SMOOTH] = h.nb_smooth > 0 ? h.l_smooth : 0.f;
}

#ifndef HGS_PIX_TRACE
#define HGS_PIX_TRACE 0   // development aid: timestamps of the side workgroups behind the block lists (tools/dev/pix_trace.py)
#endif
#define HEAD_MAX_TILES 32768     // tiles of a frame the list builder keeps the use bits of (4K: 32400); more: the hint is ignored
#define HEAD_MAX_FLAGGED 32768   // SSIM blocks of a frame the backward's block lists are built for (4K RGB: 24480); without the lists (and the zero-block flags that come with them) the SSIM pair takes 90 instead of 77 us at north_star
static_assert(HEAD_MAX_FLAGGED % (32 * 256) == 0, "whole 32-block words per thread of the list builder");
// Block lists of the SSIM backward, built by ONE workgroup of pix_fwd_kernel (the SSIM forward before that launch flagged
// the blocks whose halo tile is exactly zero in both images): lists = [n_work, n_skip, -, -][ids of the blocks with a
// non-zero gradient, logical order][the other ids].  A block has work unless its whole 3x3 neighbourhood is flagged: then
// a = dS/dmu1 == 0 on its halo tile and x1 == x2 == 0 on its own pixels, its gradient is exactly zero and the backward
// reads nothing for it.  (Round 2 built the lists in the single-workgroup finalize kernel, on the iteration's critical
// path: 2.7 us of its 8.4.)
//   tile_used (may be NULL; HgsHeadParams.tile_used): the rasterizer's per-tile contributor count.  The blend backward
//   reads dL/dimage only on tiles where some pixel blended an entry, so a block none of whose 2x2 tiles did is left out
//   of BOTH lists -- neither filtered nor zero-filled: nobody reads its gradient.  (The 3x3 rule above has to spare a
//   block next to the hair, whose gradient is not zero; this rule looks at who reads it: 23 % -> 1/3 of the blocks of a
//   hair frame need no filter pass.)
// even bits of a 64-bit value, compressed into its low 32 bits
__device__ __forceinline__ unsigned even_bits(unsigned long long x) {
  x &= 0x5555555555555555ull;
  x = (x | (x >> 1)) & 0x3333333333333333ull;
  x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
  x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
  x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
  x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
  return (unsigned)x;
}

__device__ __forceinline__ void build_block_lists(const SsimGrid& gd, const unsigned char* __restrict__ zero_flags,
                                                  int* __restrict__ lists, const unsigned int* __restrict__ tile_used,
                                                  int tiles_x, int tiles_y) {
#if HGS_PIX_TRACE
#define BL_STAMP(k) do { __syncthreads(); if (threadIdx.x == 0) lists[4 + 2 * gd.total + 7 + (k)] = (int)(unsigned)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define BL_STAMP(k) do { } while (0)
#endif
  // This workgroup is what the launch waits for (timestamps, tools/dev/pix_trace.py: the 8100 pixel workgroups are through
  // after 17 us; a builder that gave every thread a run of 24 consecutive blocks -- 13 dependent LDS reads and two LDS
  // read-modify-writes per block, then one store at a time -- ended at 21-25 us).  Now everything is done on 32-block WORDS:
  // bitmaps in LDS from loads that are all in flight together; one thread per word forms the 3x3 test from nine funnel
  // shifts of the zero bitmap (edge masks: a neighbour outside the frame is ignored) and the 2x2-tile test from the tile
  // bitmap (pairs OR-ed, even bits compressed); counts, one scan, and every thread emits the ids of its own words.
  constexpr int WPT = HEAD_MAX_FLAGGED / 32 / 256;     // words per thread of the largest frame
  __shared__ unsigned zbits[HEAD_MAX_FLAGGED / 32];    // zero flags of the blocks (bits behind the last block: 1)
  __shared__ unsigned tbits[HEAD_MAX_TILES / 32 + 2];  // tile_used as a bitmap
  __shared__ int wsum[4], zsum[4];
  const int total = gd.total, nwords = (total + 31) >> 5;
  const unsigned* zf = (const unsigned*)zero_flags;   // (4-byte aligned: the flags start on a float of the scratch)
  const int n_tiles = tiles_x * tiles_y;
  if (tile_used && (n_tiles > HEAD_MAX_TILES || ((size_t)tile_used & 15))) tile_used = nullptr;
  const uint4* tu4 = (const uint4*)tile_used;
  const int n4 = (n_tiles + 3) >> 2;
  uint4 tv[8];
  if (tile_used) {
    for (int i = threadIdx.x; i < HEAD_MAX_TILES / 32 + 2; i += 256) tbits[i] = 0u;
#pragma unroll
    for (int u = 0; u < 8; u++) tv[u] = tu4[min(u * 256 + (int)threadIdx.x, n4 - 1)];   // (the words behind the last tile belong to the image buffer)
  }
  // bit b of zbits = zero flag of block b: a thread packs runs of 32 flag bytes (eight independent word loads; the bytes
  // behind the last flag belong to the same scratch buffer and are overwritten with ones: neutral in the tests below)
#pragma unroll 1
  for (int w = threadIdx.x; w < nwords; w += 256) {
    unsigned v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = zf[8 * w + u];
    unsigned bits = 0;                                 // flags are bytes 0 / 1: bit 0 of each byte of a word -> 4 adjacent bits
#pragma unroll
    for (int u = 0; u < 8; u++) bits |= (((v[u] & 0x01010101u) * 0x00204081u) >> 21 & 0xFu) << (4 * u);
    const int left = total - 32 * w;
    if (left < 32) bits |= ~((1u << left) - 1u);
    zbits[w] = bits;
  }
  if (tile_used) {
    __syncthreads();                                   // (tbits cleared)
    for (int base = 0; base < n4; base += 8 * 256) {
      if (base) {
#pragma unroll
        for (int u = 0; u < 8; u++) tv[u] = tu4[min(base + u * 256 + (int)threadIdx.x, n4 - 1)];
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int q = base + u * 256 + (int)threadIdx.x, t = 4 * q;      // tiles t .. t + 3: one nibble of a word
        if (q < n4) {
          unsigned nib = (tv[u].x != 0u ? 1u : 0u) | (tv[u].y != 0u ? 2u : 0u) | (tv[u].z != 0u ? 4u : 0u) | (tv[u].w != 0u ? 8u : 0u);
          if (t + 4 > n_tiles) nib &= (1u << (n_tiles - t)) - 1u;
          if (nib) atomicOr(&tbits[t >> 5], nib << (t & 31));
        }
      }
    }
  }
  __syncthreads();
  BL_STAMP(0);
  const int per = gd.nbx * gd.nby, nbx = gd.nbx;
  // 32 bits of the zero bitmap from bit `start` on; bits outside [0, total) read as 1
  auto zfetch = [&](int start) -> unsigned {
    const int wi = start >> 5, sh = start & 31;        // (arithmetic shift: floor)
    const unsigned lo = (wi >= 0 && wi < nwords) ? zbits[wi] : 0xFFFFFFFFu;
    const unsigned hi = (wi + 1 >= 0 && wi + 1 < nwords) ? zbits[wi + 1] : 0xFFFFFFFFu;
    return sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
  };
  // 64 bits of the tile bitmap from bit `start` (>= 0) on (the bitmap is zero behind the last tile)
  auto tfetch = [&](int start) -> unsigned long long {
    const int wi = start >> 5, sh = start & 31;
    const unsigned long long a = tbits[wi], b = tbits[wi + 1], c = tbits[wi + 2];
    const unsigned long long lo = a | (b << 32);
    return sh ? (lo >> sh) | (c << (64 - sh)) : lo;
  };
  unsigned work_w[WPT], fill_w[WPT];
  int cnt = 0, cnt_z = 0;
  const int wpt = (nwords + 255) >> 8, w_first = threadIdx.x * wpt;
#pragma unroll
  for (int q = 0; q < WPT; q++) {
    work_w[q] = 0u; fill_w[q] = 0u;
    const int w = w_first + q;
    if (q >= wpt || w >= nwords) continue;
    const int b0 = 32 * w;
    // edge masks of the word's blocks, and its row segments for the tile test
    unsigned L = 0u, R = 0u, T = 0u, Bm = 0u, used = tile_used ? 0u : 0xFFFFFFFFu;
    int r = b0 % per, by = r / nbx, bx = r - by * nbx;
    for (int i = 0; i < 32;) {
      const int len = min(32 - i, nbx - bx);           // blocks i .. i + len - 1 of the word lie in block row `by`
      const unsigned seg = (len == 32 ? 0xFFFFFFFFu : ((1u << len) - 1u)) << i;
      if (bx == 0) L |= 1u << i;
      if (bx + len == nbx) R |= 1u << (i + len - 1);
      if (by == 0) T |= seg;
      if (by == gd.nby - 1) Bm |= seg;
      if (tile_used) {
#pragma unroll
        for (int ty = 0; ty < 2; ty++) {               // the blocks' 2 x 2 tiles of 16 x 16 pixels (LT = 2 * HGS_TILE)
          const int Y = 2 * by + ty;
          if (Y < tiles_y) {
            unsigned long long v = tfetch(Y * tiles_x + 2 * bx);
            const int nt = min(2 * len, tiles_x - 2 * bx);               // tiles of this row that belong to the segment
            if (nt < 64) v &= (1ull << nt) - 1ull;
            used |= (even_bits(v | (v >> 1)) & (len == 32 ? 0xFFFFFFFFu : ((1u << len) - 1u))) << i;
          }
        }
      }
      i += len; bx = 0;
      if (++by == gd.nby) by = 0;
    }
    auto H = [&](int start) { return zfetch(start) & (zfetch(start - 1) | L) & (zfetch(start + 1) | R); };
    const unsigned zero = H(b0) & (H(b0 - nbx) | T) & (H(b0 + nbx) | Bm);
    const int left = total - b0;
    const unsigned valid = left >= 32 ? 0xFFFFFFFFu : ((1u << left) - 1u);
    work_w[q] = used & ~zero & valid;
    fill_w[q] = used & zero & valid;                   // read by the blend backward, gradient exactly zero: zero-filled
    cnt += __popc(work_w[q]); cnt_z += __popc(fill_w[q]);
  }
  BL_STAMP(1);
  int inc = cnt, inc_z = cnt_z;                        // inclusive scans over the wave, then over the 4 wave totals
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int v = __shfl_up(inc, d, 64), vz = __shfl_up(inc_z, d, 64);
    if ((int)(threadIdx.x & 63) >= d) { inc += v; inc_z += vz; }
  }
  if ((threadIdx.x & 63) == 63) { wsum[threadIdx.x >> 6] = inc; zsum[threadIdx.x >> 6] = inc_z; }
  __syncthreads();
  int base = 0, all = 0, base_z = 0, all_z = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int v = wsum[k], vz = zsum[k];
    if (k < (int)(threadIdx.x >> 6)) { base += v; base_z += vz; }
    all += v; all_z += vz;
  }
  int w = base + inc - cnt, z = base_z + inc_z - cnt_z;   // work / zero-fill blocks before this thread's first word
  int* work = lists + 4;
  int* skipped = work + total;
  BL_STAMP(2);
#pragma unroll
  for (int q = 0; q < WPT; q++) {
    unsigned mw = work_w[q], mz = fill_w[q];
    const int id0 = 32 * (w_first + q);
    while (mw) { const int bpos = __ffs(mw) - 1; mw &= mw - 1u; work[w++] = id0 + bpos; }       // (ascending ids: logical order)
    while (mz) { const int bpos = __ffs(mz) - 1; mz &= mz - 1u; skipped[z++] = id0 + bpos; }
  }
  if (threadIdx.x == 0) { lists[0] = all; lists[1] = all_z; }
}

#define PIX_SIDE_WGS 2
// (8 waves per SIMD, i.e. at most 64 VGPRs: the pixel workgroups need the occupancy; the two side workgroups fit in)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8))) void pix_fwd_kernel(int N, HeadFlags fl, float bg0, float bg1, float bg2, float min_val,
                                                      const float* __restrict__ mask_img, const float* __restrict__ omap,
                                                      const HgsViewTargets* __restrict__ tgt, float* __restrict__ partials,
                                                      float g_mask, float g_ori, float* __restrict__ d_unit, SsimGrid gd,
                                                      const unsigned char* __restrict__ zero_flags,
                                                      int* __restrict__ lists, HeadReduce h,
                                                      const float* __restrict__ p_ssim, const float* __restrict__ p_smooth,
                                                      float* __restrict__ out, const unsigned int* __restrict__ tile_used,
                                                      int tiles_x, int tiles_y, int W, float inv_w) {
  __shared__ float red[4];
  // The first two workgroups dispatched do the head's side jobs and nothing else (each is a chain of a few memory round
  // trips, several us apiece while the rest of the launch saturates HBM: as extra work of a pixel workgroup they made
  // that workgroup the launch's last, 20.5 -> 24 us; on their own 20.5 -> 22)
  if (blockIdx.x < PIX_SIDE_WGS) {
#if HGS_PIX_TRACE
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (blockIdx.x == 0) { if (lists) build_block_lists(gd, zero_flags, lists, tile_used, tiles_x, tiles_y); }
    else head_part1(h, p_ssim, p_smooth, out);
#if HGS_PIX_TRACE
    __syncthreads();
    if (threadIdx.x == 0 && lists) { lists[4 + 2 * gd.total + 2 * blockIdx.x] = (int)(unsigned)t0; lists[4 + 2 * gd.total + 2 * blockIdx.x + 1] = (int)(unsigned)__builtin_amdgcn_s_memrealtime(); }
#endif
    return;
  }
  const int blk = blockIdx.x - PIX_SIDE_WGS;
  const int i = blk * 256 + threadIdx.x;
  float s = 0.f, cnt = 0.f, b = 0.f;
  if (i < N) {
    // Every input of the pixel is loaded first, unconditionally (the targets of the orientation term also outside the
    // mask): the kernel is a chain of memory round trips otherwise -- logits, then the mask byte, then, behind the branch
    // on it, angle and confidence -- at 17 resident waves per CU.  (Round 4 tried the byte-saving forms again, now that the
    // launch moves 93 MB at ~5 TB/s: direction, angle and confidence behind a wave-uniform test of the mask byte, or of the
    // direction image for views without a mask: 19.2-19.4 us instead of 17.8-18.3 at north_star -- a wavefront's lifetime is
    // still two round trips -- and 5.0-6.5 instead of 7.2 us for the Stage-I cloud, +-1 % of the step either way.  What stayed:
    // with the consumer's tile hint, HgsHeadParams.tile_used, the four gradient planes are written only on tiles the blend
    // backward reads: 18.2 -> 17.8 us.)
    const bool has_mask = tgt->mask != nullptr;
    float xm = 0.f, ym = 0.f, o0 = 0.f, o1 = 0.f, o2 = 0.f, gt = 0.f, cf = 0.f;
    unsigned char mk = 0;
    unsigned used = 1u;
    if (fl.bce) { xm = mask_img[i]; ym = hgs_global(tgt->float_mask)[i]; }
    if (fl.ori) {
      o0 = omap[i]; o1 = omap[(size_t)N + i]; o2 = omap[2 * (size_t)N + i];
      // (without a mask the byte is read from the direction image and ignored: a load under `if (has_mask)` merged with 0
      // is waited for where it is issued)
      gt = hgs_global(tgt->orientation)[i]; cf = hgs_global(tgt->confidence)[i];
      mk = (has_mask ? hgs_global(tgt->mask) : (const HGS_GLOBAL unsigned char*)hgs_global(omap))[i];   // (last: its test is first)
    }
    if (d_unit && tile_used) {
      // pixel -> tile: y = i / W through the float reciprocal (exact below 2^24 pixels with the correction step; beyond, the
      // integer division)
      int y = N < (1 << 24) ? (int)(((float)i + 0.5f) * inv_w) : i / W;
      int x = i - y * W;
      if (x < 0) { y--; x += W; } else if (x >= W) { y++; x -= W; }
      static_assert(HGS_TILE == 16, "pixel -> tile by a shift of 4");
      used = tile_used[(y >> 4) * tiles_x + (x >> 4)];
    }
    float gm = 0.f, g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (fl.bce) {
      const float x = xm, y = ym;
      // (hardware exp2 / log2: en in (0, 1], so log(1 + en) is within 1e-7 absolute of log1p(en) -- of a term of order 0.1-1)
      const float en = __expf(-fabsf(x));
      b = fmaxf(x, 0.f) - x * y + __logf(1.f + en);
      if (d_unit) { const float r1 = __builtin_amdgcn_rcpf(1.f + en); gm = g_mask * ((x >= 0.f ? r1 : en * r1) - y); }   // sigmoid(x) - y
    }
    if (fl.ori) {
      OriParams p;
      p.view = tgt->viewmatrix; p.bg0 = bg0; p.bg1 = bg1; p.bg2 = bg2; p.min_val = min_val; p.has_mask = has_mask;
      const bool m = p.has_mask ? mk != 0 : (o0 != p.bg0 || o1 != p.bg1 || o2 != p.bg2);
      if (m) {
        float px, py, r, n, x, y, yq, th;
        ori_pixel(p, o0, o1, o2, px, py, r, n, x, y, yq, th);
        const float hp = 1.57079632679489661923f;
        s = (hp - fabsf(fabsf(th - gt) - hp)) * cf;
        cnt = 1.f;
        if (d_unit) ori_pixel_grad(p, px, py, r, n, x, yq, th, gt, cf, g_ori / tgt->mask_count, g0, g1, g2);
      }
    }
    if (d_unit && used) {
      d_unit[i] = gm; d_unit[(size_t)N + i] = g0; d_unit[2 * (size_t)N + i] = g1; d_unit[3 * (size_t)N + i] = g2;
    }
  }
  const float bs = block_sum(s, red);
  const float bc = block_sum(cnt, red);
  const float bb = block_sum(b, red);
  if (threadIdx.x == 0) { partials[3 * blk] = bs; partials[3 * blk + 1] = bc; partials[3 * blk + 2] = bb; }
#if HGS_PIX_TRACE
  if (threadIdx.x == 0 && lists && (blk == 0 || blk == (int)gridDim.x - PIX_SIDE_WGS - 1 || blk == 4000))
    lists[4 + 2 * gd.total + 4 + (blk == 0 ? 0 : (blk == 4000 ? 1 : 2))] = (int)(unsigned)__builtin_amdgcn_s_memrealtime();
#endif
}

__global__ __launch_bounds__(256) void pix_bwd_kernel(int N, HeadFlags fl, float bg0, float bg1, float bg2, float min_val,
                                                      const float* __restrict__ mask_img, const float* __restrict__ omap,
                                                      const HgsViewTargets* __restrict__ tgt, const float* __restrict__ out,
                                                      const float* __restrict__ go, float* __restrict__ d_mask_img,
                                                      float* __restrict__ d_omap) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const float up = *go;
  float gm = 0.f;
  if (fl.bce) {
    const float x = mask_img[i], y = hgs_global(tgt->float_mask)[i];
    gm = out[HGS_HEAD_G_MASK] * up * (1.f / (1.f + expf(-x)) - y);
  }
  d_mask_img[i] = gm;
  float g0 = 0.f, g1 = 0.f, g2 = 0.f;
  if (fl.ori) {
    OriParams p;
    p.view = tgt->viewmatrix; p.bg0 = bg0; p.bg1 = bg1; p.bg2 = bg2; p.min_val = min_val; p.has_mask = tgt->mask != nullptr;
    const float o0 = omap[i], o1 = omap[(size_t)N + i], o2 = omap[2 * (size_t)N + i];
    const bool m = p.has_mask ? hgs_global(tgt->mask)[i] != 0 : (o0 != p.bg0 || o1 != p.bg1 || o2 != p.bg2);
    if (m) {
      float px, py, r, n, x, y, yq, th;
      ori_pixel(p, o0, o1, o2, px, py, r, n, x, y, yq, th);
      ori_pixel_grad(p, px, py, r, n, x, yq, th, hgs_global(tgt->orientation)[i], hgs_global(tgt->confidence)[i],
                     (out[HGS_HEAD_G_ORI] * up) / out[HGS_HEAD_ORI_COUNT], g0, g1, g2);
    }
  }
  d_omap[i] = g0; d_omap[(size_t)N + i] = g1; d_omap[2 * (size_t)N + i] = g2;
}

// One block: fixed-order sums of the three partial arrays (bitwise reproducible), the loss terms, the total, and the
// derivative of the total w.r.t. each term (what the backward kernels scale by).

__global__ __launch_bounds__(256) void head_tail_kernel(HgsHeadTail t) { hgs_head_tail_block(t); }

}  // namespace

#if HGS_SSIM_TRACE
extern "C" int hgs_debug_set_ssim_trace(void* fwd, void* bwd) {
  void* p[2] = {fwd, bwd};
  HGS_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_ssim_trace), p, sizeof(p)));
  return 0;
}
#endif

extern "C" {

size_t hgs_ssim_l1_scratch_floats(int C, int H, int W) {
  const size_t nb = (size_t)((W + LT - 1) / LT) * ((H + LT - 1) / LT) * C;
  return 3 * (size_t)C * H * W + 2 * nb;
}
int hgs_ssim_l1_num_blocks(int C, int H, int W) { return ((W + LT - 1) / LT) * ((H + LT - 1) / LT) * C; }

int hgs_ssim_l1_forward(void* stream, int C, int H, int W, const float* window11_host, const float* img1, const float* img2,
                        float* dmaps, float* partials) {
  if (!window11_host || !img1 || !img2 || !dmaps || !partials || C <= 0 || H <= 0 || W <= 0) {
    hgs_set_error("hgs_ssim_l1_forward: bad arguments");
    return 1;
  }
  SsimWin win;
  for (int k = 0; k < 11; k++) win.w[k] = window11_host[k];
  hipStream_t s = (hipStream_t)stream;
  {
    HgsProfScope _prof(s, HGS_K_SSIM_FWD);
    hipLaunchKernelGGL(ssim_l1_fwd_kernel, dim3(ssim_grid_size(ssim_grid(C, H, W), SSIM_FWD_WG_PER_XCD)), dim3(256), 0, s, H, W, ssim_grid(C, H, W), win, img1,
                       img2, (const HgsViewTargets*)nullptr, dmaps, partials, (unsigned char*)nullptr);
  }
  HGS_CHECK_LAUNCH();
  return 0;
}

int hgs_ssim_l1_backward(void* stream, int C, int H, int W, const float* window11_host, const float* img1, const float* img2,
                         const float* dmaps, const float* g_ssim_mean, const float* g_l1_mean, float* dL_dimg1) {
  if (!window11_host || !img1 || !img2 || !dmaps || !g_ssim_mean || !g_l1_mean || !dL_dimg1) {
    hgs_set_error("hgs_ssim_l1_backward: bad arguments");
    return 1;
  }
  SsimWin win;
  for (int k = 0; k < 11; k++) win.w[k] = window11_host[k];
  hipStream_t s = (hipStream_t)stream;
  {
    HgsProfScope _prof(s, HGS_K_SSIM_BWD);
    hipLaunchKernelGGL(ssim_l1_bwd_kernel, dim3(ssim_grid_size(ssim_grid(C, H, W), SSIM_BWD_WG_PER_XCD)), dim3(256), 0, s, H, W, ssim_grid(C, H, W), win, img1,
                       img2, (const HgsViewTargets*)nullptr, dmaps, g_ssim_mean, g_l1_mean, (const float*)nullptr, dL_dimg1,
                       (float*)nullptr, 0, (const int*)nullptr, (const unsigned char*)nullptr);
  }
  HGS_CHECK_LAUNCH();
  return 0;
}

static OriParams ori_params(const float* viewmatrix, const float* bg3_host, float min_val, const uint8_t* mask) {
  OriParams p;
  p.view = viewmatrix;
  p.bg0 = bg3_host[0]; p.bg1 = bg3_host[1]; p.bg2 = bg3_host[2];
  p.min_val = min_val;
  p.has_mask = mask != nullptr;
  return p;
}

int hgs_orientation_loss_num_blocks(int H, int W) { return (int)(((size_t)H * W + 255) / 256); }

int hgs_orientation_loss_forward(void* stream, int H, int W, const float* omap, const float* viewmatrix,
                                 const float* bg3_host, float min_val, const float* gt_theta, const float* confidence,
                                 const uint8_t* mask, float* partials) {
  if (!omap || !viewmatrix || !bg3_host || !gt_theta || !confidence || !partials) { hgs_set_error("hgs_orientation_loss_forward: null argument"); return 1; }
  const int N = H * W;
  hipStream_t s = (hipStream_t)stream;
  {
    HgsProfScope _prof(s, HGS_K_ORI_FWD);
    hipLaunchKernelGGL(ori_fwd_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, ori_params(viewmatrix, bg3_host, min_val, mask),
                       omap, gt_theta, confidence, mask, partials);
  }
  HGS_CHECK_LAUNCH();
  return 0;
}

int hgs_orientation_loss_backward(void* stream, int H, int W, const float* omap, const float* viewmatrix,
                                  const float* bg3_host, float min_val, const float* gt_theta, const float* confidence,
                                  const uint8_t* mask, const float* g_loss, const float* mask_count, float* d_omap) {
  if (!omap || !viewmatrix || !bg3_host || !gt_theta || !confidence || !g_loss || !mask_count || !d_omap) { hgs_set_error("hgs_orientation_loss_backward: null argument"); return 1; }
  const int N = H * W;
  hipStream_t s = (hipStream_t)stream;
  {
    HgsProfScope _prof(s, HGS_K_ORI_BWD);
    hipLaunchKernelGGL(ori_bwd_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, ori_params(viewmatrix, bg3_host, min_val, mask),
                       omap, gt_theta, confidence, mask, g_loss, mask_count, d_omap);
  }
  HGS_CHECK_LAUNCH();
  return 0;
}

// ---- loss head ------------------------------------------------------------------------------------------------------
static inline int head_nb_ssim(const HgsHeadParams* p) { return ((p->W + LT - 1) / LT) * ((p->H + LT - 1) / LT) * 3; }
static inline int head_nb_pix(const HgsHeadParams* p) { return (int)(((size_t)p->H * p->W + 255) / 256); }
static inline int head_nb_smooth(const HgsHeadParams* p) { return p->lambda_smooth > 0.f ? (p->n_smooth + 255) / 256 : 0; }
// scratch: [dmaps 9*H*W][ssim partials 2*nb][pix partials 3*nb][smooth partials 2*nb][all-zero flags, 1 byte per SSIM block][as many spare bytes: the list builder reads whole 32-byte runs][block lists]
static inline size_t head_flags_offset(const HgsHeadParams* p) {
  return 9 * (size_t)p->H * p->W + 2 * (size_t)head_nb_ssim(p) + 3 * (size_t)head_nb_pix(p) + 2 * (size_t)head_nb_smooth(p) + 8;
}
static inline size_t head_flag_floats(const HgsHeadParams* p) { return ((size_t)head_nb_ssim(p) + 3) / 4 + 4; }
static inline unsigned char* head_zero_flags(const HgsHeadParams* p, float* scratch) { return (unsigned char*)(scratch + head_flags_offset(p)); }
// [n_work, n_skip, -, -][work ids][skipped ids] after the flags; NULL when the frame has more blocks than the list
// builder's LDS bitmap holds (the backward then walks every block)
static inline int* head_block_lists(const HgsHeadParams* p, float* scratch) {
  if (head_nb_ssim(p) > HEAD_MAX_FLAGGED) return nullptr;
  return (int*)(scratch + head_flags_offset(p) + 2 * head_flag_floats(p));
}
size_t hgs_loss_head_scratch_floats(const HgsHeadParams* p) {
  return head_flags_offset(p) + 2 * head_flag_floats(p) + 4 + 2 * (size_t)head_nb_ssim(p) + 12;   // (+ 8 words of HGS_PIX_TRACE stamps)
}

int hgs_loss_head_tail(const HgsHeadParams* p, const float* scratch, float* out, HgsHeadTail* tail) {
  if (!p || !scratch || !out || !tail) { hgs_set_error("hgs_loss_head_tail: null argument"); return 1; }
  tail->pix_partials = scratch + 9 * (size_t)p->H * p->W + 2 * (size_t)head_nb_ssim(p);
  tail->nb_pix = head_nb_pix(p);
  tail->out = out;
  tail->inv_hw = 1.f / (float)((size_t)p->H * p->W);
  tail->l_mask = p->lambda_mask; tail->l_ori = p->lambda_orientation; tail->l_smooth = p->lambda_smooth;
  tail->bce = p->lambda_mask > 0.f; tail->ori = p->lambda_orientation > 0.f; tail->smooth = head_nb_smooth(p) > 0;
  return 0;
}

int hgs_loss_head_forward(void* stream, const HgsHeadParams* p, const float* image, const float* mask_img,
                          const float* omap, const HgsViewTargets* targets, const float* endpoints,
                          const long long* smooth_pairs, float* scratch, float* out, float* d_extra_unit,
                          const float* smooth_partials_ext) {
  if (!p || !image || !mask_img || !omap || !targets || !scratch || !out || p->H <= 0 || p->W <= 0) {
    hgs_set_error("hgs_loss_head_forward: bad arguments");
    return 1;
  }
  const int H = p->H, W = p->W, N = H * W;
  // the tile hint indexes the rasterizer's tile grid of THIS frame: any other geometry would read the wrong tile (or out of bounds)
  // and leave needed planes of d_extra_unit unwritten
  if (p->tile_used && (p->tiles_x != (W + HGS_TILE - 1) / HGS_TILE || p->tiles_y != (H + HGS_TILE - 1) / HGS_TILE)) {
    hgs_set_error("hgs_loss_head_forward: tile_used given with tiles_x / tiles_y = %d x %d, the %d x %d frame has %d x %d tiles", p->tiles_x,
                  p->tiles_y, W, H, (W + HGS_TILE - 1) / HGS_TILE, (H + HGS_TILE - 1) / HGS_TILE);
    return 1;
  }
  const int nbs = head_nb_ssim(p), nbp = head_nb_pix(p), nbm = head_nb_smooth(p);
  if (nbm > 0 && !smooth_partials_ext && (!endpoints || !smooth_pairs)) { hgs_set_error("hgs_loss_head_forward: smoothness term without endpoints"); return 1; }
  float* dmaps = scratch;
  float* p_ssim = dmaps + 9 * (size_t)N;
  float* p_pix = p_ssim + 2 * (size_t)nbs;
  const float* p_smooth = smooth_partials_ext ? smooth_partials_ext : p_pix + 3 * (size_t)nbp;
  SsimWin win;
  for (int k = 0; k < 11; k++) win.w[k] = p->window[k];
  hipStream_t s = (hipStream_t)stream;
  int* lists = head_block_lists(p, scratch);
  {
    HgsProfScope _prof(s, HGS_K_SSIM_FWD);
    hipLaunchKernelGGL(ssim_l1_fwd_kernel, dim3(ssim_grid_size(ssim_grid(3, H, W), SSIM_FWD_WG_PER_XCD)), dim3(256), 0, s, H, W, ssim_grid(3, H, W), win, image,
                       (const float*)nullptr, targets, dmaps, p_ssim, lists ? head_zero_flags(p, scratch) : nullptr);
  }
  HeadFlags fl;
  fl.bce = p->lambda_mask > 0.f; fl.ori = p->lambda_orientation > 0.f;   // a NULL float_mask with lambda_mask > 0 is the caller's error
  if (nbm > 0 && !smooth_partials_ext &&
      hgs_launch_smooth_fwd(s, p->n_smooth, endpoints, smooth_pairs, p->cos_threshold, p->eps, p_pix + 3 * (size_t)nbp)) return 1;
  HeadReduce h;
  h.nb_ssim = nbs; h.nb_pix = nbp; h.nb_smooth = nbm;
  h.inv_chw = 1.f / (3.f * (float)N); h.inv_hw = 1.f / (float)N;
  h.l_dssim = p->lambda_dssim; h.l_mask = p->lambda_mask; h.l_ori = p->lambda_orientation; h.l_smooth = p->lambda_smooth;
  h.bce = fl.bce; h.ori = fl.ori;
  {
    HgsProfScope _prof(s, HGS_K_ORI_FWD);
    hipLaunchKernelGGL(pix_fwd_kernel, dim3(nbp + PIX_SIDE_WGS), dim3(256), 0, s, N, fl, p->bg[0], p->bg[1], p->bg[2], p->min_val, mask_img,
                       omap, targets, p_pix, fl.bce ? p->lambda_mask / (float)N : 0.f, fl.ori ? p->lambda_orientation : 0.f,
                       d_extra_unit, ssim_grid(3, H, W), (const unsigned char*)head_zero_flags(p, scratch), lists, h,
                       (const float*)p_ssim, p_smooth, out, p->tile_used, p->tiles_x, p->tiles_y, W, 1.f / (float)W);
  }
  if (!p->defer_tail) {
    HgsHeadTail t;
    hgs_loss_head_tail(p, scratch, out, &t);
    HgsProfScope _prof(s, HGS_K_HEAD);
    hipLaunchKernelGGL(head_tail_kernel, dim3(1), dim3(256), 0, s, t);
  }
  HGS_CHECK_LAUNCH();
  return 0;
}

int hgs_loss_head_backward(void* stream, const HgsHeadParams* p, const float* image, const float* mask_img,
                           const float* omap, const HgsViewTargets* targets, const float* endpoints,
                           const long long* smooth_pairs, const float* scratch, const float* out,
                           const float* grad_out, int skip, float* d_image, float* d_mask_img,
                           float* d_omap, float* d_endpoints) {
  const int skip_pixel_pass = skip & HGS_HEAD_SKIP_PIXELS;
  if (!p || !image || !mask_img || !omap || !targets || !scratch || !out || !grad_out || !d_image ||
      (!skip_pixel_pass && (!d_mask_img || !d_omap))) {
    hgs_set_error("hgs_loss_head_backward: bad arguments");
    return 1;
  }
  const int H = p->H, W = p->W, N = H * W;
  const int nbp = head_nb_pix(p), nbm = head_nb_smooth(p);
  const float* dmaps = scratch;
  SsimWin win;
  for (int k = 0; k < 11; k++) win.w[k] = p->window[k];
  hipStream_t s = (hipStream_t)stream;
  if (p->defer_tail && !skip_pixel_pass) {   // pix_bwd_kernel reads out[HGS_HEAD_ORI_COUNT]: the deferred tail cannot wait
    HgsHeadTail t;
    hgs_loss_head_tail(p, scratch, (float*)out, &t);
    HgsProfScope _prof(s, HGS_K_HEAD);
    hipLaunchKernelGGL(head_tail_kernel, dim3(1), dim3(256), 0, s, t);
  }
  {
    HgsProfScope _prof(s, HGS_K_SSIM_BWD);
    hipLaunchKernelGGL(ssim_l1_bwd_kernel, dim3(ssim_grid_size(ssim_grid(3, H, W), SSIM_BWD_WG_PER_XCD)), dim3(256), 0, s, H, W, ssim_grid(3, H, W), win, image,
                       (const float*)nullptr, targets, dmaps, out + HGS_HEAD_G_SSIM, out + HGS_HEAD_G_L1, grad_out, d_image,
                       d_endpoints, d_endpoints ? p->n_endpoints * 3 : 0, head_block_lists(p, (float*)scratch),
                       head_block_lists(p, (float*)scratch) ? (const unsigned char*)head_zero_flags(p, (float*)scratch) : nullptr);
  }
  HeadFlags fl;
  fl.bce = p->lambda_mask > 0.f; fl.ori = p->lambda_orientation > 0.f;
  if (!skip_pixel_pass) {
    HgsProfScope _prof(s, HGS_K_ORI_BWD);
    hipLaunchKernelGGL(pix_bwd_kernel, dim3(nbp), dim3(256), 0, s, N, fl, p->bg[0], p->bg[1], p->bg[2], p->min_val, mask_img,
                       omap, targets, out, grad_out, d_mask_img, d_omap);
  }
  if (d_endpoints && !(skip & HGS_HEAD_SKIP_SMOOTH)) {   // (cleared by the SSIM kernel above)
    if (nbm > 0 && hgs_launch_smooth_bwd(s, p->n_smooth, endpoints, smooth_pairs, p->cos_threshold, p->eps,
                                         out + HGS_HEAD_G_SMOOTH, out + HGS_HEAD_SMOOTH_COUNT, grad_out, d_endpoints)) return 1;
  }
  HGS_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
