// hgs_blend.hip -- per-tile alpha compositing, forward and backward, for wave64.
//
// replaces renderCUDA (cuda_rasterizer/forward.cu:261-374) and renderCUDABW_* (backward_distwar.cu:400-1014).
//
// CDNA4 mapping (not the reference's 256-entry shared-memory staging):
//   * a 16x16 tile = one 256-thread workgroup = 4 wavefronts; wavefront w owns the 8x8 pixel quadrant
//     (w&1, w>>1): compact footprints make whole-wave rejection of thin strand Gaussians likely;
//   * the tile's depth-sorted instance records (48 B: xy, conic, opacity, rgb, id -- 64 B with the 4 extra channels of
//     the single-pass mode; written by sort_tiles_kernel) are wave-uniform data: they are streamed with SCALAR loads (s_load_dwordx4) straight
//     into SGPRs and used as SGPR operands of the per-pixel VALU math -- no LDS staging, no barriers in the
//     forward, each wavefront leaves its loop on its own when its 64 pixels are saturated;
//   * backward: per (wave, entry) the 9 partial sums are reduced across the 64 lanes, the 4 wavefronts'
//     results are combined through LDS in fixed order and written ONCE per (tile, entry) with plain stores
//     into an instance-indexed scratch; preprocess_bwd gathers them per Gaussian through the inverse index.
//     No float atomics anywhere -> gradients are bitwise reproducible run to run (the reference's are not).
#include "hgs_common.h"

// development aid: per-workgroup start/end timestamps (hgs_debug_set_wg_trace)
__device__ unsigned long long* g_wg_trace_fwd = nullptr;
__device__ unsigned long long* g_wg_trace_bwd = nullptr;
struct WgTrace {
  unsigned long long* buf; int tile;
  __device__ WgTrace(unsigned long long* b, int t) : buf(b), tile(t) {
    if (buf && threadIdx.x == 0) buf[2 * tile] = __builtin_amdgcn_s_memrealtime();
  }
  __device__ ~WgTrace() {
    if (buf && threadIdx.x == 0) buf[2 * tile + 1] = __builtin_amdgcn_s_memrealtime();
  }
};

namespace {

#define BWD_BATCH 64   // entries combined per LDS flush in the backward
#define REC_BATCH 64   // instance records staged in LDS per batch (<= one float4 per thread)

// ---- wavefront-wide reduction of 9 per-lane values (CDNA4: v_permlane32_swap / v_permlane16_swap + DPP) ----------
// Transposing butterfly: a swap of the upper half-wave of `a` with the lower half-wave of `b` followed by one add
// folds TWO values by a factor 2 into ONE register; the same with 16-lane rows folds four values into one register
// with one value per row; bank-masked DPP adds continue the transposition inside the rows (row_transpose_sum).
__device__ __forceinline__ float fold32(float a, float b) {  // lanes 0-31: a[l]+a[l+32]; lanes 32-63: b[l-32]+b[l]
  auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
  unsigned lo = r[0], hi = r[1];
  asm volatile("" : "+v"(hi));  // ROCm 7.2 hipcc folds r[0] + r[1] of the swap builtin into r[0] + r[0] without this
  return __builtin_bit_cast(float, lo) + __builtin_bit_cast(float, hi);
}
__device__ __forceinline__ float fold16(float a, float b) {  // rows: (a.r0+a.r1, b.r0+b.r1, a.r2+a.r3, b.r2+b.r3)
  auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
  unsigned lo = r[0], hi = r[1];
  asm volatile("" : "+v"(hi));  // same miscompile guard as fold32
  return __builtin_bit_cast(float, lo) + __builtin_bit_cast(float, hi);
}
// Channel layouts.  C = 3: the reference's RGB pass.  C = 7 (single-pass mode): RGB + 4 extra
// unclamped channels (mask, world-space direction xyz) blended with the SAME weights in one traversal, which is what
// the reference's three render() calls per iteration compute separately (train.py:146, loss/losses.py:247,312).
//   record  : [x, y, conic a, b, c, opacity, f0 .. f(C-1), id, quadrant mask, pad]   REC4(C) float4 per instance
//   partials: [S(u dx), S(u dy), S(u dx dx), S(u dx dy), S(u dy dy), S(u) = dopacity, dcolor 0..C-1, (C>3: S(u_rgb dx), S(u_rgb dy))]
//             with u = G dL/dalpha: the moments from which preprocess_bwd_kernel forms dmean2D and dconic (see blend_bwd_kernel)
// The RGB-only screen-space gradient is what the reference's densification statistics see (the mask / orientation
// passes use their own throw-away screenspace tensors), so it is accumulated separately from the total.
template <int C> struct Chan {
  static constexpr int REC4 = (C <= 3) ? 3 : 4;                    // float4 per packed record
  static constexpr int NPART = (C <= 3) ? 9 : 6 + C + 2;           // partial sums per (tile, entry)
  static constexpr int NREG = (NPART + 3) / 4;                     // registers after the transposing fold
  static constexpr int ROW = (C <= 3) ? HGS_INST_GRAD_FLOATS : 16; // floats per instance row in the scratch
};

// in : v[0 .. 4*NREG) per lane.  out: x[r] rows (0,1,2,3) hold the wave totals of v[4r], v[4r+2], v[4r+1], v[4r+3].
// In-row part of the reduction, transposing as well.  After fold16 each of the NREG registers holds, in every 16-lane row,
// 16 partial sums of one value.  Bank-masked DPP adds (a bank = 4 lanes; masked-off lanes keep the destination) fold
// them so that ONE register ends with: row r, quad q  ->  total of value 4 * reg(q) + k(r), reg = (0,2,1,3)[q],
// k = (0,2,1,3)[r], the same number in all four lanes of the quad:
//   x0.lanes 0-7  = x0 + x0 rotated by 8      x0.lanes 8-15 = x1 + x1 rotated by 8        (x2 / x3 likewise)
//   x0.quads 0,2  = x0 + x0 rotated by 12     x0.quads 1,3  = x2 + x2 rotated by 4
//   + the two quad butterflies
// 8 DPP adds instead of the 16 of four independent row sums.  (row_ror:n: lane i reads lane i - n of its row.)  s_nop:
// a DPP operand written by one of the two preceding VALU instructions needs the wait states.
#define HGS_DPP(ctrl, banks) " " ctrl " row_mask:0xf bank_mask:" banks "\n\t"
__device__ __forceinline__ float row_transpose_sum(float x0, float x1, float x2, float x3) {
  asm volatile("s_nop 1\n\t"
               "v_add_f32_dpp %0, %0, %0" HGS_DPP("row_ror:8", "0x3")
               "v_add_f32_dpp %1, %1, %1" HGS_DPP("row_ror:8", "0x3")
               "v_add_f32_dpp %0, %2, %2" HGS_DPP("row_ror:8", "0xc")
               "v_add_f32_dpp %1, %3, %3" HGS_DPP("row_ror:8", "0xc")
               "s_nop 1\n\t"
               "v_add_f32_dpp %0, %0, %0" HGS_DPP("row_ror:12", "0x5")
               "v_add_f32_dpp %0, %1, %1" HGS_DPP("row_ror:4", "0xa")
               "s_nop 1\n\t"
               "v_add_f32_dpp %0, %0, %0" HGS_DPP("quad_perm:[1,0,3,2]", "0xf")
               "s_nop 1\n\t"
               "v_add_f32_dpp %0, %0, %0" HGS_DPP("quad_perm:[2,3,0,1]", "0xf")
               : "+v"(x0), "+v"(x2) : "v"(x1), "v"(x3));
  return x0;
}
// in : v[0 .. 4*NREG) per lane.  out: lane 16 r + 4 q (+0..3) holds the wave total of v[4 * (0,2,1,3)[q] + (0,2,1,3)[r]]
template <int NREG>
__device__ __forceinline__ float wave_reduce(const float* v) {
  float x[4];
#pragma unroll
  for (int r = 0; r < NREG; r++) x[r] = fold16(fold32(v[4 * r], v[4 * r + 1]), fold32(v[4 * r + 2], v[4 * r + 3]));
#pragma unroll
  for (int r = NREG; r < 4; r++) x[r] = 0.f;
  return row_transpose_sum(x[0], x[1], x[2], x[3]);
}

// ---- record staging ------------------------------------------------------------------------------------------
// A tile's list is consumed in batches of REC_BATCH entries.  One batch is REC_BATCH * REC4 float4 = at most one float4
// per thread, fetched with a single fully coalesced global load per thread; the NEXT batch's load is issued before the
// current batch is evaluated, so the HBM / Infinity-Cache round trip (about 1 us on this part, and the whole critical
// path of a long tile when it is paid per handful of entries) overlaps the math.  Double-buffered LDS, one barrier per
// batch in the forward.
//
// Each wavefront then builds, with one ballot over the records' quadrant masks (sort_tiles_kernel), the 64-bit set of
// batch entries whose alpha >= 1/255 footprint can touch ITS 8x8 pixels, and walks only those bits on the scalar unit
// (s_ff1 / s_flbit); for thin strand Gaussians that is about 30% of the tile's entries.  The selected record is read
// from LDS with wave-uniform (broadcast) ds_read_b128, one entry ahead of the one being evaluated.
template <int C> struct Rec { float4 q[Chan<C>::REC4]; };

template <int C>
__device__ __forceinline__ Rec<C> lds_record(const float4* recs, int e) {
  Rec<C> r;
#pragma unroll
  for (int w = 0; w < Chan<C>::REC4; w++) r.q[w] = recs[e * Chan<C>::REC4 + w];
  return r;
}

// ------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(HGS_BLOCK) void blend_fwd_kernel(const uint2* __restrict__ ranges,
                                                              const float4* __restrict__ packed, int W, int H, int gx,
                                                              uint32_t Rcap, const float* __restrict__ bg,
                                                              float* __restrict__ final_T,
                                                              uint32_t* __restrict__ n_contrib,
                                                              uint32_t* __restrict__ tile_maxc,
                                                              const uint32_t* __restrict__ tile_order,
                                                              float* __restrict__ out_color) {
  constexpr int REC4 = Chan<C>::REC4;
  __shared__ float4 recs[2][REC_BATCH * REC4];
  __shared__ uint32_t alive[2][4];
  const int tile = (int)tile_order[blockIdx.x];   // longest lists first (scan_kernel)
  WgTrace _trace(g_wg_trace_fwd, tile);
  const int tx = tile % gx, ty = tile / gx;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int px = tx * HGS_TILE + (wave & 1) * 8 + (lane & 7);
  const int py = ty * HGS_TILE + (wave >> 1) * 8 + (lane >> 3);
  const bool inside = px < W && py < H;
  const float pxf = (float)px, pyf = (float)py;
  uint2 range = ranges[tile];
  if (range.y > Rcap) range = make_uint2(0u, 0u);  // binning buffer under-sized (flagged by the scatter kernel)
  const uint32_t L = range.y - range.x;
  const int nb = (int)((L + REC_BATCH - 1) / REC_BATCH);
  // T < 0 marks a pixel that is done (saturated, forward.cu:346-351, or outside the image): its magnitude stays the
  // transmittance it stopped at.  A separate per-lane flag costs a byte register and ~6 vector instructions per entry to
  // test, merge and update; the sign costs one compare.
  float T = inside ? 1.f : -1.f, acc[C];
#pragma unroll
  for (int k = 0; k < C; k++) acc[k] = 0.f;
  uint32_t last = 0;

  const float4* src = packed + (size_t)range.x * REC4;
  const uint32_t nf4 = L * REC4;
  float4 stage = make_float4(0.f, 0.f, 0.f, 0.f);
  if (nb > 0) {
    if (threadIdx.x < REC_BATCH * REC4 && threadIdx.x < nf4) stage = src[threadIdx.x];
    if (threadIdx.x < REC_BATCH * REC4) recs[0][threadIdx.x] = stage;
    if (threadIdx.x < 4) alive[0][threadIdx.x] = 1u;
  }
  for (int b = 0; b < nb; b++) {
    const int cur = b & 1;
    if (b + 1 < nb) {
      const uint32_t i = (uint32_t)(b + 1) * (REC_BATCH * REC4) + threadIdx.x;
      if (threadIdx.x < REC_BATCH * REC4 && i < nf4) stage = src[i];
    }
    __syncthreads();
    // forward.cu:309-311: the tile stops when every pixel is saturated (flags written before the barrier above)
    if ((alive[cur][0] | alive[cur][1] | alive[cur][2] | alive[cur][3]) == 0u) break;
    const int cnt = min(REC_BATCH, (int)L - b * REC_BATCH);
    const float* rf = (const float*)&recs[cur][0];
    const uint32_t mk = lane < cnt ? __float_as_uint(rf[lane * 4 * REC4 + 7 + C]) : 0u;
    uint64_t m = __ballot(((mk >> wave) & 1u) != 0u);
    if (__ballot(T > 0.f) == 0) m = 0;
    bool wave_done = false;   // every pixel of this wavefront saturated: leave the batch
    auto process = [&](const Rec<C>& r, int e) {
      const float4 r0 = r.q[0], r1 = r.q[1];
      const float dx = r0.x - pxf, dy = r0.y - pyf;
      const float power = -0.5f * (r0.z * dx * dx + r1.x * dy * dy) - r0.w * dx * dy;  // forward.cu:335
      const float alpha = fminf(0.99f, r1.y * __expf(power));                           // :343
      const bool ok = T > 0.f && power <= 0.f && alpha >= (1.0f / 255.0f);              // :336, :344
      if (__ballot(ok) == 0) return;
      const float test_T = T * (1.f - alpha);
      const bool sat = ok && test_T < 0.0001f;                                          // :346-351
      if (ok && !sat) {
        const float w = alpha * T;
        const float* f = (const float*)&r.q[0];                                         // features start at float 6
#pragma unroll
        for (int k = 0; k < C; k++) acc[k] += f[6 + k] * w;                             // :354-355
        last = (uint32_t)(b * REC_BATCH + e + 1);                                       // :328, :361
      }
      if (ok) T = sat ? -T : test_T;
      if (__ballot(sat) != 0 && __ballot(T > 0.f) == 0) wave_done = true;
    };
    if (m) {
      // two record buffers used alternately (LDS reads of the next entry in flight, no register rotation)
      int ea = __builtin_ctzll(m), eb = 0;
      Rec<C> ra = lds_record<C>(recs[cur], ea), rb = ra;
      while (true) {
        m &= m - 1;
        if (m) { eb = __builtin_ctzll(m); rb = lds_record<C>(recs[cur], eb); }
        process(ra, ea);
        if (m == 0 || wave_done) break;
        m &= m - 1;
        if (m) { ea = __builtin_ctzll(m); ra = lds_record<C>(recs[cur], ea); }
        process(rb, eb);
        if (m == 0 || wave_done) break;
      }
    }
    if (b + 1 < nb) {
      if (threadIdx.x < REC_BATCH * REC4) recs[cur ^ 1][threadIdx.x] = stage;
      const uint32_t wave_alive = __ballot(T > 0.f) != 0 ? 1u : 0u;   // all lanes vote: taken outside the lane-0 branch
      if (lane == 0) alive[cur ^ 1][wave] = wave_alive;
    }
  }
  uint32_t wmax = last;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, d, 64));
  if (lane == 0 && wmax) atomicMax(&tile_maxc[tile], wmax);
  if (inside) {
    const size_t pix = (size_t)py * W + px, HW = (size_t)H * W;
    T = fabsf(T);
    final_T[pix] = T;
    n_contrib[pix] = last;
#pragma unroll
    for (int k = 0; k < C; k++) out_color[k * HW + pix] = acc[k] + T * bg[k];           // :372
  }
}

// ------------------------------------------------------------------------------------------------
template <int C> struct PixGrad { const float* plane[C]; };  // dL/d(output channel k) as [H,W] planes (need not be adjacent)

template <int C>
__global__ __launch_bounds__(HGS_BLOCK) __attribute__((amdgpu_waves_per_eu(6))) void blend_bwd_kernel(const uint2* __restrict__ ranges,
                                                              const float4* __restrict__ packed, int W, int H, int gx,
                                                              uint32_t Rcap, const float* __restrict__ bg,
                                                              const float* __restrict__ final_Ts,
                                                              const uint32_t* __restrict__ n_contrib,
                                                              const uint32_t* __restrict__ tile_maxc,
                                                              PixGrad<C> dL_dpix, const uint32_t* __restrict__ tile_order,
                                                              float* __restrict__ inst_grad) {
  constexpr int REC4 = Chan<C>::REC4, NPART = Chan<C>::NPART, NREG = Chan<C>::NREG, NV = 4 * NREG, ROW = Chan<C>::ROW;
  static_assert(BWD_BATCH == REC_BATCH, "one record batch per partial-sum flush");
  __shared__ float part[4][BWD_BATCH][NV];
  __shared__ float4 recs[2][REC_BATCH * REC4];
  const int tile = (int)tile_order[blockIdx.x];   // longest lists first (scan_kernel)
  WgTrace _trace(g_wg_trace_bwd, tile);
  const uint2 range = ranges[tile];
  const uint32_t maxc = range.y > Rcap ? 0u : tile_maxc[tile];   // (list beyond an under-sized binning buffer: no gradients)
  {
    // Every instance row of the scratch is written by exactly one tile, so the scratch needs no clearing pass: entries
    // past the last one any pixel needed (positions >= maxc; ~2% of the instances) get explicit zero rows here.
    const uint32_t end = min(range.y, Rcap), first = range.x + maxc;
    if (end > first) {
      const uint32_t n4 = (end - first) * (ROW / 4);
      for (uint32_t i = threadIdx.x; i < n4; i += HGS_BLOCK) {
        const uint32_t inst = first + i / (ROW / 4);
        const uint32_t slot = __float_as_uint(((const float*)packed)[(size_t)inst * 4 * REC4 + 8 + C]);
        if (slot < Rcap) ((float4*)(inst_grad + (size_t)slot * ROW))[i % (ROW / 4)] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
  if (maxc == 0) return;  // nothing in this tile contributed to any pixel
  const int tx = tile % gx, ty = tile / gx;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int px = tx * HGS_TILE + (wave & 1) * 8 + (lane & 7);
  const int py = ty * HGS_TILE + (wave >> 1) * 8 + (lane >> 3);
  const bool inside = px < W && py < H;
  const float pxf = (float)px, pyf = (float)py;
  const size_t pix = (size_t)py * W + px;

  // first batch (the END of the list: the walk is back to front) is in flight while the per-pixel state is set up
  const float4* src = packed + (size_t)range.x * REC4;
  float4 stage = make_float4(0.f, 0.f, 0.f, 0.f);
  {
    const int lo = max(0, (int)maxc - BWD_BATCH), cnt = (int)maxc - lo;
    if (threadIdx.x < cnt * REC4) stage = src[(size_t)lo * REC4 + threadIdx.x];
  }

  const float T_final = inside ? final_Ts[pix] : 0.f;
  float T = T_final;
  const uint32_t last = inside ? n_contrib[pix] : 0u;
  uint32_t wlast = last;                                                       // wave-wide last contributor
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) wlast = max(wlast, (uint32_t)__shfl_xor((int)wlast, d, 64));
  wlast = __builtin_amdgcn_readfirstlane(wlast);
  // The colour accumulated behind an entry (accum_rec, :972) enters only through its dot product with this pixel's
  // upstream gradient (:979-984): the state carried per pixel is that scalar (and the RGB-only one), not C colours.
  float dpx[C], acc_dot = 0.f, acc_dot_rgb = 0.f;
  float bg_dot = 0.f, bg_dot_rgb = 0.f;                                       // backward_distwar.cu:988-990
#pragma unroll
  for (int k = 0; k < C; k++) {
    dpx[k] = inside ? dL_dpix.plane[k][pix] : 0.f;
    bg_dot += bg[k] * dpx[k];
    if (k < 3) bg_dot_rgb += bg[k] * dpx[k];
  }

  for (int i = threadIdx.x; i < 4 * BWD_BATCH * NV; i += HGS_BLOCK) (&part[0][0][0])[i] = 0.f;
  if (threadIdx.x < REC_BATCH * REC4) recs[0][threadIdx.x] = stage;
  __syncthreads();

  // walk the list back to front in batches of BWD_BATCH positions; position p (0-based) is valid for a pixel
  // iff p < n_contrib (backward_distwar.cu:943-945)
  int cur = 0;
  for (int hi = (int)maxc; hi > 0; hi -= BWD_BATCH, cur ^= 1) {
    const int lo = max(0, hi - BWD_BATCH);
    const int cnt = hi - lo;
    if (lo > 0) {  // next batch's records: in flight during this batch's math
      const int nlo = max(0, lo - BWD_BATCH), ncnt = lo - nlo;
      if (threadIdx.x < ncnt * REC4) stage = src[(size_t)nlo * REC4 + threadIdx.x];
    }
    const float* rf = (const float*)&recs[cur][0];
    const uint32_t mk = lane < cnt ? __float_as_uint(rf[lane * 4 * REC4 + 7 + C]) : 0u;
    uint64_t m = __ballot(((mk >> wave) & 1u) != 0u);
    // positions at or past this wavefront's last contributor cannot be valid for any of its pixels
    if ((int)wlast <= lo) m = 0;
    else if ((int)wlast - lo < 64) m &= (1ull << ((int)wlast - lo)) - 1ull;
    // one entry: evaluate, update the per-pixel state, reduce the NPART partial sums over the wavefront.  The gradient
    // arithmetic is branch-free: lanes that do not blend the entry run it with alpha = G = 0, which makes every
    // partial sum an exact zero (masked vector instructions cost the same as unmasked ones, and this way no register
    // has to be cleared per entry); only the state recurrences sit under the lane mask.
    auto process = [&](const Rec<C>& r, int e) {
      const int p = lo + e;
      const float* f = (const float*)&r.q[0];
      const float4 r0 = r.q[0], r1 = r.q[1];
      const float dx = r0.x - pxf, dy = r0.y - pyf;
      const float power = -0.5f * (r0.z * dx * dx + r1.x * dy * dy) - r0.w * dx * dy;
      const float Gx = __expf(power);
      const float ax = fminf(0.99f, r1.y * Gx);
      const bool ok = (uint32_t)p < last && power <= 0.f && ax >= (1.0f / 255.0f);
      if (__ballot(ok) == 0) return;
      const float G = ok ? Gx : 0.f, alpha = ok ? ax : 0.f;
      const float inv_one_m_a = __builtin_amdgcn_rcpf(1.f - alpha);               // 1 ulp; alpha <= 0.99; rcp(1) == 1
      T = T * inv_one_m_a;                                                         // :960
      float v[NV];
      const float dchannel_dcolor = alpha * T;
      // sum_k (c_k - accum_rec_k) dL_dpix_k = c . dL_dpix - accum_rec . dL_dpix  (:979-984)
      float col_dot = 0.f, col_dot_rgb = 0.f;
#pragma unroll
      for (int k = 0; k < C; k++) {
        col_dot = __builtin_fmaf(f[6 + k], dpx[k], col_dot);
        if (k == 2) col_dot_rgb = col_dot;
        v[6 + k] = dchannel_dcolor * dpx[k];                                       // :980
      }
      float dL_dalpha = col_dot - acc_dot;
      const float dL_dalpha_rgb = col_dot_rgb - acc_dot_rgb;
      dL_dalpha *= T;
      const float bgw = -T_final * inv_one_m_a;
      dL_dalpha += bgw * bg_dot;                                                   // :991
      // Everything downstream of dL/dalpha is LINEAR in u = G * dL_dalpha with per-Gaussian coefficients (:1002-1011):
      //   dL_dG = opacity * dL_dalpha, dG_ddelx = -G (a dx + b dy), dG_ddely = -G (c dy + b dx)
      //   dmean2D = dL_dG * dG_ddel{x,y} * ddel_d{x,y},  dconic = -0.5 * G * (dx dx, dx dy, dy dy) * dL_dG,  dopacity = u
      // so the wave sums are taken of the five moments u dx, u dy, u dx dx, u dx dy, u dy dy (and of u itself), and
      // preprocess_bwd_kernel applies opacity, conic and the 0.5 W / 0.5 H factors once per Gaussian after adding up its
      // instances' rows: 11 multiplications per (pixel, entry) here instead of 30.
      const float u = G * dL_dalpha;
      const float ux = u * dx, uy = u * dy;
      v[0] = ux;
      v[1] = uy;
      v[2] = ux * dx;
      v[3] = ux * dy;
      v[4] = uy * dy;
      v[5] = u;
      if (C > 3) {  // the same moments for the RGB channels alone (densification statistics see only those)
        const float u_rgb = G * (dL_dalpha_rgb * T + bgw * bg_dot_rgb);
        v[6 + C] = u_rgb * dx;
        v[7 + C] = u_rgb * dy;
      }
      // colour accumulated behind the NEXT (nearer) entry: the reference's accum_rec = last_alpha * last_color +
      // (1 - last_alpha) * accum_rec (:972), evaluated here, right after its operands were used, instead of at the next
      // contributing entry -- same operands, same order, and no (last_alpha, last_color) state to carry.  With
      // alpha = 0 (lanes that do not blend this entry) it is the identity.
      acc_dot = alpha * col_dot + (1.f - alpha) * acc_dot;
      acc_dot_rgb = alpha * col_dot_rgb + (1.f - alpha) * acc_dot_rgb;
#pragma unroll
      for (int k = NPART; k < NV; k++) v[k] = 0.f;
      const float tot = wave_reduce<NREG>(v);
      if ((lane & 3) == 0) {                            // one lane per quad stores the quad's value
        const int row = lane >> 4, quad = (lane >> 2) & 3;
        const int k = ((row & 1) << 1) | (row >> 1);    // rows hold values (0,2,1,3) of each group of four,
        const int reg = ((quad & 1) << 1) | (quad >> 1);  // quads the groups (0,2,1,3)
        if (reg < NREG) part[wave][e][4 * reg + k] = tot;
      }
    };
    if (m) {
      // two record buffers used alternately: the next entry's LDS reads are in flight while this one is evaluated,
      // and no register copies rotate the pipeline
      int ea = 63 - __builtin_clzll(m), eb = 0;
      Rec<C> ra = lds_record<C>(recs[cur], ea), rb = ra;
      while (true) {
        m &= ~(1ull << ea);
        if (m) { eb = 63 - __builtin_clzll(m); rb = lds_record<C>(recs[cur], eb); }
        process(ra, ea);
        if (m == 0) break;
        m &= ~(1ull << eb);
        if (m) { ea = 63 - __builtin_clzll(m); ra = lds_record<C>(recs[cur], ea); }
        process(rb, eb);
        if (m == 0) break;
      }
    }
    __syncthreads();
    // combine the 4 wavefronts in fixed order and store one row per (tile, entry)
    for (int i = threadIdx.x; i < cnt * NPART; i += HGS_BLOCK) {
      const int e = i / NPART, k = i - e * NPART;
      const float s = ((part[0][e][k] + part[1][e][k]) + part[2][e][k]) + part[3][e][k];
      // row of the instance's Gaussian-major slot (record pad, sort_tiles_kernel): a Gaussian's rows lie together
      const uint32_t slot = __float_as_uint(rf[e * 4 * REC4 + 8 + C]);
      if (slot < Rcap) inst_grad[(size_t)slot * ROW + k] = s;
      part[0][e][k] = 0.f; part[1][e][k] = 0.f; part[2][e][k] = 0.f; part[3][e][k] = 0.f;
    }
    if (lo > 0 && threadIdx.x < REC_BATCH * REC4) recs[cur ^ 1][threadIdx.x] = stage;
    __syncthreads();
  }
}

}  // namespace

extern "C" int hgs_debug_set_wg_trace(void* device_buf_fwd, void* device_buf_bwd) {
  HGS_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_wg_trace_fwd), &device_buf_fwd, sizeof(void*)));
  HGS_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_wg_trace_bwd), &device_buf_bwd, sizeof(void*)));
  return 0;
}

int hgs_launch_blend_fwd(hipStream_t s, int W, int H, int Rcap, int channels, const float* bg, const HgsImage& im,
                         const HgsBinning& b, float* out_color) {
  const int gx = (W + HGS_TILE - 1) / HGS_TILE, gy = (H + HGS_TILE - 1) / HGS_TILE;
  {
    HgsProfScope _prof(s, HGS_K_BLEND_FWD);
    if (channels == 3)
      hipLaunchKernelGGL(blend_fwd_kernel<3>, dim3(gx * gy), dim3(HGS_BLOCK), 0, s, im.ranges, b.packed, W, H, gx,
                         (uint32_t)Rcap, bg, im.final_T, im.n_contrib, im.tile_maxc, im.tile_order, out_color);
    else
      hipLaunchKernelGGL(blend_fwd_kernel<7>, dim3(gx * gy), dim3(HGS_BLOCK), 0, s, im.ranges, b.packed, W, H, gx,
                         (uint32_t)Rcap, bg, im.final_T, im.n_contrib, im.tile_maxc, im.tile_order, out_color);
  }
  HGS_CHECK_LAUNCH();
  return 0;
}

int hgs_launch_blend_bwd(hipStream_t s, int W, int H, int Rcap, int channels, const float* bg, const HgsImage& im,
                         const HgsBinning& b, const float* const* dL_dpix_planes, float* inst_grad) {
  const int gx = (W + HGS_TILE - 1) / HGS_TILE, gy = (H + HGS_TILE - 1) / HGS_TILE;
  {
    HgsProfScope _prof(s, HGS_K_BLEND_BWD);
    if (channels == 3) {
      PixGrad<3> pg;
      for (int k = 0; k < 3; k++) pg.plane[k] = dL_dpix_planes[k];
      hipLaunchKernelGGL(blend_bwd_kernel<3>, dim3(gx * gy), dim3(HGS_BLOCK), 0, s, im.ranges, b.packed, W, H, gx,
                         (uint32_t)Rcap, bg, im.final_T, im.n_contrib, im.tile_maxc, pg, im.tile_order, inst_grad);
    } else {
      PixGrad<7> pg;
      for (int k = 0; k < 7; k++) pg.plane[k] = dL_dpix_planes[k];
      hipLaunchKernelGGL(blend_bwd_kernel<7>, dim3(gx * gy), dim3(HGS_BLOCK), 0, s, im.ranges, b.packed, W, H, gx,
                         (uint32_t)Rcap, bg, im.final_T, im.n_contrib, im.tile_maxc, pg, im.tile_order, inst_grad);
    }
  }
  HGS_CHECK_LAUNCH();
  return 0;
}
