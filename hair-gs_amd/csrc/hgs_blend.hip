// hgs_blend.hip -- per-tile alpha compositing, forward and backward, for wave64.
//
// replaces renderCUDA (cuda_rasterizer/forward.cu:261-374) and renderCUDABW_* (backward_distwar.cu:400-1014).
//
// CDNA4 mapping (not the reference's 256-entry shared-memory staging):
//   * a 16x16 tile = one 256-thread workgroup = 4 wavefronts; wavefront w owns the 8x8 pixel quadrant
//     (w&1, w>>1): compact footprints make whole-wave rejection of thin strand Gaussians likely;
//   * the tile's depth-sorted instance records (48 B: xy, conic, opacity, rgb, id, quadrant mask, Gaussian-major slot --
//     64 B with the 4 extra channels of the single-pass mode; written by sort_tiles_kernel) are staged through LDS in
//     batches of 64 with one coalesced float4 load per thread, the next batch in flight during the math; every wavefront
//     ballots the batch's quadrant masks into a 64-bit set and walks only ITS entries on the scalar unit, reading the
//     record with wave-uniform ds_read_b128;
//   * backward: per (wave, entry) the partial sums are reduced across the 64 lanes (permlane swaps + bank-masked DPP),
//     the 4 wavefronts' results are combined through LDS in fixed order and written ONCE per (tile, entry) with plain
//     stores into the instance's Gaussian-major row of the scratch; preprocess_bwd sums a Gaussian's contiguous rows.
//     No float atomics anywhere -> gradients are bitwise reproducible run to run (the reference's are not);
//   * LONG LISTS ARE SPLIT ACROSS WORKGROUPS (work list of sort_tiles_kernel: segments of 256-1024 entries).  The
//     reference walks a tile's list with one thread block (forward.cu:295-362); a scene whose Gaussians pile up on a
//     few hundred tiles (Stage I: 1000-3000 entries per tile) then runs at the speed of one workgroup per tile.
//     Transmittance is a product and blending is linear in the transmittance in front, so segments compose:
//       forward  pass A:  every segment's workgroup runs the reference's walk over ITS entries from a transmittance of 1
//                         (no dependence on the segments in front) and publishes the per-pixel product (agent-scope
//                         stores + the tile's progress mask, hgs_common.h);
//                scaling: it reads its predecessors' products -- the true transmittance T_in in front of the segment.
//                         A pixel the walk would not have stopped on (T_in * product clear of the 1e-4 rule) takes its
//                         local colours times T_in; a pixel that is done before the segment contributes nothing;
//                pass B:  only the pixels that stop INSIDE this segment (one segment per pixel; waves without such a
//                         pixel skip) are walked again from T_in with the reference's rules (alpha test,
//                         T (1 - alpha) < 1e-4 stop), which fixes their colour, final transmittance and last contributor;
//                         the LAST workgroup of the tile to finish (ticket) combines the segments per pixel in list
//                         order: first stop wins, colours summed back to front (fixed order: reproducible), and leaves
//                         the SUFFIX sums of the segment colours in place;
//                (round 4: the FIRST segment's pass A starts from the true transmittance and is final -- no pass B --, and a
//                 list of exactly two segments is walked serially by its first segment's workgroup: two links of this
//                 chain take longer than the walk)
//       backward          needs no communication: a segment starts from the forward's transmittance in front of the NEXT
//                         segment and from the colour behind it (that suffix sum) -- exactly the state the serial walk
//                         would carry into it.
//     Numerical differences to the serial walk: the association of the transmittance product (one rounding per segment
//     boundary) and, for the segments a pixel passes through, c alpha T_local T_in in place of c alpha T (one rounding
//     per contribution).  Integer results (last contributor) follow the reference's rules on those products.
#include "hgs_common.h"

// development aid: per-workgroup start/end timestamps (hgs_debug_set_wg_trace)
__device__ unsigned long long* g_wg_trace_fwd = nullptr;
__device__ unsigned long long* g_wg_trace_bwd = nullptr;
#ifndef HGS_WG_TRACE
#define HGS_WG_TRACE 0   // build with -DHGS_WG_TRACE=1 (tools/build_variant.sh trace hgs_blend -DHGS_WG_TRACE=1) to record
#endif
struct WgTrace {   // 8 words per workgroup: start, marks 1..5, (tile | seg << 24 | nseg << 32 | list entries << 40), end
#if HGS_WG_TRACE
  unsigned long long* buf;
  __device__ WgTrace(unsigned long long* b) : buf(b ? b + 8 * (size_t)blockIdx.x : nullptr) { mark(0); }
  __device__ void mark(int k) const { if (buf && threadIdx.x == 0) buf[k] = __builtin_amdgcn_s_memrealtime(); }
  __device__ void item(int tile, uint32_t seg, uint32_t nseg, uint32_t len) const {
    if (buf && threadIdx.x == 0) buf[6] = (unsigned long long)tile | (unsigned long long)seg << 24 | (unsigned long long)nseg << 32 | (unsigned long long)len << 40;
  }
  // backward statistics (words 1..3 of the workgroup): (entry, wavefront) pairs evaluated / of those with no blending
  // lane / sum of blending lanes
  __device__ void count(int k, unsigned v) const { if (buf && (threadIdx.x & 63) == 0) atomicAdd(&buf[k], (unsigned long long)v); }
  __device__ ~WgTrace() { mark(7); }
#else
  __device__ WgTrace(unsigned long long*) {}
  __device__ void mark(int) const {}
  __device__ void item(int, uint32_t, uint32_t, uint32_t) const {}
  __device__ void count(int, unsigned) const {}
#endif
};

namespace {

// Backward: 32 entries staged and combined per LDS flush and one record in registers at a time keep the kernel at 53 VGPRs
// and 12 KB of LDS, i.e. 8 resident waves per SIMD: measured 2-4 % faster than 64-entry batches with a second record in
// flight (80 VGPRs, 24.5 KB, 6 waves) on every BASELINE workload.
#ifndef HGS_BWD_WAVES
#define HGS_BWD_WAVES 8
#endif
#ifndef BWD_BATCH
#define BWD_BATCH 32
#endif
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
  // (round 4: all half-wave folds first, then the row folds -- 5 instead of 9 hazard s_nop per entry -- measured no faster:
  // north_star 48.6 against 48.5 us, C4 398 against 391.5; the nested form stays)
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

// Quarter q of the record of list position `pos` (a lazy pass, hgs_common.h: from the sorted key and the Gaussian's template; what
// emit_instance of the sort kernel wrote otherwise).  The record's last quarter carries the instance's own words: Gaussian id,
// quadrant mask (the key's low bits), gradient-row slot (the Gaussian's first row + the cell of its tile rectangle).
template <int C>
__device__ __forceinline__ float4 stage_quarter(const HgsBinning& bn, bool lazy, uint32_t pos, int q, int tx, int ty) {
  constexpr int REC4 = Chan<C>::REC4;
  if (!lazy) return bn.packed[(size_t)pos * REC4 + q];
  const uint64_t key = bn.keys_sorted[pos];
  const uint32_t id = (uint32_t)key >> HGS_QMASK_SHIFT;
  const float4* t = bn.grec + 4 * (size_t)id;
  if (q < REC4 - 1) return t[q];
  const uint4 u3 = ((const uint4*)t)[3];
  const uint32_t slot = u3.y + ((uint32_t)ty - (u3.z >> 16)) * u3.w + ((uint32_t)tx - (u3.z & 0xFFFFu));
  const uint32_t w0 = C <= 3 ? __float_as_uint(t[2].x) : u3.x;
  return make_float4(__uint_as_float(w0), __uint_as_float(id), __uint_as_float((uint32_t)key & HGS_QMASK_BITS), __uint_as_float(slot));
}
// (id, slot) of list position `pos` without the rest of the record (the backward's zero rows)
template <int C>
__device__ __forceinline__ uint2 stage_id_slot(const HgsBinning& bn, bool lazy, uint32_t pos, int tx, int ty) {
  if (!lazy) {
    const float* r = (const float*)(bn.packed + (size_t)pos * Chan<C>::REC4);
    return make_uint2(__float_as_uint(r[6 + C]), __float_as_uint(r[8 + C]));
  }
  const uint32_t id = (uint32_t)bn.keys_sorted[pos] >> HGS_QMASK_SHIFT;
  const uint4 u3 = ((const uint4*)(bn.grec + 4 * (size_t)id))[3];
  return make_uint2(id, u3.y + ((uint32_t)ty - (u3.z >> 16)) * u3.w + ((uint32_t)tx - (u3.z & 0xFFFFu)));
}

// ---- work items ------------------------------------------------------------------------------------------------
struct BlendItem { int tile; uint32_t seg, nseg, s, e, w; uint2 range; bool split, lazy; };
// blockIdx -> (tile, list segment) through the work list of the sort kernel (im.tile_order: segments of split lists
// first, a tile's segments consecutive, then the other tiles in descending order of list length).  The work item and the
// counters are independent loads, the tile's range the only dependent one: two memory round trips before the walk starts.
__device__ __forceinline__ bool blend_item(const HgsImage& im, uint32_t Rcap, BlendItem& it) {
  const uint4 st = *(const uint4*)(im.status + HGS_ST_SORT_ITEMS);   // [4..7]: -, split items, segment length, work items
  const uint32_t item = im.tile_order[blockIdx.x];
  if (blockIdx.x >= st.w || item == HGS_ITEM_NONE) return false;
  it.w = blockIdx.x;
  it.lazy = im.status[HGS_ST_LAZY] != 0u;
  it.split = blockIdx.x < st.y;
  it.tile = (int)HGS_ITEM_TILE(item);
  it.seg = HGS_ITEM_PART(item);
  it.range = im.ranges[it.tile];
  // binning buffer under-sized (flagged by the scatter kernel): the pass is void -- lists were dropped, sorted keys of the others may
  // never have been written, and a record is built THROUGH its key (a stray id would be a stray address): every list reads as empty
  if (it.range.y > Rcap || im.status[HGS_ST_OVERFLOW] != 0u) it.range = make_uint2(0u, 0u);
  const uint32_t n = it.range.y - it.range.x;
  it.nseg = 1u; it.s = 0u; it.e = n;
  if (it.split) {
    const HgsSplit sp = hgs_split_of(n, st.z);
    it.nseg = sp.nseg;
    // (round 5: the list's remainder as its FIRST segment instead of its last -- every later segment waits for all its predecessors'
    // products and the front segments hold the nearest, largest Gaussians -- measured within noise: C2 3831 -> 3865 it/s, the forward
    // 69.5 -> 69.0 us, C4 unchanged, profiles/r05_wg_trace_c2.txt; not kept)
    it.s = it.seg * sp.seglen;
    it.e = min(n, it.s + sp.seglen);
  }
  return true;
}

// Front-to-back walk of list positions [s, e) of a tile (src = the tile's first record).
// The reference's loop (forward.cu:309-362): alpha test, stop rule, colour accumulation, last contributor.
// T < 0 marks a pixel that is done (saturated, forward.cu:346-351, or outside the image): its magnitude stays the
// transmittance it stopped at.  A separate per-lane flag costs a byte register and ~6 vector instructions per
// entry to test, merge and update; the sign costs one compare.
template <int C>
__device__ __forceinline__ void fwd_walk(const HgsBinning& bn, bool lazy, uint32_t first, int tx, int ty, uint32_t s, uint32_t e,
                                         float4 (&recs)[2][REC_BATCH * Chan<C>::REC4], uint32_t (&alive)[2][4], float pxf,
                                         float pyf, int wave, int lane, float& T, float (&acc)[C], uint32_t& last) {
  constexpr int REC4 = Chan<C>::REC4;
  const uint32_t L = e - s;
  const int nb = (int)((L + REC_BATCH - 1) / REC_BATCH);
  const uint32_t base = first + s;                       // list position of the walk's first entry
  const uint32_t sq = threadIdx.x / REC4;                // this thread stages quarter sr of entry sq of every batch
  const int sr = (int)(threadIdx.x - sq * REC4);
  const uint32_t nf4 = L * REC4;
  float4 stage = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();   // (the staging buffers may still be read by a previous walk)
  if (nb > 0) {
    if (threadIdx.x < REC_BATCH * REC4 && threadIdx.x < nf4) stage = stage_quarter<C>(bn, lazy, base + sq, sr, tx, ty);
    if (threadIdx.x < REC_BATCH * REC4) recs[0][threadIdx.x] = stage;
    if (threadIdx.x < 4) alive[0][threadIdx.x] = 1u;
  }
  for (int b = 0; b < nb; b++) {
    const int cur = b & 1;
    if (b + 1 < nb) {
      const uint32_t i = (uint32_t)(b + 1) * (REC_BATCH * REC4) + threadIdx.x;
      if (threadIdx.x < REC_BATCH * REC4 && i < nf4) stage = stage_quarter<C>(bn, lazy, base + (uint32_t)(b + 1) * REC_BATCH + sq, sr, tx, ty);
    }
    __syncthreads();
    // forward.cu:309-311: the tile stops when every pixel is saturated (flags written before the barrier above)
    if ((alive[cur][0] | alive[cur][1] | alive[cur][2] | alive[cur][3]) == 0u) break;
    const int cnt = min(REC_BATCH, (int)L - b * REC_BATCH);
    const float* rf = (const float*)&recs[cur][0];
    const uint32_t mk = lane < cnt ? __float_as_uint(rf[lane * 4 * REC4 + 7 + C]) : 0u;
    uint64_t m = __ballot(((mk >> wave) & 1u) != 0u);
    if (__ballot(T > 0.f) == 0) m = 0;
    // One entry, branch-free: the counters of round 3 show the scalar unit as busy as the vector pipe in this loop (195 scalar
    // against 277 vector instructions per wavefront, together 0.95 of the issue slots) -- exec-mask set-up around the
    // "blends here" lanes, a wave-wide vote per entry for the early return, 64-bit m & (m - 1).  Lanes that do not blend the
    // entry add w = 0 (same bits: x + 0 * c, finite c), `last` and T are selects, and the only votes left are the rare ones
    // behind a saturating lane.
    auto process = [&](const Rec<C>& r, int ent) {
      const float4 r0 = r.q[0], r1 = r.q[1];
      const float dx = r0.x - pxf, dy = r0.y - pyf;
      const float power = -0.5f * (r0.z * dx * dx + r1.x * dy * dy) - r0.w * dx * dy;  // forward.cu:335
      const float alpha = fminf(0.99f, r1.y * __expf(power));                           // :343
      const bool ok = T > 0.f && power <= 0.f && alpha >= (1.0f / 255.0f);              // :336, :344
      const float test_T = T * (1.f - alpha);
      const bool sat = ok && test_T < 0.0001f;                                          // :346-351
      const bool add = ok && !sat;
      const float w = add ? alpha * T : 0.f;
      const float* f = (const float*)&r.q[0];                                           // features start at float 6
#pragma unroll
      for (int k = 0; k < C; k++) acc[k] = __builtin_fmaf(f[6 + k], w, acc[k]);         // :354-355
      last = add ? s + (uint32_t)(b * REC_BATCH + ent + 1) : last;                      // :328, :361
      T = ok ? (sat ? -T : test_T) : T;
    };
    if (m) {
      // two record buffers used alternately (LDS reads of the next entry in flight, no register rotation); the read behind the
      // last entry is issued all the same, from slot 63 of the batch (inside the array, never used): no branch around it.
      // A wavefront whose pixels are all saturated leaves the batch (forward.cu:309-311 per wavefront): T > 0 nowhere only
      // ever follows a saturating entry, so the vote on T alone decides, folded into the bit set the loop runs on.
      int ea = __builtin_ctzll(m), eb = 0;
      Rec<C> ra = lds_record<C>(recs[cur], ea), rb;
      while (true) {
        m &= ~(1ull << ea);
        eb = __builtin_ctzll(m | (1ull << 63));
        rb = lds_record<C>(recs[cur], eb);
        process(ra, ea);
        if (__ballot(T > 0.f) == 0ull) m = 0ull;
        if (m == 0ull) break;
        m &= ~(1ull << eb);
        ea = __builtin_ctzll(m | (1ull << 63));
        ra = lds_record<C>(recs[cur], ea);
        process(rb, eb);
        if (__ballot(T > 0.f) == 0ull) m = 0ull;
        if (m == 0ull) break;
      }
    }
    if (b + 1 < nb) {
      if (threadIdx.x < REC_BATCH * REC4) recs[cur ^ 1][threadIdx.x] = stage;
      const uint32_t wave_alive = __ballot(T > 0.f) != 0 ? 1u : 0u;   // all lanes vote: taken outside the lane-0 branch
      if (lane == 0) alive[cur ^ 1][wave] = wave_alive;
    }
  }
}

// ------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(HGS_BLOCK) __attribute__((amdgpu_waves_per_eu(8))) void blend_fwd_kernel(HgsImage im, HgsBinning bn, int W, int H, int gx, uint32_t Rcap,
                                                              const float* __restrict__ bg, float* __restrict__ out_color) {
  constexpr int REC4 = Chan<C>::REC4;
  __shared__ float4 recs[2][REC_BATCH * REC4];
  __shared__ uint32_t alive[2][4];
  __shared__ uint32_t s_flag;
  BlendItem it;
  if (!blend_item(im, Rcap, it)) return;
  const int tile = it.tile;
  WgTrace _trace(g_wg_trace_fwd);
  _trace.item(tile, it.seg, it.nseg, it.e - it.s);
  const int tx = tile % gx, ty = tile / gx;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int px = tx * HGS_TILE + (wave & 1) * 8 + (lane & 7);
  const int py = ty * HGS_TILE + (wave >> 1) * 8 + (lane >> 3);
  const bool inside = px < W && py < H;
  const float pxf = (float)px, pyf = (float)py;
  const uint32_t list0 = it.range.x;                    // the tile's first list position
  float acc[C];
#pragma unroll
  for (int k = 0; k < C; k++) acc[k] = 0.f;
  uint32_t last = 0;
  const size_t pix = (size_t)py * W + px, HW = (size_t)H * W;

  if (!it.split) {
    float T = inside ? 1.f : -1.f;
    fwd_walk<C>(bn, it.lazy, list0, tx, ty, 0u, it.e, recs, alive, pxf, pyf, wave, lane, T, acc, last);
    uint32_t wmax = last;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, d, 64));
    if (lane == 0 && wmax) atomicMax(&im.tile_maxc[tile], wmax);
    if (inside) {
      T = fabsf(T);
      im.final_T[pix] = T;
      im.n_contrib[pix] = last;
#pragma unroll
      for (int k = 0; k < C; k++) out_color[k * HW + pix] = acc[k] + T * bg[k];         // :372
    }
    return;
  }

  // ---- a list of TWO segments is walked serially by its first segment's workgroup (round 4).  The two-pass scheme below costs
  // a segment its pass A, the wait for the products in front, pass B for the pixels that stop inside it and the finalisation:
  // with two segments that chain is LONGER than the serial walk (C3: 17.6 + 3.9 + 10 + 5 us against ~30 for the tile's 377
  // entries; trained state: 43 + 2 + 22 + 4 against ~63 for 599), and a frame's one list just over the split threshold was
  // what its forward launch ended with, alone on the GPU (tools/wg_trace.py).  The BACKWARD still runs the two segments as two
  // workgroups (its walk costs 3.5 times the forward's and needs no communication): what it reads of the forward is left here
  // -- the transmittance in front of the second segment and that segment's colour (nothing lies behind it).
  if (it.nseg == 2u) {
    if (it.seg == 1u) return;
    const uint32_t n = it.range.y - it.range.x;
    float T = inside ? 1.f : -1.f;
    fwd_walk<C>(bn, it.lazy, list0, tx, ty, 0u, it.e, recs, alive, pxf, pyf, wave, lane, T, acc, last);
    const size_t slot1 = ((size_t)it.w + 1) * HGS_BLOCK + threadIdx.x;
    bn.seg_T[slot1] = fabsf(T);
    // (the first segment's colour waits in its own cell of the segment array, which nobody else reads)
#pragma unroll
    for (int k = 0; k < C; k++) { bn.seg_C[((size_t)it.w * C + k) * HGS_BLOCK + threadIdx.x] = acc[k]; acc[k] = 0.f; }
    fwd_walk<C>(bn, it.lazy, list0, tx, ty, it.e, n, recs, alive, pxf, pyf, wave, lane, T, acc, last);
#pragma unroll
    for (int k = 0; k < C; k++) {
      bn.seg_C[(((size_t)it.w + 1) * C + k) * HGS_BLOCK + threadIdx.x] = acc[k];
      acc[k] += bn.seg_C[((size_t)it.w * C + k) * HGS_BLOCK + threadIdx.x];     // colours summed back to front, like the tile finalisation below
      // the SUFFIX sum in the first segment's cell, as the general path's finalisation leaves it (the backward reads only the
      // cell of segment w + 1 today; a later reader of cell w finds what the file header promises).  NOT written for a
      // two-segment tile: seg_P, seg_Tout, seg_last, tile_prog, tile_done -- the exchange between segment workgroups.
      bn.seg_C[((size_t)it.w * C + k) * HGS_BLOCK + threadIdx.x] = acc[k];
    }
    uint32_t wmax = last;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, d, 64));
    if (lane == 0 && wmax) atomicMax(&im.tile_maxc[tile], wmax);
    if (inside) {
      T = fabsf(T);
      im.final_T[pix] = T;
      im.n_contrib[pix] = last;
#pragma unroll
      for (int k = 0; k < C; k++) out_color[k * HW + pix] = acc[k] + T * bg[k];
    }
    return;
  }

  // ---- one segment of a split list (see the file header)
  const size_t slot = (size_t)it.w * HGS_BLOCK + threadIdx.x;   // this pixel's cell in the per-segment arrays
  const size_t slot0 = slot - (size_t)it.seg * HGS_BLOCK;       // the same pixel's cell of the tile's first segment
  // pass A: the reference's walk over this segment with the transmittance in front of it taken as 1 (no dependence on the
  // segments before it): local product, local colours, last contributing position
  float T = inside ? 1.f : -1.f;
  fwd_walk<C>(bn, it.lazy, list0, tx, ty, it.s, it.e, recs, alive, pxf, pyf, wave, lane, T, acc, last);
  // A pixel that stopped on the local product stops in this segment or before it for any transmittance <= 1 in front:
  // what the later segments read as this segment's product only has to be below the stop threshold.
  const float P = inside ? fmaxf(T, 0.f) : -1.f;
  _trace.mark(1);
  hgs_st_agent(&bn.seg_P[slot], P);
  hgs_drain_stores();
  __syncthreads();
  if (threadIdx.x == 0) {
    hgs_publish_part(&im.tile_prog[tile], it.seg);
    s_flag = it.seg == 0 || hgs_wait_parts(&im.tile_prog[tile], (1ull << it.seg) - 1ull, im.status) ? 1u : 0u;
  }
  __syncthreads();
  _trace.mark(5);                 // (the predecessors' products are published)
  float T_in = inside ? 1.f : -1.f;
  if (s_flag) {
    constexpr uint32_t TG = 8;
    // (several loads in flight at a time: one L2 round trip per predecessor was most of a late segment's wait)
    for (uint32_t m0 = 0; m0 < it.seg; m0 += TG) {
      float v[TG];
#pragma unroll
      for (uint32_t j = 0; j < TG; j++) v[j] = m0 + j < it.seg ? hgs_ld_agent(&bn.seg_P[slot0 + (size_t)(m0 + j) * HGS_BLOCK]) : 1.f;
#pragma unroll
      for (uint32_t j = 0; j < TG; j++) T_in *= v[j];                 // list order (a factor of 1 is exact)
    }
    if (!inside) T_in = -1.f;
  }
  _trace.mark(2);
  bn.seg_T[slot] = fabsf(T_in);  // transmittance in front of this segment: the backward of the PREVIOUS segment starts from it
  // Per pixel, with the true transmittance in front now known:
  //   below the stop threshold  -> done before this segment: any entry that passes the alpha test would stop it;
  //   no local stop and T_in * P clear of the threshold -> the walk would not have stopped here either: its colours are
  //                                the local ones times T_in (blending is linear in the transmittance in front);
  //   otherwise                 -> the pixel stops in this segment (or sits within rounding of the threshold): only
  //                                these pixels are walked again, from T_in, exactly as the serial walk would.
  // (The FIRST segment has nothing in front of it: its pass A started from the true transmittance, 1, and IS the serial walk --
  //  every pixel's local result is final.  Until round 4 its stopping pixels were walked a second time like any other
  //  segment's: on the state training leaves, where opaque strands stop most pixels of a dense tile early, that was a second
  //  full pass over the segment -- 33 us behind a first pass of 43 us in the one split tile of a frame, whose two workgroups
  //  were alone on the GPU for the last quarter of the launch, tools/wg_trace.py.)
  const bool first = it.seg == 0;
  const bool dead = inside && !first && T_in < 0.0001f;
  const bool through = inside && !dead && (first || (T > 0.f && T_in * T >= 0.00010002f));
  const bool redo = inside && !dead && !through;
  auto publish = [&]() {   // this pixel's result for the segment (the tile's finalisation and the backward read it)
#pragma unroll
    for (int k = 0; k < C; k++) hgs_st_agent(&bn.seg_C[((size_t)it.w * C + k) * HGS_BLOCK + threadIdx.x], acc[k]);
    hgs_st_agent(&bn.seg_Tout[slot], T);
    hgs_st_agent(&bn.seg_last[slot], last);
  };
  if (through) {
    T *= T_in;
#pragma unroll
    for (int k = 0; k < C; k++) acc[k] *= T_in;
  } else if (!redo) {      // done before this segment (or outside the image)
    T = -fmaxf(fabsf(T_in), 1e-30f);
    last = 0u;
#pragma unroll
    for (int k = 0; k < C; k++) acc[k] = 0.f;
  }
  if (!redo) publish();
  if (__syncthreads_or(redo ? 1 : 0)) {   // pass B (the registers of pass A are free again: its results are stored)
    T = redo ? T_in : -1.f;
    last = 0u;
#pragma unroll
    for (int k = 0; k < C; k++) acc[k] = 0.f;
    fwd_walk<C>(bn, it.lazy, list0, tx, ty, it.s, it.e, recs, alive, pxf, pyf, wave, lane, T, acc, last);
    if (redo) publish();
  }
  _trace.mark(3);
  hgs_drain_stores();
  __syncthreads();
  if (threadIdx.x == 0)
    s_flag = __hip_atomic_fetch_add(&im.tile_done[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == it.nseg - 1u ? 1u : 0u;
  __syncthreads();
  _trace.mark(4);
  if (!s_flag) return;
  // ---- the tile's last workgroup: combine the segments per pixel, in list order
  const size_t w0 = (size_t)it.w - it.seg;
  // (loads in groups: each group costs one L2 round trip, and a 60-segment tile is finalised by ONE workgroup)
  float Tfin = 1.f;
  uint32_t nc = 0u, lastseg = it.nseg - 1u;
  bool stopped = false;
  for (uint32_t m0 = 0; m0 < it.nseg; m0 += 8) {
    float to[8];
    uint32_t la[8];
#pragma unroll
    for (uint32_t j = 0; j < 8; j++) {
      const bool in = m0 + j < it.nseg;
      to[j] = in ? hgs_ld_agent(&bn.seg_Tout[slot0 + (size_t)(m0 + j) * HGS_BLOCK]) : 0.f;
      la[j] = in ? hgs_ld_agent(&bn.seg_last[slot0 + (size_t)(m0 + j) * HGS_BLOCK]) : 0u;
    }
#pragma unroll
    for (uint32_t j = 0; j < 8; j++)
      if (m0 + j < it.nseg && !stopped) {
        nc = max(nc, la[j]);
        Tfin = fabsf(to[j]);
        if (to[j] < 0.f) { lastseg = m0 + j; stopped = true; }   // stopped here: nothing behind it contributes
      }
    if (__ballot(!stopped) == 0) break;                            // (this wavefront's pixels are all accounted for)
  }
  // Colours: nothing behind a pixel's stop segment is added, and the backward reads the suffix sum of segment m + 1 only
  // for a pixel with contributors behind segment m: the sums start at the wavefront's last stop segment.
  uint32_t wseg = lastseg;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) wseg = max(wseg, (uint32_t)__shfl_xor((int)wseg, d, 64));
  wseg = __builtin_amdgcn_readfirstlane(wseg);
#pragma unroll
  for (int k = 0; k < C; k++) acc[k] = 0.f;
  constexpr int FG = C <= 4 ? 8 : 4;                               // segments per group of the colour sums
  for (int mh = (int)wseg; mh >= 0; mh -= FG) {
    float v[FG][C];
#pragma unroll
    for (int j = 0; j < FG; j++)
#pragma unroll
      for (int k = 0; k < C; k++)
        v[j][k] = mh - j >= 0 ? hgs_ld_agent(&bn.seg_C[((w0 + (size_t)(mh - j)) * C + k) * HGS_BLOCK + threadIdx.x]) : 0.f;
#pragma unroll
    for (int j = 0; j < FG; j++)
      if (mh - j >= 0) {
#pragma unroll
        for (int k = 0; k < C; k++) {
          if ((uint32_t)(mh - j) <= lastseg) acc[k] += v[j][k];
          bn.seg_C[((w0 + (size_t)(mh - j)) * C + k) * HGS_BLOCK + threadIdx.x] = acc[k];   // suffix sum: segment m and everything behind it
        }
      }
  }
  uint32_t wmax = inside ? nc : 0u;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, d, 64));
  if (lane == 0 && wmax) atomicMax(&im.tile_maxc[tile], wmax);
  if (inside) {
    im.final_T[pix] = Tfin;
    im.n_contrib[pix] = nc;
#pragma unroll
    for (int k = 0; k < C; k++) out_color[k * HW + pix] = acc[k] + Tfin * bg[k];
  }
}

// ------------------------------------------------------------------------------------------------
template <int C> struct PixGrad { const float* plane[C]; };  // dL/d(output channel k) as [H,W] planes (need not be adjacent)

// BLACK: the background is black (bg == NULL at the C ABI: what the training step renders on, train.py:94, and what the
// mask / orientation passes always use, loss/losses.py:228,296): the background terms of dL/dalpha (backward_distwar.cu:
// 988-991) vanish and their four instructions per (wavefront, entry) pair are compiled out.
template <int C, bool BLACK>
__global__ __launch_bounds__(HGS_BLOCK) __attribute__((amdgpu_waves_per_eu(HGS_BWD_WAVES))) void blend_bwd_kernel(HgsImage im, HgsBinning bn, int W, int H, int gx,
                                                              uint32_t Rcap, const float* __restrict__ bg,
                                                              PixGrad<C> dL_dpix, float* __restrict__ inst_grad, int tag_rows) {
  constexpr int REC4 = Chan<C>::REC4, NPART = Chan<C>::NPART, NREG = Chan<C>::NREG, NV = 4 * NREG, ROW = Chan<C>::ROW;
  __shared__ float part[4][BWD_BATCH][NV];
  __shared__ float4 recs[2][BWD_BATCH * REC4];
  // a forward that overflowed its binning capacity is void (records and slots of the dropped lists were never written):
  // nothing is read or written here, preprocess_bwd_kernel returns zero gradients
  const uint32_t void_pass = im.status[HGS_ST_OVERFLOW];
  BlendItem it;
  if (!blend_item(im, Rcap, it) || void_pass) return;   // same work list as the forward: (tile, list segment)
  const int tile = it.tile;
  WgTrace _trace(g_wg_trace_bwd);
  _trace.item(tile, it.seg, it.nseg, it.e - it.s);
  const uint2 range = it.range;
  const uint32_t maxc = im.tile_maxc[tile];
  const int tx = tile % gx, ty = tile / gx;
  {
    // Every instance row of the scratch is written by exactly one workgroup, so the scratch needs no clearing pass:
    // entries past the last one any pixel needed (positions >= maxc; ~2% of the instances) get explicit zero rows here.
    const uint32_t end = range.x + it.e, first = range.x + max(it.s, maxc);
    if (end > first) {
      const uint32_t n4 = (end - first) * (ROW / 4);
      for (uint32_t i = threadIdx.x; i < n4; i += HGS_BLOCK) {
        const uint32_t inst = first + i / (ROW / 4);
        const uint2 is = stage_id_slot<C>(bn, it.lazy, inst, tx, ty);
        const uint32_t slot = is.y;
        // (C = 7: the row's last float names its Gaussian -- what row_reduce_kernel finds the segments of the scratch by)
        const float tag = (C > 3 && tag_rows && i % (ROW / 4) == ROW / 4 - 1) ? __uint_as_float(is.x) : 0.f;
        if (slot < Rcap) ((float4*)(inst_grad + (size_t)slot * ROW))[i % (ROW / 4)] = make_float4(0.f, 0.f, 0.f, tag);
      }
    }
  }
  if (maxc <= it.s) return;  // nothing in this part of the list contributed to any pixel
  const int seg_lo = (int)it.s, top = (int)min(maxc, it.e);   // list positions [seg_lo, top) are walked, back to front
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int px = tx * HGS_TILE + (wave & 1) * 8 + (lane & 7);
  const int py = ty * HGS_TILE + (wave >> 1) * 8 + (lane >> 3);
  const bool inside = px < W && py < H;
  const float pxf = (float)px, pyf = (float)py;
  const size_t pix = (size_t)py * W + px;

  // first batch (the END of the list: the walk is back to front) is in flight while the per-pixel state is set up
  const uint32_t sq = threadIdx.x / REC4;                // this thread stages quarter sr of entry sq of every batch
  const int sr = (int)(threadIdx.x - sq * REC4);
  float4 stage = make_float4(0.f, 0.f, 0.f, 0.f);
  {
    const int lo = max(seg_lo, top - BWD_BATCH), cnt = top - lo;
    if (threadIdx.x < cnt * REC4) stage = stage_quarter<C>(bn, it.lazy, range.x + (uint32_t)lo + sq, sr, tx, ty);
  }

  const float T_final = inside ? im.final_T[pix] : 0.f;
  float T = T_final;
  const uint32_t last = inside ? im.n_contrib[pix] : 0u;
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
    if (!BLACK) {
      bg_dot += bg[k] * dpx[k];
      if (k < 3) bg_dot_rgb += bg[k] * dpx[k];
    }
  }
  if (it.split && last > it.e) {
    // This pixel has contributors behind this segment.  The serial walk would arrive here with T = the transmittance in
    // front of the next segment and accum_rec = (colour blended behind) / T: both were left by the forward (seg_T of the
    // next segment; seg_C holds, after the tile's finalisation, the colour of a segment and everything behind it).
    const float Tn = bn.seg_T[((size_t)it.w + 1) * HGS_BLOCK + threadIdx.x];
    float d = 0.f, d_rgb = 0.f;
#pragma unroll
    for (int k = 0; k < C; k++) {
      d = __builtin_fmaf(bn.seg_C[(((size_t)it.w + 1) * C + k) * HGS_BLOCK + threadIdx.x], dpx[k], d);
      if (k == 2) d_rgb = d;
    }
    const float inv = Tn > 0.f ? 1.f / Tn : 0.f;
    T = Tn;
    acc_dot = d * inv;
    acc_dot_rgb = d_rgb * inv;
  }

  for (int i = threadIdx.x; i < 4 * BWD_BATCH * NV; i += HGS_BLOCK) (&part[0][0][0])[i] = 0.f;
  if (threadIdx.x < BWD_BATCH * REC4) recs[0][threadIdx.x] = stage;
  __syncthreads();

  // walk the list back to front in batches of BWD_BATCH positions; position p (0-based) is valid for a pixel
  // iff p < n_contrib (backward_distwar.cu:943-945)
  int cur = 0;
  for (int hi = top; hi > seg_lo; hi -= BWD_BATCH, cur ^= 1) {
    const int lo = max(seg_lo, hi - BWD_BATCH);
    const int cnt = hi - lo;
    if (lo > seg_lo) {  // next batch's records: in flight during this batch's math
      const int nlo = max(seg_lo, lo - BWD_BATCH), ncnt = lo - nlo;
      if (threadIdx.x < ncnt * REC4) stage = stage_quarter<C>(bn, it.lazy, range.x + (uint32_t)nlo + sq, sr, tx, ty);
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
      _trace.count(1, 1u);
      if (__ballot(ok) == 0) { _trace.count(2, 1u); return; }
      _trace.count(3, (unsigned)__popcll(__ballot(ok)));
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
      const float bgw = BLACK ? 0.f : -T_final * inv_one_m_a;
      if (!BLACK) dL_dalpha += bgw * bg_dot;                                       // :991
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
        const float u_rgb = BLACK ? G * (dL_dalpha_rgb * T) : G * (dL_dalpha_rgb * T + bgw * bg_dot_rgb);
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
    while (m) {
      const int ea = 63 - __builtin_clzll(m);
      m &= ~(1ull << ea);
      process(lds_record<C>(recs[cur], ea), ea);
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
    // (C = 7, when row_reduce_kernel will sum the rows: the row's sixteenth float names its Gaussian -- the record's own word --,
    // which is how that kernel finds the segments of the scratch; the sums ignore it)
    if (C > 3 && tag_rows) {
      for (int e = threadIdx.x; e < cnt; e += HGS_BLOCK) {
        const uint32_t slot = __float_as_uint(rf[e * 4 * REC4 + 8 + C]);
        if (slot < Rcap) inst_grad[(size_t)slot * ROW + ROW - 1] = rf[e * 4 * REC4 + 6 + C];
      }
    }
    if (lo > seg_lo && threadIdx.x < BWD_BATCH * REC4) recs[cur ^ 1][threadIdx.x] = stage;
    __syncthreads();
  }
}

}  // namespace

extern "C" int hgs_debug_set_wg_trace(void* device_buf_fwd, void* device_buf_bwd) {
  HGS_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_wg_trace_fwd), &device_buf_fwd, sizeof(void*)));
  HGS_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_wg_trace_bwd), &device_buf_bwd, sizeof(void*)));
  return 0;
}

// one workgroup per unsplit tile + one per segment of a split list (at most what the work list and the segment arrays hold;
// idle workgroups leave at once)
static inline unsigned blend_grid(int T, const HgsBinning& b) {
  const unsigned cap = (unsigned)HGS_SPLIT_CAPACITY(T);
  return (unsigned)T + (b.seg_cap < cap ? b.seg_cap : cap);
}

int hgs_launch_blend_fwd(hipStream_t s, int W, int H, int Rcap, int channels, const float* bg, const HgsImage& im,
                         const HgsBinning& b, float* out_color) {
  const int gx = (W + HGS_TILE - 1) / HGS_TILE, gy = (H + HGS_TILE - 1) / HGS_TILE;
  {
    HgsProfScope _prof(s, HGS_K_BLEND_FWD);
    if (channels == 3)
      hipLaunchKernelGGL(blend_fwd_kernel<3>, dim3(blend_grid(gx * gy, b)), dim3(HGS_BLOCK), 0, s, im, b, W, H, gx,
                         (uint32_t)Rcap, bg, out_color);
    else
      hipLaunchKernelGGL(blend_fwd_kernel<7>, dim3(blend_grid(gx * gy, b)), dim3(HGS_BLOCK), 0, s, im, b, W, H, gx,
                         (uint32_t)Rcap, bg, out_color);
  }
  HGS_CHECK_LAUNCH();
  return 0;
}

int hgs_launch_blend_bwd(hipStream_t s, int W, int H, int Rcap, int channels, const float* bg, const HgsImage& im,
                         const HgsBinning& b, const float* const* dL_dpix_planes, float* inst_grad, int tag_rows) {
  const int gx = (W + HGS_TILE - 1) / HGS_TILE, gy = (H + HGS_TILE - 1) / HGS_TILE;
  {
    HgsProfScope _prof(s, HGS_K_BLEND_BWD);
    if (channels == 3) {
      PixGrad<3> pg;
      for (int k = 0; k < 3; k++) pg.plane[k] = dL_dpix_planes[k];
      if (bg) hipLaunchKernelGGL((blend_bwd_kernel<3, false>), dim3(blend_grid(gx * gy, b)), dim3(HGS_BLOCK), 0, s, im, b, W, H, gx,
                                 (uint32_t)Rcap, bg, pg, inst_grad, tag_rows);
      else hipLaunchKernelGGL((blend_bwd_kernel<3, true>), dim3(blend_grid(gx * gy, b)), dim3(HGS_BLOCK), 0, s, im, b, W, H, gx,
                              (uint32_t)Rcap, bg, pg, inst_grad, tag_rows);
    } else {
      PixGrad<7> pg;
      for (int k = 0; k < 7; k++) pg.plane[k] = dL_dpix_planes[k];
      if (bg) hipLaunchKernelGGL((blend_bwd_kernel<7, false>), dim3(blend_grid(gx * gy, b)), dim3(HGS_BLOCK), 0, s, im, b, W, H, gx,
                                 (uint32_t)Rcap, bg, pg, inst_grad, tag_rows);
      else hipLaunchKernelGGL((blend_bwd_kernel<7, true>), dim3(blend_grid(gx * gy, b)), dim3(HGS_BLOCK), 0, s, im, b, W, H, gx,
                              (uint32_t)Rcap, bg, pg, inst_grad, tag_rows);
    }
  }
  HGS_CHECK_LAUNCH();
  return 0;
}
