// hgs_binning.hip -- tile binning without a global sort.
//
// The reference sorts R (tile<<32|depth, id) pairs with one 44/45-bit stable CUB radix sort
// (cuda_rasterizer/rasterizer_impl.cu:300-308) after emitting them in Gaussian order (:70-111), then finds
// tile boundaries (:116-138).  Because a Gaussian appears at most once per tile, that stable order is the
// total order (tile, depth_bits, gaussian_id).  Here:
//   scan_kernel        exclusive prefix over the per-tile counts -> `ranges` directly (no boundary search),
//                      and over the per-block instance sums -> per-Gaussian offsets (replaces DeviceScan :277)
//   sort_tiles_kernel  one workgroup per tile sorts its segment by the unique 64-bit key
//                      depth_bits<<32|id in LDS (bitonic), emits point_list in final order, the packed
//                      per-instance records the blend kernels stream (with the instance's Gaussian-major slot,
//                      where the backward stores its row for the deterministic per-Gaussian sum).
//                      Lists longer than one LDS chunk (HGS_SORT_CAP keys) are sorted by ONE WORKGROUP PER CHUNK
//                      of the same launch: every chunk is sorted in LDS and published (agent-scope stores + the
//                      tile's progress mask, hgs_common.h), then each workgroup ranks its own keys against the
//                      other chunks (keys are unique: final position = sum of lower bounds) and emits them.
//                      The same launch builds the blend kernels' work list: segments of split lists first, then
//                      the other tiles by descending list length.
// Result: identical point_list/ranges, ~10x less sort traffic than 144 B/instance, no stability needed.
#include "hgs_common.h"

namespace {

#define SCAN_THREADS 1024
#define ORD_BUCKETS HGS_WL_BUCKETS    // list-length buckets of the tile order (<= SCAN_THREADS)
#define SORT_CAP HGS_SORT_CAP  // keys per chunk

// generic helper: scans `n` uint32 values with one 1024-thread block; calls emit(i, exclusive, value)
template <typename F>
__device__ __forceinline__ uint32_t block_scan(const uint32_t* in, int n, uint32_t* wtot, F emit, uint32_t slot_mask = 0u) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t carry = 0;
  for (int base = 0; base < n; base += SCAN_THREADS) {
    const int i = base + tid;
    const uint32_t v = i < n ? in[slot_mask ? HGS_TILE_SLOT(i, slot_mask) : (uint32_t)i] : 0u;   // (slot_mask: the scattered tile counters)
    const uint32_t incl = hgs_wave_incl_scan(v, lane);
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    uint32_t woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; w++) {
      const uint32_t t = wtot[w];
      if (w < wave) woff += t;
      total += t;
    }
    if (i < n) emit(i, carry + woff + incl - v, v);
    carry += total;
    __syncthreads();
  }
  return carry;
}

// Register-blocked variant for n <= SCAN_THREADS * SCAN_IPT: thread t owns the IPT consecutive values starting at t*IPT,
// all of its loads are issued before anything is consumed (ONE memory round trip instead of one per 1024-value chunk,
// which is what the loop above pays and what bounded this single-block kernel), then one block-wide scan of the totals.
#define SCAN_IPT 16
struct ScanRegs { uint32_t v[SCAN_IPT]; int ipt; };
__device__ __forceinline__ void scan_load(const uint32_t* in, int n, ScanRegs& r, uint32_t slot_mask = 0u) {
  r.ipt = (n + SCAN_THREADS - 1) / SCAN_THREADS;
  const int i0 = threadIdx.x * r.ipt;
#pragma unroll
  for (int k = 0; k < SCAN_IPT; k++)
    r.v[k] = (k < r.ipt && i0 + k < n) ? in[slot_mask ? HGS_TILE_SLOT(i0 + k, slot_mask) : (uint32_t)(i0 + k)] : 0u;
}
template <typename F>
__device__ __forceinline__ uint32_t scan_regs(ScanRegs& r, int n, uint32_t* wtot, F emit) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t mine = 0;
#pragma unroll
  for (int k = 0; k < SCAN_IPT; k++) mine += r.v[k];
  const uint32_t incl = hgs_wave_incl_scan(mine, lane);
  __syncthreads();
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  uint32_t woff = 0, total = 0;
#pragma unroll
  for (int w = 0; w < SCAN_THREADS / 64; w++) {
    const uint32_t t = wtot[w];
    if (w < wave) woff += t;
    total += t;
  }
  uint32_t run = woff + incl - mine;
  const int i0 = tid * r.ipt;
#pragma unroll
  for (int k = 0; k < SCAN_IPT; k++)
    if (k < r.ipt && i0 + k < n) { emit(i0 + k, run, r.v[k]); run += r.v[k]; }
  return total;
}

// (round 6) HGS_COUNT_ROW_RUNS, blocking mode / frames beyond the scatter kernel's scan: the +1 / -1 marks the preprocess launch
// left per tile row of its large rectangles become counts -- the running sum over the tiles in row-major order is the number of
// such rectangles covering each tile -- which are added to the scattered counter table; the marks are left at zero for the next
// pass.  (In capacity mode the scan workgroups of scatter_kernel do this for their own tiles.)
#define TD_IPT 8
__device__ __forceinline__ void fold_tile_delta(int T, const HgsImage& im, int* wsum, int* carry_s) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int carry = 0;
  for (int base = 0; base < T; base += SCAN_THREADS * TD_IPT) {
    const int t0 = base + (int)threadIdx.x * TD_IPT;
    int v[TD_IPT], mine = 0;
    if (t0 + TD_IPT <= T + 1) {      // (the array holds T + 1 entries, an even number of words behind a 256-byte boundary)
      const int4 a0 = *(const int4*)(im.tile_delta + t0), a1 = *(const int4*)(im.tile_delta + t0 + 4);
      v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
    } else {
#pragma unroll
      for (int i = 0; i < TD_IPT; i++) v[i] = t0 + i <= T ? im.tile_delta[t0 + i] : 0;
    }
#pragma unroll
    for (int i = 0; i < TD_IPT; i++) mine += v[i];
    const int incl = (int)hgs_wave_incl_scan((uint32_t)mine, lane);     // (two's complement: sums of signed marks)
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int run = carry + incl - mine;
    for (int w = 0; w < wave; w++) run += wsum[w];
    if (threadIdx.x == SCAN_THREADS - 1) *carry_s = run + mine;
#pragma unroll
    for (int i = 0; i < TD_IPT; i++) {
      run += v[i];
      if (t0 + i < T && run != 0) atomicAdd(&im.tile_count[HGS_TILE_SLOT((uint32_t)(t0 + i), im.tile_mask)], (uint32_t)run);
      if (v[i] != 0) im.tile_delta[t0 + i] = 0;
    }
    __syncthreads();
    carry = *carry_s;
  }
  __threadfence();      // the scan below reads the counters through other threads
  __syncthreads();
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_kernel(int nblk, int T, HgsGeom g, HgsImage im,
                                                            unsigned int* __restrict__ max_rendered, int row_runs) {
  __shared__ uint32_t wtot[SCAN_THREADS / 64];
  if (row_runs) {
    __shared__ int td_carry;
    fold_tile_delta(T, im, (int*)wtot, &td_carry);
  }
  uint32_t* bs = g.block_sums;
  uint2* ranges = im.ranges;
  auto emit_bs = [&](int i, uint32_t excl, uint32_t) { bs[i] = excl; };
  auto emit_rg = [&](int i, uint32_t excl, uint32_t v) {
    ranges[i] = v ? make_uint2(excl, excl + v) : make_uint2(0u, 0u);  // empty tiles stay (0,0): memset at :310
    hgs_emit_sort_items((uint32_t)i, v, (uint32_t)T, im);              // long lists: one sort workgroup per chunk
  };
  uint32_t R;
  if (nblk <= SCAN_THREADS * SCAN_IPT && T <= SCAN_THREADS * SCAN_IPT) {
    ScanRegs rb, rt;
    scan_load(bs, nblk, rb);
    scan_load(im.tile_count, T, rt, im.tile_mask);
    scan_regs(rb, nblk, wtot, emit_bs);
    R = scan_regs(rt, T, wtot, emit_rg);
  } else {
    block_scan(bs, nblk, wtot, emit_bs);
    R = block_scan(im.tile_count, T, wtot, emit_rg, im.tile_mask);
  }
  if (threadIdx.x == 0) {
    im.status[HGS_ST_R] = R;
    if (max_rendered) atomicMax(max_rendered, R);   // sticky maximum for graph replays (hgs.h)
  }
}

// One instance in its final list position: point_list, the sorted key and the packed record the blend kernels stream.
// Everything about the Gaussian comes from its 64-byte template (HgsGeom::grec, scatter_kernel): one contiguous gather.
template <bool EXTRA>   // EXTRA: 64-B records with the 4 extra channels of the single-pass mode, else 48-B records
__device__ __forceinline__ void emit_instance(uint64_t key, uint32_t pos, uint32_t start, int tx, int ty, const HgsGeom& g, const HgsBinning& b) {
  const uint32_t id = (uint32_t)key >> HGS_QMASK_SHIFT;
  b.point_list[pos] = id;
  b.keys_sorted[pos] = key;
  (void)start;
  if (b.lazy) return;                // the blend kernels build the record themselves (stage_quarter, hgs_blend.hip)
  const float4* t = g.grec + 4 * (size_t)id;
  const float4 t0 = t[0], t1 = t[1], t2 = t[2];
  const uint4 u3 = ((const uint4*)t)[3];
  const uint32_t qmask = (uint32_t)key & HGS_QMASK_BITS;   // quadrant mask, computed by the scatter kernel (hgs_quadrant_mask)
  // the instance's slot in Gaussian-major order (offset of the Gaussian + cell of its tile rectangle): the backward stores
  // this instance's row of partial sums there, so that a Gaussian's rows are contiguous for preprocess_bwd_kernel
  const uint32_t slot = u3.y + ((uint32_t)ty - (u3.z >> 16)) * u3.w + ((uint32_t)tx - (u3.z & 0xFFFFu));
  if (!EXTRA) {  // 48-B record: xy, conic, opacity, rgb, id, quadrant mask, slot
    float4* rec = b.packed + (size_t)pos * 3;
    rec[0] = t0;
    rec[1] = t1;
    ((uint4*)rec)[2] = make_uint4(__float_as_uint(t2.x), id, qmask, slot);
  } else {               // 64-B record: ... rgb, 4 extra channels, id, quadrant mask, slot
    float4* rec = b.packed + (size_t)pos * 4;
    rec[0] = t0;
    rec[1] = t1;
    rec[2] = t2;
    ((uint4*)rec)[3] = make_uint4(u3.x, id, qmask, slot);
  }
}

__device__ __forceinline__ void bitonic_lds(uint64_t* sk, int m) {
  for (int k = 2; k <= m; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < (m >> 1); t += HGS_BLOCK) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int l = i | j;
        const bool asc = (i & k) == 0;
        const uint64_t x = sk[i], y = sk[l];
        if ((x > y) == asc) { sk[i] = y; sk[l] = x; }
      }
      __syncthreads();
    }
  }
}

// Lists of at most 128 entries (most tiles of a hair frame) are sorted by ONE wavefront: 64 compare-exchange pairs per
// stage fit its lanes, LDS operations of a wavefront execute in order, so no workgroup barrier separates the stages --
// and the three other wavefronts of the workgroup leave at once, which frees their slots for the next tiles.
#define WAVE_SORT_MAX 128
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void bitonic_wave(uint64_t* sk, int m, int lane) {
  for (int k = 2; k <= m; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (lane < (m >> 1)) {
        const int i = ((lane & ~(j - 1)) << 1) | (lane & (j - 1));
        const int l = i | j;
        const bool asc = (i & k) == 0;
        const uint64_t x = sk[i], y = sk[l];
        if ((x > y) == asc) { sk[i] = y; sk[l] = x; }
      }
      wave_lds_fence();
    }
  }
}

// Work list of the blend kernels, computed by WL_BUILDERS extra workgroups of the sort kernel, i.e. in the shadow of the
// sorts.  im.tile_order[w] = tile | segment << 24 for workgroup w:
//   * Lists longer than 1.5 segment lengths (hgs_split_of) are SPLIT: one work item per segment, a tile's segments
//     consecutive and ascending (the forward waits on predecessors only), the tile flagged in tile_prog.  These items come
//     first: they are the longest pieces of work (and their position is the index of the segment's state in HgsBinning).
//     All long lists are split or -- if their segments do not fit the work list / the segment arrays -- none is.
//   * The other tiles follow in descending order of list length (counting sort by min(length, ORD_BUCKETS-1)).  A list
//     is consumed sequentially, so the blend kernels end when the longest pieces end: measured on the strand workload,
//     raster order started the 58-us tiles of the backward 15-30 us into the launch.
// A builder workgroup runs alone on its SIMDs, so every pass over the T tiles is a chain of latencies (measured at 1080p
// with one builder: 4 us of loads + 5.5 us per further pass = 21 us, and the sort kernel lasts as long as its slowest
// workgroup).  Hence: ONE counting pass that every builder runs over all tiles -- load a length, note the bucket in LDS,
// count it with LDS atomics that return nothing; no global store in the loop, so the loads of several iterations are in
// flight together; the rare long lists only note their tile in a candidate list --, and a placing pass that each builder
// runs over ITS share of the tiles only (the tiles of the shares before it were counted separately in the first pass).
#define WL_BUILDERS HGS_WL_BUILDERS
// LDS of a work-list builder (one set for both forms below: static LDS is what bounds the kernel's workgroups per CU)
struct WlLds { uint32_t hist[ORD_BUCKETS], before[ORD_BUCKETS], aux[ORD_BUCKETS], wsum[HGS_BLOCK / 64], small[4]; };

// The same list by builders that SHARE the counting (frames whose share of tiles fits the LDS array, i.e. all but 8K): the
// one pass every builder ran over all T tiles -- 32 dependent-batch loads per thread at 1080p, 6 of the builders' 12 us, and
// the builders are what the sort kernel waits for (14.2 us with them, 10.9 without, measured) -- becomes a pass over its
// own eighth; the builders then tell each other their histograms through im.wl_exchange (agent-scope stores, a ticket in
// the status words, agent-scope loads: eight workgroups polling one word is nothing) and derive the same positions.
//   wl_exchange: [builder][2][bucket]  counts of the builder's share: lists kept whole | long lists (candidates to split)
//                [HGS_WL_MAX_CAND]     tiles of the candidates, appended through status[HGS_ST_WL_NCAND]
__device__ __forceinline__ void work_list_shared(int T, uint32_t Rcap, HgsSegPolicy pol, const HgsImage& im, const HgsBinning& b, uint16_t* bk, WlLds& L) {
  constexpr int CAND_FLAG = 0x8000;
  uint32_t* hist = L.hist; uint32_t* before = L.before; uint32_t* hist_c = L.aux; uint32_t* wsum = L.wsum;
  uint32_t& nsplit_base = L.small[0]; uint32_t& s_ok = L.small[1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, me = (int)blockIdx.x;
  const int share = ((T + WL_BUILDERS - 1) / WL_BUILDERS + HGS_BLOCK - 1) / HGS_BLOCK * HGS_BLOCK;
  const int my0 = min(T, me * share), my1 = min(T, my0 + share);
  for (int i = tid; i < ORD_BUCKETS; i += HGS_BLOCK) { hist[i] = 0u; hist_c[i] = 0u; }
  if (tid == 0) nsplit_base = 0u;
  const uint32_t S = hgs_segment_length(im.status[HGS_ST_R], pol);
  const uint32_t seg_cap = min(b.seg_cap, (uint32_t)HGS_SPLIT_CAPACITY(T));   // (what the work list holds)
  const uint32_t thr = seg_cap ? S * HGS_SPLIT_QUARTERS / 4u : 0xFFFFFFFFu;
  __syncthreads();
  auto length_of = [&](int i) { const uint2 r = im.ranges[i]; return r.y > Rcap ? 0u : r.y - r.x; };   // (beyond the capacity: void)
  auto bucket_for = [&](uint32_t n) { return ORD_BUCKETS - 1 - (int)min(n, (uint32_t)(ORD_BUCKETS - 1)); };   // 0 = longest
  // ---- this builder's share: buckets, counts, candidates
#pragma unroll 4
  for (int i = my0 + tid; i < my1; i += HGS_BLOCK) {
    const uint32_t n = length_of(i);
    const int k = bucket_for(n);
    if (n > thr) {
      bk[i - my0] = (uint16_t)(k | CAND_FLAG);
      atomicAdd(&hist_c[k], 1u);
      const uint32_t c = atomicAdd(&im.status[HGS_ST_WL_NCAND], 1u);
      if (c < HGS_WL_MAX_CAND) hgs_st_agent(&im.wl_exchange[WL_BUILDERS * 2 * ORD_BUCKETS + c], (uint32_t)i);
      atomicAdd(&im.status[HGS_ST_WL_NSEG], hgs_split_of(n, S).nseg);
    } else {
      bk[i - my0] = (uint16_t)k;
      atomicAdd(&hist[k], 1u);
    }
  }
  __syncthreads();
  uint32_t* mine = im.wl_exchange + (size_t)me * 2 * ORD_BUCKETS;
  for (int k = tid; k < ORD_BUCKETS; k += HGS_BLOCK) { hgs_st_agent(&mine[k], hist[k]); hgs_st_agent(&mine[ORD_BUCKETS + k], hist_c[k]); }
  hgs_drain_stores();
  __syncthreads();
  if (tid == 0) {
    atomicAdd(&im.status[HGS_ST_WL_TICKET], 1u);
    uint32_t ok = 0u;
    for (int spin = 0; spin < (1 << 21); spin++) {
      if (hgs_ld_agent(&im.status[HGS_ST_WL_TICKET]) >= (uint32_t)WL_BUILDERS) { ok = 1u; break; }
      __builtin_amdgcn_s_sleep(2);
    }
    if (!ok) {   // (the builders are the launch's first workgroups: resident together; never observed)
      im.status[HGS_ST_TIMEOUT] = 1u;
      const unsigned long long report = (((unsigned long long)im.status[HGS_ST_SCANPTR_HI] << 32) | im.status[HGS_ST_SCANPTR_LO]) & ~1ull;
      if (report) atomicMax((unsigned int*)report, 0xFFFFFFFFu);
    }
    s_ok = ok;
  }
  __syncthreads();
  if (!s_ok) return;
  // ---- everybody's counts: totals per bucket, and what lies before this builder's share
  const uint32_t nc = hgs_ld_agent(&im.status[HGS_ST_WL_NCAND]), nitems = hgs_ld_agent(&im.status[HGS_ST_WL_NSEG]);
  const bool split_all = nc != 0u && nc <= HGS_WL_MAX_CAND && nitems <= seg_cap;
  const uint32_t nsplit = split_all ? nitems : 0u;
  for (int k = tid; k < ORD_BUCKETS; k += HGS_BLOCK) {
    uint32_t v[WL_BUILDERS], vc[WL_BUILDERS];
#pragma unroll
    for (int q = 0; q < WL_BUILDERS; q++) {
      v[q] = hgs_ld_agent(&im.wl_exchange[(size_t)q * 2 * ORD_BUCKETS + k]);
      vc[q] = hgs_ld_agent(&im.wl_exchange[(size_t)q * 2 * ORD_BUCKETS + ORD_BUCKETS + k]);
    }
    uint32_t tot = 0, bef = 0;
#pragma unroll
    for (int q = 0; q < WL_BUILDERS; q++) {
      const uint32_t x = v[q] + (split_all ? 0u : vc[q]);    // long lists that are not split stay whole, in their bucket
      tot += x;
      if (q < me) bef += x;
    }
    hist[k] = tot;
    before[k] = bef;
  }
  __syncthreads();
  // ---- the long lists' segments (builder 0)
  if (split_all && me == 0) {
    for (uint32_t c = tid; c < nc; c += HGS_BLOCK) {
      const int i = (int)hgs_ld_agent(&im.wl_exchange[WL_BUILDERS * 2 * ORD_BUCKETS + c]);
      const HgsSplit sp = hgs_split_of(length_of(i), S);
      const uint32_t base = atomicAdd(&nsplit_base, sp.nseg);
      for (uint32_t k = 0; k < sp.nseg; k++) im.tile_order[base + k] = (uint32_t)i | (k << 24);
      im.tile_prog[i] = HGS_PART_FLAG;
    }
  }
  uint32_t carry = 0;
  for (int base = 0; base < ORD_BUCKETS; base += HGS_BLOCK) {   // exclusive scan of the histogram
    const uint32_t v = hist[base + tid];
    const uint32_t incl = hgs_wave_incl_scan(v, lane);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t woff = 0, total = 0;
    for (int w = 0; w < HGS_BLOCK / 64; w++) { if (w < wave) woff += wsum[w]; total += wsum[w]; }
    before[base + tid] += nsplit + carry + woff + incl - v;   // first position of this builder's tiles in the bucket
    hist[base + tid] = 0u;
    carry += total;
    __syncthreads();
  }
  // ---- placing pass over this builder's share (order inside a bucket is irrelevant; unrolled: the returning LDS atomics
  // of several iterations are in flight together)
#pragma unroll 4
  for (int i = my0 + tid; i < my1; i += HGS_BLOCK) {
    const int code = (int)bk[i - my0];
    if ((code & CAND_FLAG) && split_all) continue;
    const int bkt = code & (CAND_FLAG - 1);
    im.tile_order[before[bkt] + atomicAdd(&hist[bkt], 1u)] = (uint32_t)i;
  }
  if (me == 0 && tid == 0) {
    im.status[HGS_ST_SPLIT_ITEMS] = nsplit;
    im.status[HGS_ST_SEG_LEN] = S;
    im.status[HGS_ST_WORK_ITEMS] = nsplit + (uint32_t)T - (split_all ? nc : 0u);
  }
}

__device__ __forceinline__ void work_list_block(int T, uint32_t Rcap, HgsSegPolicy pol, const HgsImage& im, const HgsBinning& b, uint16_t* bk, int bk_cap, WlLds& L) {
  constexpr int SPLIT_BUCKET = 0xFFFF, CANDIDATE = 0xFFFE, MAX_CAND = ORD_BUCKETS;
  uint32_t* hist = L.hist; uint32_t* before = L.before; uint32_t* wsum = L.wsum;   // (`before` becomes the buckets' first positions)
  uint32_t* cand = L.aux; uint32_t& ncand = L.small[0]; uint32_t& nitems = L.small[1]; uint32_t& nsplit_base = L.small[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // a builder keeps the buckets of ITS share of the tiles (the only ones it places) in LDS: 1024 entries at 1080p, so the
  // kernel's LDS stays at the 4 KB of a sort chunk and 16 of its workgroups fit a CU instead of 7 (with all 8192 buckets
  // in a 16 KB array: the tile workgroups of a hair frame live one wavefront each, LDS was what bounded their number)
  const int share_w = ((T + WL_BUILDERS - 1) / WL_BUILDERS + HGS_BLOCK - 1) / HGS_BLOCK * HGS_BLOCK;
  const bool cached = share_w <= bk_cap;
  if (!cached && blockIdx.x != 0) return;                             // huge frames: one builder, buckets from memory
  const int nbuild = cached ? WL_BUILDERS : 1, me = (int)blockIdx.x;
  const int share = ((T + nbuild - 1) / nbuild + HGS_BLOCK - 1) / HGS_BLOCK * HGS_BLOCK;
  const int my0 = min(T, me * share), my1 = min(T, my0 + share);     // the tiles this builder places
  auto keep = [&](int i, int k) { if (i >= my0 && i < my1) bk[i - my0] = (uint16_t)k; };
  for (int i = tid; i < ORD_BUCKETS; i += HGS_BLOCK) { hist[i] = 0u; before[i] = 0u; }
  if (tid == 0) { ncand = 0u; nitems = 0u; nsplit_base = 0u; }
  const uint32_t S = hgs_segment_length(im.status[HGS_ST_R], pol);
  const uint32_t seg_cap = min(b.seg_cap, (uint32_t)HGS_SPLIT_CAPACITY(T));   // (what the work list holds)
  const uint32_t thr = seg_cap ? S * HGS_SPLIT_QUARTERS / 4u : 0xFFFFFFFFu;
  __syncthreads();
  auto length_of = [&](int i) { const uint2 r = im.ranges[i]; return r.y > Rcap ? 0u : r.y - r.x; };   // (beyond the capacity: void)
  auto bucket_for = [&](uint32_t n) { return ORD_BUCKETS - 1 - (int)min(n, (uint32_t)(ORD_BUCKETS - 1)); };   // 0 = longest
  auto count = [&](int i, int k) { atomicAdd(&hist[k], 1u); if (i < my0) atomicAdd(&before[k], 1u); };
  auto bucket_mem = [&](int i) {   // (uncached frames: the bucket again from memory)
    const uint32_t n = length_of(i);
    return n > thr && nsplit_base != 0xFFFFFFFFu ? SPLIT_BUCKET : bucket_for(n);
  };
  // ---- counting pass over ALL tiles
  if (cached) {
#pragma unroll 8
    for (int i = tid; i < T; i += HGS_BLOCK) {
      const uint32_t n = length_of(i);
      const int k = n > thr ? CANDIDATE : bucket_for(n);
      keep(i, k);
      if (k != CANDIDATE) count(i, k);
      else { const uint32_t c = atomicAdd(&ncand, 1u); if (c < MAX_CAND) cand[c] = (uint32_t)i; }
    }
  } else {
    for (int i = tid; i < T; i += HGS_BLOCK) {
      const uint32_t n = length_of(i);
      if (n > thr) atomicAdd(&ncand, 1u); else count(i, bucket_for(n));
    }
  }
  __syncthreads();
  // ---- long lists: how many segments in all?  (every builder computes the same number)
  const uint32_t nc = ncand;
  const bool listed = cached && nc <= MAX_CAND;
  if (listed)
    for (uint32_t c = tid; c < nc; c += HGS_BLOCK) atomicAdd(&nitems, hgs_split_of(length_of((int)cand[c]), S).nseg);
  else if (nc)
    for (int i = tid; i < T; i += HGS_BLOCK) { const uint32_t n = length_of(i); if (n > thr) atomicAdd(&nitems, hgs_split_of(n, S).nseg); }
  __syncthreads();
  const bool split_all = nc != 0u && nitems <= seg_cap;
  const uint32_t nsplit = split_all ? nitems : 0u;
  if (!split_all && tid == 0) nsplit_base = 0xFFFFFFFFu;              // (bucket_mem: long lists stay whole)
  __syncthreads();
  // a long list's tile: its work items (written by builder 0), or its bucket among the others
  auto place_long = [&](int i) {
    const uint32_t n = length_of(i);
    if (split_all) {
      if (cached) keep(i, SPLIT_BUCKET);
      if (me == 0) {
        const HgsSplit sp = hgs_split_of(n, S);
        const uint32_t base = atomicAdd(&nsplit_base, sp.nseg);
        for (uint32_t k = 0; k < sp.nseg; k++) im.tile_order[base + k] = (uint32_t)i | (k << 24);
        im.tile_prog[i] = HGS_PART_FLAG;
      }
    } else {
      const int k = bucket_for(n);
      if (cached) keep(i, k);
      count(i, k);
    }
  };
  if (listed) {
    for (uint32_t c = tid; c < nc; c += HGS_BLOCK) place_long((int)cand[c]);
  } else if (nc) {
    for (int i = tid; i < T; i += HGS_BLOCK) if (length_of(i) > thr) place_long(i);
  }
  __syncthreads();
  uint32_t carry = 0;
  for (int base = 0; base < ORD_BUCKETS; base += HGS_BLOCK) {   // exclusive scan of the histogram
    const uint32_t v = hist[base + tid];
    const uint32_t incl = hgs_wave_incl_scan(v, lane);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t woff = 0, total = 0;
    for (int w = 0; w < HGS_BLOCK / 64; w++) { if (w < wave) woff += wsum[w]; total += wsum[w]; }
    before[base + tid] += nsplit + carry + woff + incl - v;   // first position of this builder's tiles in the bucket
    hist[base + tid] = 0u;
    carry += total;
    __syncthreads();
  }
  // ---- placing pass over this builder's share (order inside a bucket is irrelevant; unrolled: the returning LDS atomics
  // of several iterations are in flight together)
#pragma unroll 8
  for (int i = my0 + tid; i < my1; i += HGS_BLOCK) {
    const int bkt = cached ? (int)bk[i - my0] : bucket_mem(i);
    if (bkt != SPLIT_BUCKET) im.tile_order[before[bkt] + atomicAdd(&hist[bkt], 1u)] = (uint32_t)i;
  }
  if (me == 0 && tid == 0) {
    im.status[HGS_ST_SPLIT_ITEMS] = nsplit;
    im.status[HGS_ST_SEG_LEN] = S;
    im.status[HGS_ST_WORK_ITEMS] = nsplit + (uint32_t)T - (split_all ? nc : 0u);
  }
}

// ranks of a thread's KPT keys (registers) among the sorted keys staged in sk[0, cn2): lower bounds, searched in lockstep
// so that the LDS reads of a thread's searches are independent
template <int KPT>
__device__ __forceinline__ void add_lower_bounds(const uint64_t* sk, uint32_t cn2, const uint64_t (&key)[KPT], uint32_t (&rank)[KPT]) {
  uint32_t lo[KPT], hi[KPT];
#pragma unroll
  for (int i = 0; i < KPT; i++) { lo[i] = 0u; hi[i] = cn2; }
  for (uint32_t span = cn2; span > 0; span >>= 1) {      // ceil(log2(cn2)) + 1 halvings bound every search
#pragma unroll
    for (int i = 0; i < KPT; i++) {
      if (lo[i] < hi[i]) {
        const uint32_t mid = (lo[i] + hi[i]) >> 1;
        if (sk[mid] < key[i]) lo[i] = mid + 1; else hi[i] = mid;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < KPT; i++) rank[i] += lo[i];
}

// grid: [0, WL_BUILDERS) work list, then one workgroup per tile, then one per chunk work item of a long list
template <bool EXTRA>
__global__ __launch_bounds__(HGS_BLOCK) void sort_tiles_kernel(int gx, int T, uint32_t Rcap, HgsSegPolicy pol, HgsGeom g, HgsImage im,
                                                               HgsBinning b) {
  // (one sort chunk; the work-list builders keep one 16-bit word per tile of their share of the frame in it)
  constexpr int SK_WORDS = SORT_CAP > 256 ? SORT_CAP : 256;
  __shared__ uint64_t sk[SK_WORDS];
  constexpr int KPT = SORT_CAP / HGS_BLOCK;
  if (blockIdx.x < WL_BUILDERS) {
    if (blockIdx.x == 0 && threadIdx.x == 0) im.status[HGS_ST_LAZY] = b.lazy ? 1u : 0u;   // (for the blend kernels of this pass, backward included)
    const int share_w = ((T + WL_BUILDERS - 1) / WL_BUILDERS + HGS_BLOCK - 1) / HGS_BLOCK * HGS_BLOCK;
    __shared__ WlLds wl;
    if (share_w <= SK_WORDS * 4) work_list_shared(T, Rcap, pol, im, b, (uint16_t*)sk, wl);
    else work_list_block(T, Rcap, pol, im, b, (uint16_t*)sk, SK_WORDS * 4, wl);
    return;
  }
  if ((int)blockIdx.x >= T + WL_BUILDERS) {
    // ---- one chunk of a long list (work items from the scan: hgs_emit_sort_items)
    const uint32_t j = blockIdx.x - (uint32_t)T - WL_BUILDERS;
    // A pass that overflowed the binning capacity is void as a whole (status[1]; its backward returns zeros), and its chunk
    // items may outnumber the workgroups this launch was sized for (2 Rcap / SORT_CAP): a chunk whose sibling has no
    // workgroup would spin out its bounded wait -- seconds of GPU time -- and turn a recoverable overflow into a timeout.
    if (im.status[HGS_ST_R] > Rcap) return;
    if (j >= min(im.status[HGS_ST_SORT_ITEMS], (uint32_t)T)) return;
    const uint32_t item = im.sort_items[j];
    if (item == HGS_ITEM_NONE) return;
    const int tile = (int)HGS_ITEM_TILE(item);
    const uint32_t c = HGS_ITEM_PART(item);
    const uint2 range = im.ranges[tile];
    if (range.y > Rcap) return;                            // binning buffer overflow: the pass is void (status[1])
    const uint32_t start = range.x, n = range.y - range.x, nchunks = (n + SORT_CAP - 1) / SORT_CAP;
    const int tx = tile % gx, ty = tile / gx;
    const uint32_t cbase = start + c * SORT_CAP, cn = min((uint32_t)SORT_CAP, n - c * SORT_CAP);
    int m = 2;
    while ((uint32_t)m < cn) m <<= 1;
    for (int i = threadIdx.x; i < m; i += HGS_BLOCK) sk[i] = (uint32_t)i < cn ? b.keys[cbase + i] : ~0ull;
    __syncthreads();
    bitonic_lds(sk, m);
    // publish the sorted chunk in place (each chunk owns its part of the key array), keep this thread's keys
    uint64_t key[KPT];
    uint32_t rank[KPT];
#pragma unroll
    for (int i = 0; i < KPT; i++) {
      const uint32_t e = threadIdx.x + (uint32_t)i * HGS_BLOCK;
      key[i] = e < cn ? sk[e] : ~0ull;
      rank[i] = e;
      if (e < cn) hgs_st_agent((unsigned long long*)&b.keys[cbase + e], (unsigned long long)key[i]);
    }
    hgs_drain_stores();
    __syncthreads();
    __shared__ int ok;
    if (threadIdx.x == 0) {
      hgs_publish_part(&im.tile_sortprog[tile], c);
      ok = hgs_wait_parts(&im.tile_sortprog[tile], (1ull << nchunks) - 1ull, im.status) ? 1 : 0;
    }
    __syncthreads();
    if (!ok) return;
    for (uint32_t c2 = 0; c2 < nchunks; c2++) {
      if (c2 == c) continue;
      const uint32_t cn2 = min((uint32_t)SORT_CAP, n - c2 * SORT_CAP);
      __syncthreads();
      for (uint32_t i = threadIdx.x; i < cn2; i += HGS_BLOCK)
        sk[i] = hgs_ld_agent((const unsigned long long*)&b.keys[start + c2 * SORT_CAP + i]);
      __syncthreads();
      add_lower_bounds<KPT>(sk, cn2, key, rank);
    }
#pragma unroll
    for (int i = 0; i < KPT; i++) {
      const uint32_t e = threadIdx.x + (uint32_t)i * HGS_BLOCK;
      if (e < cn) emit_instance<EXTRA>(key[i], start + rank[i], start, tx, ty, g, b);
    }
    return;
  }
  const int tile = blockIdx.x - WL_BUILDERS;
  const uint2 range = im.ranges[tile];
  // empty, or binning buffer overflow (status[1] already set), or (more than one chunk: the only lists that can carry the
  // flag) a long list sorted by its chunk workgroups above
  if (range.y <= range.x || range.y > Rcap) return;
  if (range.y - range.x > SORT_CAP && (im.tile_sortprog[tile] & HGS_PART_FLAG)) return;
  const uint32_t start = range.x, n = range.y - range.x;
  const int tx = tile % gx, ty = tile / gx;
  if (n <= WAVE_SORT_MAX) {   // short list: one wavefront, no barriers (see bitonic_wave)
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    int m = 2;
    while ((uint32_t)m < n) m <<= 1;
    for (int i = lane; i < m; i += 64) sk[i] = (uint32_t)i < n ? b.keys[start + i] : ~0ull;
    wave_lds_fence();
    bitonic_wave(sk, m, lane);
    for (uint32_t i = lane; i < n; i += 64) emit_instance<EXTRA>(sk[i], start + i, start, tx, ty, g, b);
    return;
  }
  const uint32_t nchunks = (n + SORT_CAP - 1) / SORT_CAP;
  for (uint32_t c = 0; c < nchunks; c++) {
    const uint32_t cbase = start + c * SORT_CAP;
    const uint32_t cn = min((uint32_t)SORT_CAP, n - c * SORT_CAP);
    int m = 2;
    while ((uint32_t)m < cn) m <<= 1;
    for (int i = threadIdx.x; i < m; i += HGS_BLOCK) sk[i] = (uint32_t)i < cn ? b.keys[cbase + i] : ~0ull;
    __syncthreads();
    bitonic_lds(sk, m);
    if (nchunks == 1) {
      for (uint32_t i = threadIdx.x; i < cn; i += HGS_BLOCK) emit_instance<EXTRA>(sk[i], cbase + i, start, tx, ty, g, b);
    } else {
      for (uint32_t i = threadIdx.x; i < cn; i += HGS_BLOCK) b.keys[cbase + i] = sk[i];
    }
    __syncthreads();
  }
  if (nchunks > 1) {
    // Fallback for lists the chunk work list could not take (more than HGS_MAX_PARTS chunks, or the list was full): the
    // same rank merge, chunk after chunk in this workgroup.
    __threadfence_block();
    __syncthreads();
    for (uint32_t c = 0; c < nchunks; c++) {
      const uint32_t cn = min((uint32_t)SORT_CAP, n - c * SORT_CAP);
      uint64_t key[KPT];
      uint32_t rank[KPT];
#pragma unroll
      for (int i = 0; i < KPT; i++) {
        const uint32_t e = threadIdx.x + (uint32_t)i * HGS_BLOCK;
        key[i] = e < cn ? b.keys[start + c * SORT_CAP + e] : ~0ull;
        rank[i] = e;
      }
      for (uint32_t c2 = 0; c2 < nchunks; c2++) {
        if (c2 == c) continue;
        const uint32_t cn2 = min((uint32_t)SORT_CAP, n - c2 * SORT_CAP);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < cn2; i += HGS_BLOCK) sk[i] = b.keys[start + c2 * SORT_CAP + i];
        __syncthreads();
        add_lower_bounds<KPT>(sk, cn2, key, rank);
      }
#pragma unroll
      for (int i = 0; i < KPT; i++) {
        const uint32_t e = threadIdx.x + (uint32_t)i * HGS_BLOCK;
        if (e < cn) emit_instance<EXTRA>(key[i], start + rank[i], start, tx, ty, g, b);
      }
    }
  }
}

}  // namespace

static HgsSegPolicy g_seg_policy = {128u, 1024u, 2048u};
extern "C" int hgs_set_segment_policy(int min_len, int max_len, int target_segments) {
  if (min_len < HGS_SEG_MIN_LEN || (min_len & 63) || max_len < min_len || (max_len & 63) || target_segments < 1) {
    hgs_set_error("hgs_set_segment_policy: lengths must be multiples of 64 with 128 <= min <= max, target >= 1");
    return 1;
  }
  g_seg_policy.min_len = (uint32_t)min_len; g_seg_policy.max_len = (uint32_t)max_len; g_seg_policy.target = (uint32_t)target_segments;
  return 0;
}

int hgs_launch_scan(hipStream_t s, int P, int T, const HgsGeom& g, const HgsImage& im, unsigned int* max_rendered, int row_runs) {
  const int nblk = (P + HGS_BLOCK - 1) / HGS_BLOCK;
  {
    HgsProfScope _prof(s, HGS_K_SCAN);
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(SCAN_THREADS), 0, s, nblk, T, g, im, max_rendered, row_runs);
  }
  HGS_CHECK_LAUNCH();
  return 0;
}

int hgs_launch_sort_tiles(hipStream_t s, int W, int H, int Rcap, int n_extra, const HgsGeom& g, const HgsImage& im,
                          const HgsBinning& b) {
  const int gx = (W + HGS_TILE - 1) / HGS_TILE, gy = (H + HGS_TILE - 1) / HGS_TILE;
  {
    HgsProfScope _prof(s, HGS_K_SORT_TILES);
    // chunk work items: at most 2 Rcap / HGS_SORT_CAP (every long list has more than HGS_SORT_CAP entries), and at most T
    const int T = gx * gy;
    const long long items = 2ll * Rcap / HGS_SORT_CAP + 2;
    const int extra_wgs = (int)(items < T ? items : T);
    if (n_extra)
      hipLaunchKernelGGL(sort_tiles_kernel<true>, dim3(T + WL_BUILDERS + extra_wgs), dim3(HGS_BLOCK), 0, s, gx, T, (uint32_t)Rcap, g_seg_policy, g, im, b);
    else
      hipLaunchKernelGGL(sort_tiles_kernel<false>, dim3(T + WL_BUILDERS + extra_wgs), dim3(HGS_BLOCK), 0, s, gx, T, (uint32_t)Rcap, g_seg_policy, g, im, b);
  }
  HGS_CHECK_LAUNCH();
  return 0;
}
