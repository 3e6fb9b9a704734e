// hgs_knn.hip -- distCUDA2: mean squared distance to the 3 nearest neighbours (simple-knn/simple_knn.cu).
//
// Same plan as the reference (origin-including bounding box :192-201, 30-bit Morton codes :46-71, stable sort
// by code :207-214, 1024-point boxes :79-118, box-pruned exact search :148-184) with these CDNA4 choices:
//   * no host round trips: min/max stay on the device (the reference does two blocking D2H copies);
//   * the stable (code, index) sort is a sort by the unique 64-bit key code<<32|index -> plain bitonic network
//     (LDS for strides < 4096, global otherwise); called once per run, P <= a few million;
//   * points are gathered once into Morton order; the search (round 6) runs one lane per point over a GRID OF MORTON CELLS:
//     the points that share the top 3 L bits of their code are one contiguous run of the sorted array and fill one cube, so a
//     table of 8^L cells (first / last position, the cell's exact bounding box: one pass of integer atomics) is an octree level
//     whose boxes do not overlap.  A lane walks its own cell first, then the cells of the cube range its radius -- min(reject,
//     third-best so far) -- reaches, each behind the reference's box test on the cell's own box.  Rounds 1-5 tested boxes of
//     1024 Morton-CONSECUTIVE points: such a run straddles the curve's jumps, its bounding box holds most of the scene, every
//     point lies inside most boxes and the pruning pruned nothing (3.8 ms for 50 k points, 96 % of the call in the search).
//     The three smallest distances of a point do not depend on the traversal, so the result is exactly the reference's (the
//     exact 3-NN in its arithmetic).
#include <float.h>

#include "hgs_common.h"

namespace {

#define KNN_LDS_KEYS 4096

struct KnnScratch { float* minmax; uint64_t* keys; float4* sorted; uint32_t* cells; };   // cells: 8 words per cell

__device__ __forceinline__ uint32_t prep_morton(uint32_t x) {
  x = (x | (x << 16)) & 0x030000FF;
  x = (x | (x << 8)) & 0x0300F00F;
  x = (x | (x << 4)) & 0x030C30C3;
  x = (x | (x << 2)) & 0x09249249;
  return x;
}
__device__ __forceinline__ uint32_t f2u_sat(float v) {
  if (!(v > 0.f)) return 0u;
  if (v >= 4294967040.f) return 0xFFFFFFFFu;
  return (uint32_t)v;
}

// component-wise min/max with init (0,0,0) (simple_knn.cu:192): MM_BLOCKS partial results, folded by whoever reads them
#define MM_BLOCKS 64
__global__ __launch_bounds__(1024) void minmax_kernel(int P, const float* __restrict__ pts, float* __restrict__ mm) {
  __shared__ float red[16][6];
  float mn[3] = {0.f, 0.f, 0.f}, mx[3] = {0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < (size_t)P; i += (size_t)MM_BLOCKS * 1024)
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const float v = pts[3 * i + k];
      mn[k] = fminf(mn[k], v);
      mx[k] = fmaxf(mx[k], v);
    }
#pragma unroll
  for (int k = 0; k < 3; k++)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      mn[k] = fminf(mn[k], __shfl_xor(mn[k], d, 64));
      mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], d, 64));
    }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0)
    for (int k = 0; k < 3; k++) { red[wave][k] = mn[k]; red[wave][3 + k] = mx[k]; }
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = red[0][threadIdx.x];
    for (int w = 1; w < 16; w++) v = threadIdx.x < 3 ? fminf(v, red[w][threadIdx.x]) : fmaxf(v, red[w][threadIdx.x]);
    mm[8 * (size_t)blockIdx.x + threadIdx.x] = v;
  }
}

__global__ __launch_bounds__(256) void morton_kernel(int P, int Npad, const float* __restrict__ pts,
                                                     const float* __restrict__ mm, uint64_t* __restrict__ keys) {
  __shared__ float smm[6];
  if (threadIdx.x < 6) {                      // min / max are exact: the order of folding the partial results does not matter
    float v = mm[threadIdx.x];
    for (int b = 1; b < MM_BLOCKS; b++) v = threadIdx.x < 3 ? fminf(v, mm[8 * b + threadIdx.x]) : fmaxf(v, mm[8 * b + threadIdx.x]);
    smm[threadIdx.x] = v;
  }
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Npad) return;
  if (i >= P) { keys[i] = ~0ull; return; }
  const float mnx = smm[0], mny = smm[1], mnz = smm[2], mxx = smm[3], mxy = smm[4], mxz = smm[5];
  const float x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
  const uint32_t cx = prep_morton(f2u_sat(((x - mnx) / (mxx - mnx)) * 1023));
  const uint32_t cy = prep_morton(f2u_sat(((y - mny) / (mxy - mny)) * 1023));
  const uint32_t cz = prep_morton(f2u_sat(((z - mnz) / (mxz - mnz)) * 1023));
  keys[i] = ((uint64_t)(cx | (cy << 1) | (cz << 2)) << 32) | (uint32_t)i;
}

// bitonic steps j = jstart .. 1 of stage k inside LDS chunks of KNN_LDS_KEYS keys (jstart < KNN_LDS_KEYS);
// with full=true runs every stage k = 2..KNN_LDS_KEYS (initial chunk sort)
__global__ __launch_bounds__(1024) void bitonic_lds_kernel(uint64_t* __restrict__ keys, int k_stage, int jstart, bool full) {
  __shared__ uint64_t sk[KNN_LDS_KEYS];
  const size_t base = (size_t)blockIdx.x * KNN_LDS_KEYS;
  for (int i = threadIdx.x; i < KNN_LDS_KEYS; i += 1024) sk[i] = keys[base + i];
  __syncthreads();
  const int k0 = full ? 2 : k_stage, k1 = full ? KNN_LDS_KEYS : k_stage;
  for (int k = k0; k <= k1; k <<= 1) {
    for (int j = full ? (k >> 1) : jstart; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < KNN_LDS_KEYS / 2; t += 1024) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int l = i | j;
        const bool asc = (((base + i) & (size_t)k) == 0);
        const uint64_t x = sk[i], y = sk[l];
        if ((x > y) == asc) { sk[i] = y; sk[l] = x; }
      }
      __syncthreads();
    }
    if (!full) break;
  }
  for (int i = threadIdx.x; i < KNN_LDS_KEYS; i += 1024) keys[base + i] = sk[i];
}

__global__ __launch_bounds__(256) void bitonic_global_kernel(uint64_t* __restrict__ keys, size_t half, int k, int j) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= half) return;
  const size_t i = ((t & ~((size_t)j - 1)) << 1) | (t & ((size_t)j - 1));
  const size_t l = i | (size_t)j;
  const bool asc = (i & (size_t)k) == 0;
  const uint64_t x = keys[i], y = keys[l];
  if ((x > y) == asc) { keys[i] = y; keys[l] = x; }
}

__global__ __launch_bounds__(256) void gather_kernel(int P, const float* __restrict__ pts, const uint64_t* __restrict__ keys,
                                                     float4* __restrict__ sorted) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const uint32_t id = (uint32_t)keys[i];
  sorted[i] = make_float4(pts[3 * (size_t)id], pts[3 * (size_t)id + 1], pts[3 * (size_t)id + 2], __uint_as_float(id));
}

__device__ __forceinline__ void kbest3(float px, float py, float pz, float qx, float qy, float qz, float* knn) {
  const float dx = qx - px, dy = qy - py, dz = qz - pz;
  float dist = dx * dx + dy * dy + dz * dz;  // simple_knn.cu:135-136 (no contraction: built with -ffp-contract=off)
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const float t = knn[j];
    const bool sw = t > dist;
    knn[j] = sw ? dist : t;
    dist = sw ? t : dist;
  }
}

// squared distance of a point to a box, exactly as distBoxPoint (simple_knn.cu:120-130)
__device__ __forceinline__ float box_point_dist2(const float* bx, float x, float y, float z) {
  float d0 = 0.f, d1 = 0.f, d2 = 0.f;
  if (x < bx[0] || x > bx[3]) d0 = fminf(fabsf(x - bx[0]), fabsf(x - bx[3]));
  if (y < bx[1] || y > bx[4]) d1 = fminf(fabsf(y - bx[1]), fabsf(y - bx[4]));
  if (z < bx[2] || z > bx[5]) d2 = fminf(fabsf(z - bx[2]), fabsf(z - bx[5]));
  return d0 * d0 + d1 * d1 + d2 * d2;
}

// ---- the cell table: per cell of level L (code >> (30 - 3 L)) [first, end) in the sorted array and the points' bounding box.
// Floats are kept as order-preserving integers so that min / max are integer atomics (exact, order-independent).
__device__ __forceinline__ int fkey(float v) { const int b = __float_as_int(v); return b >= 0 ? b : b ^ 0x7FFFFFFF; }   // monotone: a < b <=> fkey(a) < fkey(b)
__device__ __forceinline__ float funkey(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7FFFFFFF); }

__global__ __launch_bounds__(256) void cells_clear_kernel(uint32_t n_cells, uint32_t* __restrict__ cells) {
  const uint32_t c = blockIdx.x * 256 + threadIdx.x;
  if (c >= n_cells) return;
  uint4* t = (uint4*)(cells + 8 * (size_t)c);
  t[0] = make_uint4(0u, 0u, (uint32_t)0x7FFFFFFF, (uint32_t)0x7FFFFFFF);              // first, end, min x, min y
  t[1] = make_uint4((uint32_t)0x7FFFFFFF, 0x80000000u, 0x80000000u, 0x80000000u);     // min z, max x, max y, max z
}
__global__ __launch_bounds__(256) void cells_mark_kernel(int P, int shift, const uint64_t* __restrict__ keys,
                                                         const float4* __restrict__ sorted, uint32_t* __restrict__ cells) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const uint32_t c = (uint32_t)(keys[i] >> 32) >> shift;
  const bool first = i == 0 || ((uint32_t)(keys[i - 1] >> 32) >> shift) != c;
  const bool last = i == P - 1 || ((uint32_t)(keys[i + 1] >> 32) >> shift) != c;
  uint32_t* t = cells + 8 * (size_t)c;
  if (first) t[0] = (uint32_t)i;
  if (last) t[1] = (uint32_t)i + 1u;
  const float4 p = sorted[i];
  atomicMin((int*)&t[2], fkey(p.x)); atomicMin((int*)&t[3], fkey(p.y)); atomicMin((int*)&t[4], fkey(p.z));
  atomicMax((int*)&t[5], fkey(p.x)); atomicMax((int*)&t[6], fkey(p.y)); atomicMax((int*)&t[7], fkey(p.z));
}

// One lane per point of the sorted array.  `mm`: the MM_BLOCKS partial min / max (the Morton kernel's quantisation).
__global__ __launch_bounds__(256) void mean_dist_kernel(int P, int L, const float4* __restrict__ sorted,
                                                        const uint64_t* __restrict__ keys, const uint32_t* __restrict__ cells,
                                                        const float* __restrict__ mm, float* __restrict__ out) {
  __shared__ float smm[6];
  if (threadIdx.x < 6) {
    float v = mm[threadIdx.x];
    for (int b = 1; b < MM_BLOCKS; b++) v = threadIdx.x < 3 ? fminf(v, mm[8 * b + threadIdx.x]) : fmaxf(v, mm[8 * b + threadIdx.x]);
    smm[threadIdx.x] = v;
  }
  __syncthreads();
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= P) return;
  const float4 me = sorted[idx];
  float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
  {
    const int lo = max(0, idx - 3), hi = min(P - 1, idx + 3);
    for (int i = lo; i <= hi; i++) {
      if (i == idx) continue;
      const float4 q = sorted[i];
      kbest3(me.x, me.y, me.z, q.x, q.y, q.z, best);
    }
  }
  const float reject = best[2];               // simple_knn.cu:163-165: a box farther than this holds none of the three nearest
  best[0] = best[1] = best[2] = FLT_MAX;
  const int shift = 30 - 3 * L;
  auto walk_cell = [&](uint32_t c) {
    const uint4 h0 = *(const uint4*)(cells + 8 * (size_t)c), h1 = *(const uint4*)(cells + 8 * (size_t)c + 4);
    if (h0.x == h0.y) return;                                           // empty
    const float bx[6] = {funkey((int)h0.z), funkey((int)h0.w), funkey((int)h1.x), funkey((int)h1.y), funkey((int)h1.z), funkey((int)h1.w)};
    const float dist = box_point_dist2(bx, me.x, me.y, me.z);
    if (dist > reject || dist > best[2]) return;                        // simple_knn.cu:173-175
    for (uint32_t i = h0.x; i < h0.y; i++) {
      const float4 q = sorted[i];
      if ((int)i != idx) kbest3(me.x, me.y, me.z, q.x, q.y, q.z, best);
    }
  };
  // the own cell first: after it the third-best distance is (nearly always) the true one
  const uint32_t own = (uint32_t)(keys[idx] >> 32) >> shift;
  walk_cell(own);
  // the cube of cells the radius reaches.  The quantisation q(x) of the Morton kernel is monotone in x, so every point within
  // r of this one along an axis has its cell coordinate between q(x - r) and q(x + r).
  const float r2 = fminf(reject, best[2]);
  const float r = sqrtf(r2) * 1.000001f;
  const int cshift = 10 - L;
  int lo[3], hi[3];
  const float pc[3] = {me.x, me.y, me.z};
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float mn = smm[k], mx = smm[3 + k];
    lo[k] = (int)(min(f2u_sat(((pc[k] - r - mn) / (mx - mn)) * 1023), 1023u) >> cshift);
    hi[k] = (int)(min(f2u_sat(((pc[k] + r - mn) / (mx - mn)) * 1023), 1023u) >> cshift);
    if (!(r < FLT_MAX)) { lo[k] = 0; hi[k] = (1 << L) - 1; }            // (fewer than four points: no radius yet)
  }
  for (int cz = lo[2]; cz <= hi[2]; cz++)
    for (int cy = lo[1]; cy <= hi[1]; cy++)
      for (int cx = lo[0]; cx <= hi[0]; cx++) {
        const uint32_t c = (prep_morton((uint32_t)cx << cshift) | (prep_morton((uint32_t)cy << cshift) << 1) |
                            (prep_morton((uint32_t)cz << cshift) << 2)) >> shift;
        if (c != own) walk_cell(c);
      }
  out[__float_as_uint(me.w)] = (best[0] + best[1] + best[2]) / 3.0f;
}

// level of the cell grid: about 16 points per cell of a uniform cloud (clustered data fills fewer, fuller cells)
static int knn_level(size_t P) {
  int L = 1;
  while (L < 7 && ((size_t)1 << (3 * (L + 1))) * 16 <= P) L++;
  return L;
}

size_t knn_carve(char* base, size_t P, size_t Npad, KnnScratch& s) {
  char* cur = base;
  hgs_carve(cur, s.minmax, 8 * MM_BLOCKS);
  hgs_carve(cur, s.keys, Npad);
  hgs_carve(cur, s.sorted, P + 1);
  hgs_carve(cur, s.cells, 8 * ((size_t)1 << (3 * knn_level(P))));
  return hgs_align_up((size_t)(cur - base)) + HGS_ALIGN;
}
size_t pad_pow2(size_t P) {
  size_t n = KNN_LDS_KEYS;
  while (n < P) n <<= 1;
  return n;
}

}  // namespace

size_t hgs_dist2_scratch(int P) {
  KnnScratch s;
  return knn_carve(nullptr, (size_t)P, pad_pow2((size_t)P), s);
}

int hgs_launch_dist2(hipStream_t st, int P, const float* points, float* out, void* scratch, size_t scratch_bytes) {
  const size_t Npad = pad_pow2((size_t)P);
  KnnScratch s;
  knn_carve((char*)scratch, (size_t)P, Npad, s);
  if (hgs_dist2_scratch(P) > scratch_bytes || ((size_t)scratch & (HGS_ALIGN - 1))) {
    hgs_set_error("hgs_dist2: scratch must be %d-byte aligned and >= %zu bytes (got %zu)", HGS_ALIGN, hgs_dist2_scratch(P), scratch_bytes);
    return 1;
  }
  HgsProfScope _prof(st, HGS_K_KNN);
  hipLaunchKernelGGL(minmax_kernel, dim3(MM_BLOCKS), dim3(1024), 0, st, P, points, s.minmax);
  hipLaunchKernelGGL(morton_kernel, dim3((unsigned)((Npad + 255) / 256)), dim3(256), 0, st, P, (int)Npad, points, s.minmax, s.keys);
  const unsigned nchunks = (unsigned)(Npad / KNN_LDS_KEYS);
  hipLaunchKernelGGL(bitonic_lds_kernel, dim3(nchunks), dim3(1024), 0, st, s.keys, 0, 0, true);
  for (size_t k = 2 * KNN_LDS_KEYS; k <= Npad; k <<= 1) {
    size_t j = k >> 1;
    for (; j >= KNN_LDS_KEYS; j >>= 1)
      hipLaunchKernelGGL(bitonic_global_kernel, dim3((unsigned)((Npad / 2 + 255) / 256)), dim3(256), 0, st, s.keys, Npad / 2, (int)k, (int)j);
    hipLaunchKernelGGL(bitonic_lds_kernel, dim3(nchunks), dim3(1024), 0, st, s.keys, (int)k, (int)j, false);
  }
  const int L = knn_level((size_t)P);
  const unsigned n_cells = 1u << (3 * L);
  hipLaunchKernelGGL(gather_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, points, s.keys, s.sorted);
  hipLaunchKernelGGL(cells_clear_kernel, dim3((n_cells + 255) / 256), dim3(256), 0, st, n_cells, s.cells);
  hipLaunchKernelGGL(cells_mark_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, 30 - 3 * L, s.keys, s.sorted, s.cells);
  hipLaunchKernelGGL(mean_dist_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, L, s.sorted, s.keys, s.cells, s.minmax, out);
  HGS_CHECK_LAUNCH();
  return 0;
}

// ---- fixed-radius pair search over strand ends (hgs_radius_pairs) ------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void radius_pairs_kernel(int N, const float* __restrict__ pos, const float* __restrict__ dir,
                                                           float r2, float min_cos, int bidirectional, int capacity,
                                                           int* __restrict__ pairs, float* __restrict__ dist,
                                                           int* __restrict__ count, float reach_x) {
  __shared__ float sp[256][3], sd[256][3];
  const int a = blockIdx.x * 256 + threadIdx.x;
  float ax = 0.f, ay = 0.f, az = 0.f, ux = 0.f, uy = 0.f, uz = 0.f;
  if (a < N) { ax = pos[3 * a]; ay = pos[3 * a + 1]; az = pos[3 * a + 2]; ux = dir[3 * a]; uy = dir[3 * a + 1]; uz = dir[3 * a + 2]; }
  // only tiles at or after this block's own: every unordered pair is tested once, by the block of its smaller index
  // reach_x >= 0: the points are sorted by x; a tile that starts farther than the radius beyond this block's last point
  // (and every tile after it) holds no partner
  const float block_max_x = reach_x >= 0.f ? pos[3 * min(N - 1, (int)blockIdx.x * 256 + 255)] + reach_x : 0.f;
  for (int t0 = blockIdx.x * 256; t0 < N; t0 += 256) {
    if (reach_x >= 0.f && pos[3 * t0] > block_max_x) break;
    const int j = t0 + threadIdx.x;
    __syncthreads();
    if (j < N) {
#pragma unroll
      for (int c = 0; c < 3; c++) { sp[threadIdx.x][c] = pos[3 * j + c]; sd[threadIdx.x][c] = dir[3 * j + c]; }
    }
    __syncthreads();
    const int nb = min(256, N - t0);
    if (a < N) {
      for (int k = 0; k < nb; k++) {                       // broadcast LDS reads
        const int b = t0 + k;
        if (b <= a) continue;
        const float dx = sp[k][0] - ax, dy = sp[k][1] - ay, dz = sp[k][2] - az;
        const float d2 = dx * dx + dy * dy + dz * dz;
        if (d2 > r2) continue;
        float dot = -(ux * sd[k][0] + uy * sd[k][1] + uz * sd[k][2]);
        if (bidirectional) dot = fabsf(dot);
        if (!(dot >= min_cos)) continue;
        const int slot = atomicAdd(count, 1);
        if (slot < capacity) { pairs[2 * slot] = a; pairs[2 * slot + 1] = b; dist[slot] = sqrtf(d2); }
      }
    }
  }
}
}  // namespace

extern "C" int hgs_radius_pairs(void* stream, int N, const float* pos, const float* dir, float radius, float min_cos,
                                int bidirectional, int capacity, int* pairs, float* dist, int* count, int sorted_by_x) {
  if (N <= 1) return 0;
  if (!pos || !dir || !count || capacity < 0 || (capacity > 0 && (!pairs || !dist))) { hgs_set_error("hgs_radius_pairs: bad arguments"); return 1; }
  hipStream_t s = (hipStream_t)stream;
  {
    HgsProfScope _prof(s, HGS_K_KNN);
    hipLaunchKernelGGL(radius_pairs_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, pos, dir, radius * radius, min_cos,
                       bidirectional, capacity, pairs, dist, count, sorted_by_x ? radius : -1.f);
  }
  HGS_CHECK_LAUNCH();
  return 0;
}

// ---- K = 3 nearest neighbours of every point within its own set (hgs_knn3) -----------------------------------------------
// What the magnet loss asks pytorch3d for (loss/losses.py:139-144: knn_points(ends, ends, K=3, return_sorted=True)) over
// the strand ends -- thousands to 10^5 points that move every iteration, so no tree: one query per lane, all points
// through LDS in 256-point tiles (broadcast reads), the three best kept sorted in registers.  The point itself is a
// candidate like any other (pytorch3d returns it first, at distance 0); ties are ordered by index.
namespace {
__global__ __launch_bounds__(256) void knn3_kernel(int N, const float* __restrict__ pos, int* __restrict__ idx, float* __restrict__ d2out) {
  __shared__ float sp[256][3];
  const int a = blockIdx.x * 256 + threadIdx.x;
  float ax = 0.f, ay = 0.f, az = 0.f;
  if (a < N) { ax = pos[3 * a]; ay = pos[3 * a + 1]; az = pos[3 * a + 2]; }
  float b0 = INFINITY, b1 = INFINITY, b2 = INFINITY;
  int i0 = -1, i1 = -1, i2 = -1;
  for (int t0 = 0; t0 < N; t0 += 256) {
    const int j = t0 + threadIdx.x;
    __syncthreads();
    if (j < N) {
#pragma unroll
      for (int c = 0; c < 3; c++) sp[threadIdx.x][c] = pos[3 * j + c];
    }
    __syncthreads();
    const int nb = min(256, N - t0);
    for (int k = 0; k < nb; k++) {
      const float dx = ax - sp[k][0], dy = ay - sp[k][1], dz = az - sp[k][2];
      const float d = dx * dx + dy * dy + dz * dz;
      if (d < b2) {                               // (indices ascend along the walk: an equal distance never displaces)
        if (d < b1) {
          b2 = b1; i2 = i1;
          if (d < b0) { b1 = b0; i1 = i0; b0 = d; i0 = t0 + k; }
          else { b1 = d; i1 = t0 + k; }
        } else { b2 = d; i2 = t0 + k; }
      }
    }
  }
  if (a < N) {
    idx[3 * a] = i0; idx[3 * a + 1] = i1; idx[3 * a + 2] = i2;
    d2out[3 * a] = b0; d2out[3 * a + 1] = b1; d2out[3 * a + 2] = b2;
  }
}
}  // namespace

extern "C" int hgs_knn3(void* stream, int N, const float* points, int* idx, float* dist2) {
  if (N <= 0) return 0;
  if (!points || !idx || !dist2) { hgs_set_error("hgs_knn3: null argument"); return 1; }
  hipStream_t s = (hipStream_t)stream;
  {
    HgsProfScope _prof(s, HGS_K_KNN);
    hipLaunchKernelGGL(knn3_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, points, idx, dist2);
  }
  HGS_CHECK_LAUNCH();
  return 0;
}

// ---- distance to the nearest of a small set of reference points, in float64 ------------------------------------------------
// What compute_strands_info asks a scipy cKDTree of the reference strand roots for (scene/hair_gaussian_model.py:1466-1470: the
// two ends of every strand, to orient it root -> tip): a few thousand roots against 10^5-10^6 strand ends.  One lane per point,
// the roots staged through LDS in float64; the distance is sqrt(dx^2 + dy^2 + dz^2) evaluated in float64 in that order (what a
// brute force over the tree's points gives).  torch.cdist's float64 kernel took 8 ms per call for this shape -- a third of the
// GPU time of a training run with the topology operators.
#define ND_TILE 1024
__global__ __launch_bounds__(256) void nearest_dist_f64_kernel(int N, int M, const float* __restrict__ pts, const double* __restrict__ refs,
                                                               double* __restrict__ out) {
#pragma clang fp contract(off)
  __shared__ double r[ND_TILE * 3];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool live = i < N;
  const double px = live ? (double)pts[3 * (size_t)i] : 0.0, py = live ? (double)pts[3 * (size_t)i + 1] : 0.0,
               pz = live ? (double)pts[3 * (size_t)i + 2] : 0.0;
  double best = __builtin_inf();
  for (int m0 = 0; m0 < M; m0 += ND_TILE) {
    const int cnt = min(ND_TILE, M - m0);
    __syncthreads();
    for (int k = threadIdx.x; k < cnt * 3; k += 256) r[k] = refs[(size_t)m0 * 3 + k];
    __syncthreads();
    for (int k = 0; k < cnt; k++) {
      const double dx = px - r[3 * k], dy = py - r[3 * k + 1], dz = pz - r[3 * k + 2];
      const double d2 = dx * dx + dy * dy + dz * dz;
      best = d2 < best ? d2 : best;
    }
  }
  if (live) out[i] = sqrt(best);
}

extern "C" int hgs_nearest_distance_f64(void* stream, int N, int M, const float* points, const double* refs, double* out) {
  if (N < 0 || M < 0) { hgs_set_error("hgs_nearest_distance_f64: bad sizes"); return 1; }
  if (N == 0) return 0;
  if (M == 0 || !points || !refs || !out) { hgs_set_error("hgs_nearest_distance_f64: no reference points / null argument"); return 1; }
  hipLaunchKernelGGL(nearest_dist_f64_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, M, points, refs, out);
  HGS_CHECK_LAUNCH();
  return 0;
}

// ---- strand walk (hgs_strand_walk_ends / hgs_strand_walk_fill) -------------------------------------------------------------------
// compute_strands_info (reference scene/hair_gaussian_model.py:1410-1498) orders every open polyline of the segment table from one
// end to the other.  The torch form (walk_chains_torch) doubles pointers over the 2 n arcs: ~10 rounds of gathers over every arc,
// ~120 launches, 2.7 ms on a 4 10^5-segment model, twice per topology event.  A strand is a CHAIN: here one lane per strand END
// walks it -- one 16-byte load per step from a node table (for every endpoint its at most two (row, neighbour) entries) -- first
// to find the other end and the length, then (one lane per strand, once the host side has numbered the strands and decided their
// direction) to write the ordered rows.  Chains of ~100 segments: ~100 dependent L2 hits, tens of microseconds, all strands in
// parallel.  Closed loops have no end and are never entered, like in the reference's walk.
struct WalkNode { int row0, nb0, row1, nb1; };      // (-1: no entry)

__global__ __launch_bounds__(256) void walk_nodes_kernel(int n, int n_ep, const long long* __restrict__ pairs, int* __restrict__ deg,
                                                         WalkNode* __restrict__ nodes, int* __restrict__ flags) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  const long long a = pairs[2 * (size_t)r], b = pairs[2 * (size_t)r + 1];
  if (a < 0 || b < 0 || a >= n_ep || b >= n_ep) { flags[0] = 1; return; }
#pragma unroll
  for (int s = 0; s < 2; s++) {
    const int id = (int)(s ? b : a), nb = (int)(s ? a : b);
    const int slot = atomicAdd(&deg[id], 1);
    if (slot == 0) { nodes[id].row0 = r; nodes[id].nb0 = nb; }
    else if (slot == 1) { nodes[id].row1 = r; nodes[id].nb1 = nb; }
    else flags[0] = 1;                                   // an endpoint of degree > 2: not a set of chains
  }
}

// other[e] = the end the chain that starts at end e runs into, len[e] = its segments; -1 / 0 for ids that are no chain end
__global__ __launch_bounds__(256) void walk_ends_kernel(int n, int n_ep, const int* __restrict__ deg, const WalkNode* __restrict__ nodes,
                                                        int* __restrict__ other, int* __restrict__ len, int* __restrict__ flags) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n_ep) return;
  int o = -1, L = 0;
  if (deg[e] == 1) {
    const int4 first = *(const int4*)&nodes[e];
    int row = first.x, v = first.y;                      // arrived at v through `row`
    L = 1;
    while (L <= n) {
      const int4 nd = *(const int4*)&nodes[v];
      const bool use1 = nd.x == row;                     // leave through the entry that is not the row we came by
      const int nrow = use1 ? nd.z : nd.x, nnb = use1 ? nd.w : nd.y;
      if (nrow < 0) break;                               // v has no other row: the chain's other end
      row = nrow; v = nnb; L++;
    }
    if (L > n) { flags[0] = 1; L = 0; } else o = v;
  }
  other[e] = o; len[e] = L;
}

__global__ __launch_bounds__(64) void walk_fill_kernel(int S, const long long* __restrict__ starts, const long long* __restrict__ offsets,
                                                       const unsigned char* __restrict__ flip, const WalkNode* __restrict__ nodes,
                                                       long long* __restrict__ rows, long long* __restrict__ seg_rows,
                                                       int* __restrict__ id_to_strand) {
  const int s = blockIdx.x * 64 + threadIdx.x;
  if (s >= S) return;
  const long long o0 = offsets[s], o1 = offsets[s + 1];
  const bool rev = flip[s] != 0;
  int cur = (int)starts[s];
  const int4 first = *(const int4*)&nodes[cur];
  int row = first.x, v = first.y;
  id_to_strand[cur] = s;
  for (long long k = o0; k < o1; k++) {
    const long long at = rev ? o1 - 1 - (k - o0) : k;
    rows[2 * at] = rev ? v : cur;
    rows[2 * at + 1] = rev ? cur : v;
    seg_rows[at] = row;
    id_to_strand[v] = s;
    const int4 nd = *(const int4*)&nodes[v];
    const bool use1 = nd.x == row;
    cur = v;
    row = use1 ? nd.z : nd.x;
    v = use1 ? nd.w : nd.y;
  }
}

extern "C" int hgs_strand_walk_ends(void* stream, int n, int n_ep, const long long* pairs, int* deg, void* nodes, int* other, int* len,
                                    int* flags) {
  if (n < 0 || n_ep < 0) { hgs_set_error("hgs_strand_walk_ends: bad sizes"); return 1; }
  if (n_ep == 0) return 0;
  if ((n > 0 && !pairs) || !deg || !nodes || !other || !len || !flags || ((size_t)nodes & 15)) {
    hgs_set_error("hgs_strand_walk_ends: null argument (or nodes not 16-byte aligned)"); return 1;
  }
  hipStream_t s = (hipStream_t)stream;
  HGS_CHECK_HIP(hipMemsetAsync(deg, 0, sizeof(int) * (size_t)n_ep, s));
  HGS_CHECK_HIP(hipMemsetAsync(nodes, 0xFF, sizeof(WalkNode) * (size_t)n_ep, s));
  HGS_CHECK_HIP(hipMemsetAsync(flags, 0, sizeof(int), s));
  if (n > 0) hipLaunchKernelGGL(walk_nodes_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, n_ep, pairs, deg, (WalkNode*)nodes, flags);
  hipLaunchKernelGGL(walk_ends_kernel, dim3((n_ep + 255) / 256), dim3(256), 0, s, n, n_ep, deg, (const WalkNode*)nodes, other, len, flags);
  HGS_CHECK_LAUNCH();
  return 0;
}

extern "C" int hgs_strand_walk_fill(void* stream, int S, const long long* starts, const long long* offsets, const unsigned char* flip,
                                    const void* nodes, long long* rows, long long* seg_rows, int* id_to_strand) {
  if (S < 0) { hgs_set_error("hgs_strand_walk_fill: bad size"); return 1; }
  if (S == 0) return 0;
  if (!starts || !offsets || !flip || !nodes || !rows || !seg_rows || !id_to_strand) { hgs_set_error("hgs_strand_walk_fill: null argument"); return 1; }
  hipLaunchKernelGGL(walk_fill_kernel, dim3((S + 63) / 64), dim3(64), 0, (hipStream_t)stream, S, starts, offsets, flip, (const WalkNode*)nodes,
                     rows, seg_rows, id_to_strand);
  HGS_CHECK_LAUNCH();
  return 0;
}

