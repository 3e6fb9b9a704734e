// hgs_common.h -- shared declarations of libhgs.so (gfx950 only; wave64 is assumed everywhere).
//
// Workspace layout (all sub-arrays 256-B aligned inside caller-owned byte buffers):
//   geom    (per Gaussian)  depths, clamped, means2D, cov3D, conic_opacity, rgb, tiles_touched,
//                           point_offsets, rect(+exclusive instance offset), block_sums
//   image   (per pixel/tile) final_T, n_contrib, ranges, tile_count, tile_cursor, tile_maxc, tile_done, status, tile_prog,
//                           tile_sortprog, sort_items, tile_order
//   binning (per instance)  keys (depth<<32|id<<4|quadrant mask), point_list, packed records, (reserved), sorted keys,
//                           + per list SEGMENT (long tile lists are split across workgroups, see hgs_blend.hip):
//                           seg_P, seg_T, seg_Tout, seg_last, seg_C
// They play the roles of GeometryState / ImageState / BinningState of the reference
// (cuda_rasterizer/rasterizer_impl.h:23-71) but the layout is this library's own.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/hgs.h"

#define HGS_BLOCK 256   // threads per workgroup for per-Gaussian kernels and per-tile kernels (4 waves)
#define HGS_WAVE 64
#define HGS_ALIGN 256

struct HgsRect {        // 16 B per Gaussian: tile rectangle + exclusive instance offset
  uint16_t x0, y0, x1, y1;
  uint32_t off;         // exclusive prefix of tiles_touched (filled by the scatter kernel)
  uint32_t pad;
};

struct HgsGeom {
  float* depths; uint8_t* clamped; float2* means2D; float* cov3D; float4* conic_opacity; float* rgb;
  uint32_t* tiles_touched; uint32_t* point_offsets; HgsRect* rect; uint32_t* block_sums;
  // 64 B per Gaussian, everything an instance record needs, assembled once by the scatter kernel so that the sort kernel's
  // emission is ONE contiguous gather per instance instead of five scattered ones:
  //   [x, y, conic a, b | conic c, opacity, f0, f1 | f2, extra0..2 | extra3, offset, x0 | y0 << 16, width of the rectangle]
  float4* grec;
};
struct HgsImage {
  float* final_T; uint32_t* n_contrib; uint2* ranges; uint32_t* tile_count; uint32_t* tile_cursor;
  // row runs of large tile rectangles (HGS_COUNT_ROW_RUNS passes): +1 at (row, x0), -1 at (row, x1) of the tiles in row-major
  // order, T + 1 entries; the pass's scan adds the running sum to the tiles' counts and leaves the array at zero
  int32_t* tile_delta;
  uint32_t* tile_maxc; uint32_t* status; uint32_t* tile_order; uint32_t* wl_exchange;
  uint32_t tile_mask;   // slots of tile_count / tile_cursor - 1 (hgs_tile_slot)
  // long tile lists (hgs_binning.hip / hgs_blend.hip): per-tile ticket of finished blend segments, bit masks of published
  // blend segments / sorted chunks (bit 63: the list is handled by several workgroups), chunk work items of the sort
  uint32_t* tile_done; unsigned long long* tile_prog; unsigned long long* tile_sortprog; uint32_t* sort_items;
};
struct HgsBinning {
  uint64_t* keys; uint32_t* point_list; float4* packed; uint32_t* inv; uint64_t* keys_sorted;
  // per list segment of a split tile (index = the segment's position in the work list im.tile_order), 256 pixels each:
  float* seg_P;         // transmittance product of the segment's entries (forward, phase 1)
  float* seg_T;         // transmittance in front of the segment (kept for the backward)
  float* seg_Tout;      // transmittance after the segment's walk, negative: the pixel stopped in it
  uint32_t* seg_last;   // last contributing list position + 1 inside the segment (0: none)
  float* seg_C;         // [C][256] colour added by the segment; after the tile's finalisation: by it and all behind it
  uint32_t seg_cap;     // segments the arrays hold
  // Lazy records (round 6).  lazy != 0: the sort kernel orders keys only, and the blend kernels stage an entry's record from its
  // Gaussian's 64-byte template (grec = HgsGeom::grec of the pass) and the sorted key -- quadrant mask = the key's low bits,
  // gradient-row slot = a function of the template's rectangle and the tile.  lazy == 0: the sort kernel gathers the template and
  // writes a 48- / 64-byte record per instance into `packed`, which the blend kernels stream.  Both set by the host side next to
  // every hgs_binning_carve of a launch; the pass's choice is parked in status word HGS_ST_LAZY for its backward.
  const float4* grec;
  int lazy;
};
// Which passes are lazy (hgs_set_lazy_records(-1), the default): those whose capacity (or exact instance count) is at least this
// many entries per tile (at 64 the whole loop of tools/soak.py, whose model passes through 50-200, was 1 % slower than packed; at
// 128 the strand workloads below config 5's size stay packed).  Two thirds of a dense Stage-I frame's records were written and
// never read (they lie behind their tile's last contributor), and a long list hides the template gather behind the previous batch's arithmetic: config 4 +2.0 %,
// stage1_1080p +1.9 %, stage3_merged +1.3 %, config 5 +0.9 %, config 3 even.  A fresh strand model's tiles hold ~20 entries and
// their blend workgroups live for a few microseconds: the gather is one more dependent round trip in them (north_star, everything
// lazy: sort 12.0 -> 9.7 us, blend forward 24.4 -> 26.8, backward 48.2 -> 49.3: -0.7 %), so they keep the packed records.
// (Also measured: records packed for every list's first 64 entries only -- slower than either form at north_star.)
#ifndef HGS_LAZY_MIN_MEAN_LIST
#define HGS_LAZY_MIN_MEAN_LIST 128
#endif

// status words of the image buffer (HGS_IMG_STATUS)
// ([4..7] are read as ONE 16-byte scalar load by every blend workgroup)
enum { HGS_ST_R = 0, HGS_ST_OVERFLOW = 1, HGS_ST_SCANPTR_LO = 2, HGS_ST_SCANPTR_HI = 3, HGS_ST_SORT_ITEMS = 4,
       HGS_ST_SPLIT_ITEMS = 5, HGS_ST_SEG_LEN = 6, HGS_ST_WORK_ITEMS = 7, HGS_ST_TIMEOUT = 8, HGS_ST_SCAN_DONE = 9,
       HGS_ST_WL_TICKET = 10, HGS_ST_WL_NCAND = 11, HGS_ST_WL_NSEG = 12,    // exchange of the sort kernel's work-list builders
       HGS_ST_ALLOC = 13,                                                     // allocation cursor of the tiles' segments (capacity mode, scatter_kernel)
       HGS_ST_LAZY = 14 };                                                    // != 0: this pass's records are built by the blend kernels (HgsBinning::lazy; written by the sort kernel)
#ifndef HGS_SCAN_WGS
#define HGS_SCAN_WGS 32  // workgroups of the scatter kernel that allocate the tiles' segments: one tile per thread up to HGS_FUSED_SCAN_MAX_T
                         // (round 5; rounds 3-4: four workgroups sharing an exclusive scan in tile order)
#endif
#ifndef HGS_WL_BUILDERS
#define HGS_WL_BUILDERS 8        // work-list builder workgroups of the sort kernel
#endif
#define HGS_WL_BUCKETS 512       // list-length buckets of the blend work list's order
#define HGS_WL_MAX_CAND 4096     // long lists a frame can have split
#define HGS_WL_EXCHANGE_WORDS (HGS_WL_BUILDERS * 2 * HGS_WL_BUCKETS + HGS_WL_MAX_CAND)
#ifndef HGS_SORT_CAP
#define HGS_SORT_CAP 512         // keys of one sort chunk = one workgroup (measured on the Stage-I workload: 2048 -> 65 us, 1024 -> 39 us, 512 -> 29 us for the sort kernel; no effect where lists are short)
#endif
#define HGS_MAX_PARTS 63         // chunks / segments of one tile list that cooperate (bits 0..62 of the progress masks)
#define HGS_PART_FLAG (1ull << 63)
#define HGS_ITEM_NONE 0xFFFFFFFFu
#define HGS_ITEM_TILE(it) ((it) & 0xFFFFFFu)
#define HGS_ITEM_PART(it) ((it) >> 24)
#define HGS_MAX_TILES (1u << 24)
#ifndef HGS_SEG_MIN_LEN
#define HGS_SEG_MIN_LEN 128      // shortest segment of a split list (multiple of 64)
#endif
#ifndef HGS_SPLIT_PER_TILE
#define HGS_SPLIT_PER_TILE 1     // segment work items a frame can hold, per tile of the image
#endif
static inline uint32_t hgs_seg_capacity(size_t R) { return R ? (uint32_t)(R / (HGS_SEG_MIN_LEN / 2) + 2) : 0u; }   // segments >= MIN_LEN entries of lists > 1.5 MIN_LEN: < 2 R / MIN_LEN of them

static inline size_t hgs_align_up(size_t v) { return (v + HGS_ALIGN - 1) & ~(size_t)(HGS_ALIGN - 1); }

template <typename T>
static inline void hgs_carve(char*& cur, T*& ptr, size_t count) {
  cur = (char*)hgs_align_up((size_t)cur);
  ptr = (T*)cur;
  cur += count * sizeof(T);
}

static inline size_t hgs_geom_carve(char* base, size_t P, HgsGeom& g, size_t* offs) {
  char* cur = base;
  size_t nblk = (P + HGS_BLOCK - 1) / HGS_BLOCK;
  hgs_carve(cur, g.depths, P);              if (offs) offs[HGS_GEOM_DEPTHS] = (char*)g.depths - base;
  hgs_carve(cur, g.clamped, 3 * P);         if (offs) offs[HGS_GEOM_CLAMPED] = (char*)g.clamped - base;
  hgs_carve(cur, g.means2D, P);             if (offs) offs[HGS_GEOM_MEANS2D] = (char*)g.means2D - base;
  hgs_carve(cur, g.cov3D, 6 * P);           if (offs) offs[HGS_GEOM_COV3D] = (char*)g.cov3D - base;
  hgs_carve(cur, g.conic_opacity, P);       if (offs) offs[HGS_GEOM_CONIC_OPACITY] = (char*)g.conic_opacity - base;
  hgs_carve(cur, g.rgb, 3 * P);             if (offs) offs[HGS_GEOM_RGB] = (char*)g.rgb - base;
  hgs_carve(cur, g.tiles_touched, P);       if (offs) offs[HGS_GEOM_TILES_TOUCHED] = (char*)g.tiles_touched - base;
  hgs_carve(cur, g.point_offsets, P);       if (offs) offs[HGS_GEOM_POINT_OFFSETS] = (char*)g.point_offsets - base;
  hgs_carve(cur, g.rect, P);                if (offs) offs[HGS_GEOM_RECT] = (char*)g.rect - base;
  hgs_carve(cur, g.block_sums, nblk + 1);   if (offs) offs[HGS_GEOM_BLOCK_SUMS] = (char*)g.block_sums - base;
  hgs_carve(cur, g.grec, 4 * P);
  return hgs_align_up((size_t)(cur - base)) + HGS_ALIGN;
}
#define HGS_SPLIT_CAPACITY(T) ((size_t)HGS_SPLIT_PER_TILE * ((size_t)(T) > 1024 ? (size_t)(T) : (size_t)1024))   // segment work items a frame can hold
// The per-tile instance counters (tile_count) and segment cursors (tile_cursor) take one atomic per (workgroup, tile) of the
// binning kernels.  A hair frame concentrates them: the tiles of the dense region are neighbours, sixteen of them share a
// 64-byte line, and the memory side serialises the read-modify-writes of a line -- half of preprocess_fwd_kernel's time at
// every size (11.4 -> 6.3 us at 100 k Gaussians, 68 -> 32 us at 1 M with the publish removed).  Tile t's counter therefore
// lives in slot (t * odd constant) mod 2^k of a power-of-two table: neighbours land on different lines, the hot tiles
// spread over all of them.
static inline size_t hgs_tile_slots(size_t T) { size_t n = 64; while (n < T) n <<= 1; return n; }
#define HGS_TILE_SLOT(t, mask) ((uint32_t)((uint32_t)(t) * 0x9E3779B1u) & (uint32_t)(mask))
static inline size_t hgs_tile_delta_words(size_t T) { return (T + 2) & ~(size_t)1; }   // T + 1 entries, even (tile_prog stays 8-byte aligned)
static inline size_t hgs_image_zero_words(size_t T) { return 2 * hgs_tile_slots(T) + hgs_tile_delta_words(T) + 2 * T + HGS_STATUS_WORDS + 4 * T; }
static inline size_t hgs_image_carve(char* base, size_t W, size_t H, HgsImage& im, size_t* offs) {
  char* cur = base;
  size_t N = W * H, T = ((W + HGS_TILE - 1) / HGS_TILE) * ((H + HGS_TILE - 1) / HGS_TILE);
  hgs_carve(cur, im.final_T, N);            if (offs) offs[HGS_IMG_FINAL_T] = (char*)im.final_T - base;
  hgs_carve(cur, im.n_contrib, N);          if (offs) offs[HGS_IMG_N_CONTRIB] = (char*)im.n_contrib - base;
  hgs_carve(cur, im.ranges, T);             if (offs) offs[HGS_IMG_RANGES] = (char*)im.ranges - base;
  // the next seven are zeroed together by one fill in hgs_forward_preprocess (HGS_IMG_ZERO_WORDS)
  const size_t Tp = hgs_tile_slots(T);
  im.tile_mask = (uint32_t)(Tp - 1);
  hgs_carve(cur, im.tile_count, T);         if (offs) offs[HGS_IMG_TILE_COUNT] = (char*)im.tile_count - base;   // (Tp slots: below)
  // (tile_delta sits between the counters and the cursors: like tile_count it is left at zero by its last reader, not by the
  // prologue rider, whose range starts at tile_cursor -- the rider runs beside the workgroups that add to both)
  im.tile_delta = (int32_t*)(im.tile_count + Tp);
  im.tile_cursor = im.tile_count + Tp + hgs_tile_delta_words(T);  if (offs) offs[HGS_IMG_TILE_CURSOR] = (char*)im.tile_cursor - base;
  im.tile_maxc = im.tile_cursor + Tp;       if (offs) offs[HGS_IMG_TILE_MAXC] = (char*)im.tile_maxc - base;
  im.tile_done = im.tile_maxc + T;
  im.status = im.tile_done + T;             if (offs) offs[HGS_IMG_STATUS] = (char*)im.status - base;
  im.tile_prog = (unsigned long long*)(im.status + HGS_STATUS_WORDS);   // 8-byte aligned: an even number of words past a 256-B boundary
  im.tile_sortprog = im.tile_prog + T;
  cur += hgs_image_zero_words(T) * sizeof(uint32_t);
  hgs_carve(cur, im.sort_items, T);
  // the blend kernels' work list (sort_tiles_kernel): workgroup -> tile | segment << 24; segments of split lists first,
  // then the other tiles in descending order of list length.  T + HGS_SPLIT_CAPACITY(T) entries.
  hgs_carve(cur, im.tile_order, T + HGS_SPLIT_CAPACITY(T));  if (offs) offs[HGS_IMG_TILE_ORDER] = (char*)im.tile_order - base;
  hgs_carve(cur, im.wl_exchange, HGS_WL_EXCHANGE_WORDS);    // what the work-list builders of the sort kernel tell each other
  return hgs_align_up((size_t)(cur - base)) + HGS_ALIGN;
}
static inline size_t hgs_binning_carve(char* base, size_t R, HgsBinning& b, size_t* offs, int channels = 3) {
  char* cur = base;
  hgs_carve(cur, b.keys, R);                                  if (offs) offs[HGS_BIN_KEYS] = (char*)b.keys - base;
  hgs_carve(cur, b.point_list, R);                            if (offs) offs[HGS_BIN_POINT_LIST] = (char*)b.point_list - base;
  hgs_carve(cur, b.packed, R * (channels <= 3 ? HGS_PACKED_FLOATS / 4 : 4) + 4);  if (offs) offs[HGS_BIN_PACKED] = (char*)b.packed - base;
  hgs_carve(cur, b.inv, R);                                   if (offs) offs[HGS_BIN_INV] = (char*)b.inv - base;
  hgs_carve(cur, b.keys_sorted, R);                           if (offs) offs[HGS_BIN_KEYS_TMP] = (char*)b.keys_sorted - base;
  b.seg_cap = hgs_seg_capacity(R);
  hgs_carve(cur, b.seg_P, (size_t)b.seg_cap * 256);
  hgs_carve(cur, b.seg_T, (size_t)b.seg_cap * 256);
  hgs_carve(cur, b.seg_Tout, (size_t)b.seg_cap * 256);
  hgs_carve(cur, b.seg_last, (size_t)b.seg_cap * 256);
  hgs_carve(cur, b.seg_C, (size_t)b.seg_cap * 256 * (channels <= 3 ? 3 : 7));
  return hgs_align_up((size_t)(cur - base)) + HGS_ALIGN;
}

// ---- error plumbing -------------------------------------------------------------------------
void hgs_set_error(const char* fmt, ...);
#define HGS_CHECK_HIP(expr)                                                                       \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess) {                                                                       \
      hgs_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);   \
      return 1;                                                                                   \
    }                                                                                             \
  } while (0)
#define HGS_CHECK_LAUNCH() HGS_CHECK_HIP(hipGetLastError())

// Zero-fill by a kernel instead of hipMemsetAsync: memset NODES of a captured HIP graph did not re-execute reliably
// on replay (ROCm 7.2: stale tile counters on the second replay -> out-of-bounds); kernel nodes do.
int hgs_zero_async(hipStream_t s, void* ptr, size_t bytes);

// ---- optional per-kernel timing (hgs_api.hip) ---------------------------------------------------
void hgs_prof_begin(hipStream_t s, int kernel_id);
void hgs_prof_end(hipStream_t s);
struct HgsProfScope {
  hipStream_t s;
  HgsProfScope(hipStream_t st, int id) : s(st) { hgs_prof_begin(st, id); }
  ~HgsProfScope() { hgs_prof_end(s); }
};

// ---- launchers implemented in the kernel translation units -------------------------------------
int hgs_launch_smooth_fwd(hipStream_t s, int N, const float* endpoints, const long long* index_pairs, float cos_th, float eps,
                          float* partials);
int hgs_launch_smooth_bwd(hipStream_t s, int N, const float* endpoints, const long long* index_pairs, float cos_th, float eps,
                          const float* g_loss, const float* count, const float* go, float* d_endpoints);

// ---- quadrant masks ---------------------------------------------------------------------------------------------------
// Instance key = depth_bits << 32 | gaussian_id << 4 | quadrant mask (ordering by depth, then id, as the reference's stable
// sort by depth: the mask sits below the id).  Bit w of the mask is set when wavefront w's 8x8 pixel quadrant of the tile
// may hold a pixel that blends the Gaussian; the blend kernels cull on it on the scalar unit.
#define HGS_QMASK_SHIFT 4
#define HGS_QMASK_BITS 0xFu
#define HGS_MAX_GAUSSIANS (1u << 28)
#if defined(__HIPCC__)
struct HgsQuadCull { float tau, hx, hy, nx, ny, rn; int mode; };   // mode 0: never blended, 1: test, 2: keep all (degenerate)
// A pixel can only blend the Gaussian if opacity * exp(power) >= 1/255 (forward.cu:358), i.e. q(d) = 0.5 d^T Q d <= tau =
// ln(255 opacity) (Q = conic): an ellipse whose extent along a unit direction n is sqrt(2 tau n^T cov n), cov = Q^-1.
// Two separating axes are tested per quadrant: the image axes (the ellipse's bounding box) and the ellipse's minor axis,
// which is what culls a thin diagonal strand Gaussian from the quadrants its bounding box covers but it never enters.
// Extents are inflated by 0.05% + 0.01 px, orders of magnitude above any rounding of the per-pixel test; the minor-axis
// variance is evaluated for the direction actually used (any unit n gives a valid test) plus its own rounding bound.
__device__ __forceinline__ HgsQuadCull hgs_quad_cull(const float4 co) {
  HgsQuadCull c = {0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 2};
  c.tau = logf(255.f * co.w);
  const float det = co.x * co.z - co.y * co.y;
  if (!(c.tau > 0.f)) { c.mode = 0; return c; }
  if (!(det > 0.f)) return c;
  const float idet = 1.f / det;
  const float cxx = co.z * idet, cyy = co.x * idet, cxy = -co.y * idet;
  c.hx = sqrtf(2.f * c.tau * cxx) * 1.0005f + 0.01f;
  c.hy = sqrtf(2.f * c.tau * cyy) * 1.0005f + 0.01f;
  if (!(c.hx == c.hx && c.hy == c.hy)) return c;
  // minor axis: eigenvector of the smaller eigenvalue, taken from the better conditioned of the two rows
  const float hd = 0.5f * (cxx - cyy), rad = sqrtf(hd * hd + cxy * cxy);
  const float lmin = 0.5f * (cxx + cyy) - rad;
  float vx = cxy, vy = lmin - cxx;                 // (cxx - l) vx + cxy vy = 0
  if (cxx < cyy) { vx = lmin - cyy; vy = cxy; }    // cxy vx + (cyy - l) vy = 0
  const float vn = sqrtf(vx * vx + vy * vy);
  if (vn > 1e-12f * (cxx + cyy)) { c.nx = vx / vn; c.ny = vy / vn; } else { c.nx = 1.f; c.ny = 0.f; }
  const float t0 = cxx * c.nx * c.nx, t1 = cyy * c.ny * c.ny, t2 = 2.f * cxy * c.nx * c.ny;
  const float var_n = (t0 + t1 + t2) + 1e-6f * (t0 + t1 + fabsf(t2));
  c.rn = sqrtf(2.f * c.tau * fmaxf(var_n, 0.f)) * 1.0005f + 0.01f;
  c.mode = (c.rn == c.rn) ? 1 : 2;
  return c;
}
__device__ __forceinline__ uint32_t hgs_quadrant_mask(const HgsQuadCull& c, const float2 xy, int tx, int ty) {
  if (c.mode != 1) return c.mode == 0 ? 0u : HGS_QMASK_BITS;
  uint32_t qmask = 0u;
  const float x0 = (float)(tx * HGS_TILE), y0 = (float)(ty * HGS_TILE);
  const float reach = c.rn + 3.5f * (fabsf(c.nx) + fabsf(c.ny));   // the quadrant's 7x7 px rectangle of pixel centres, projected
#pragma unroll
  for (int w = 0; w < 4; w++) {
    const float qx0 = x0 + (float)((w & 1) * 8), qy0 = y0 + (float)((w >> 1) * 8);
    const bool ox = xy.x + c.hx >= qx0 && xy.x - c.hx <= qx0 + 7.f;
    const bool oy = xy.y + c.hy >= qy0 && xy.y - c.hy <= qy0 + 7.f;
    const float along = c.nx * (qx0 + 3.5f - xy.x) + c.ny * (qy0 + 3.5f - xy.y);
    if (ox && oy && fabsf(along) <= reach) qmask |= 1u << w;
  }
  return qmask;
}
#endif

struct HgsFwdArgs {
  int P, D, M, W, H;
  const float *means3D, *shs, *colors_precomp, *opacities, *scales, *rotations, *cov3D_precomp;
  const float *viewmatrix, *projmatrix, *campos;
  float scale_modifier, tan_fovx, tan_fovy;
  int prefiltered;
  int tile_cull;       // shrink every tile rectangle to the alpha >= 1/255 ellipse's bounding box (hgs_set_tile_cull)
  int row_runs;        // HGS_COUNT_ROW_RUNS: large rectangles leave two marks per tile row in im.tile_delta; the pass's scan (scan_kernel, or
                       // the scan workgroups of scatter_kernel, told through bit 0 of the parked pointer) adds their running sum
  // != 0: no scan launch follows; the scatter kernel scans the tile counts itself (see scatter_kernel) and reports the
  // instance count through this device pointer (the caller's max_rendered).  Parked in status[2..3] of the image buffer.
  unsigned long long fused_scan_ptr;
};
#ifndef HGS_FUSED_SCAN_MAX_P
#define HGS_FUSED_SCAN_MAX_P 0x7FFFFFFF   // Gaussians up to which the scatter kernel's scan workgroup replaces the scan kernel: no limit since ONE workgroup scans (with every workgroup scanning for itself the limit was 150 k).  Same box, scan workgroup against scan kernel: +0.6 % at 200 k, +1.0 % at 500 k, +0.2 % at 1 M Gaussians (the scan LDS of every workgroup costs the big launches what the scan kernel cost)
#endif
#define HGS_FUSED_SCAN_MAX_T 8192    // tiles the scatter kernel's scan workgroups take (a share's offsets live in the block's tile table); 1080p has 8160
int hgs_launch_preprocess_fwd(hipStream_t s, const HgsFwdArgs& a, const HgsGeom& g, const HgsImage& im, int* radii);
// strand parameters -> Gaussians -> preprocess in one launch (hgs_hair_forward_preprocess), riders of `fusion` included
int hgs_launch_hair_preprocess_fwd(hipStream_t s, const HgsFwdArgs& a, const HgsGeom& g, const HgsImage& im, int* radii,
                                   const float* endpoints, const long long* pairs, const float* width, float f,
                                   const float* opacity_raw, const float* mask_raw, float* xyz, float* scale, float* quat,
                                   float* opacity, float* extra4, const HgsStrandFusion& fusion);
int hgs_launch_cloud_preprocess_fwd(hipStream_t s, const HgsFwdArgs& a, const HgsGeom& g, const HgsImage& im, int* radii,
                                    const float* scaling_raw, const float* rotation_raw, const float* opacity_raw,
                                    const float* mask_raw, float* scale, float* quat, float* opacity, float* extra4,
                                    const HgsPrologue& pro);
int hgs_launch_scan(hipStream_t s, int P, int T, const HgsGeom& g, const HgsImage& im, unsigned int* max_rendered, int row_runs = 0);
int hgs_launch_scatter(hipStream_t s, int P, int W, int H, int Rcap, const float* features, const float* extra, int n_extra,
                       const HgsGeom& g, const HgsImage& im, const HgsBinning& b);
int hgs_launch_sort_tiles(hipStream_t s, int W, int H, int Rcap, int n_extra, const HgsGeom& g, const HgsImage& im,
                          const HgsBinning& b);
int hgs_launch_blend_fwd(hipStream_t s, int W, int H, int Rcap, int channels, const float* bg, const HgsImage& im,
                         const HgsBinning& b, float* out_color);
int hgs_launch_blend_bwd(hipStream_t s, int W, int H, int Rcap, int channels, const float* bg, const HgsImage& im,
                         const HgsBinning& b, const float* const* dL_dpix_planes, float* inst_grad, int tag_rows = 0);
struct HgsBwdArgs {
  int P, D, M, W, H;
  const float *means3D, *shs, *colors_precomp, *scales, *rotations, *cov3D_precomp;
  const float *viewmatrix, *projmatrix, *campos;
  float scale_modifier, tan_fovx, tan_fovy;
  const int* radii;
  float *dL_dmeans2D, *dL_dconic, *dL_dopacity, *dL_dcolors, *dL_dmeans3D, *dL_dcov3D, *dL_dsh, *dL_dscales,
      *dL_drotations;
  int n_extra;         // 0, or 4 in the single-pass mode
  float* dL_dextra;    // [P, n_extra]
  // row_reduce_kernel's results (hgs_launch_row_reduce ran in front; 7-channel pass only), or null: the kernel sums the rows itself
  const float* row_sums;       // [P][16]: the summed rows of the Gaussians whose rows lie inside one run of HGS_RR_RPW rows
  const float* row_partials;   // [runs][2][16]: per run, the sums of the segment that came in (`first`) and of the one left open (`last`)
};
#define HGS_RR_RPW 512         // rows per run (one wavefront) of row_reduce_kernel
// the scratch of the 7-channel backward behind the R instance rows: [P][16] floats + [ceil(R / HGS_RR_RPW)][2][16] floats
static inline size_t hgs_row_reduce_floats(size_t P, size_t R) { return P * 16 + ((R + HGS_RR_RPW - 1) / HGS_RR_RPW) * 32; }
int hgs_launch_row_reduce(hipStream_t s, int P, int Rcap, const float* inst_grad, const uint32_t* status, float* row_sums,
                          float* row_partials);
int hgs_launch_preprocess_bwd(hipStream_t s, const HgsBwdArgs& a, const HgsGeom& g, const HgsBinning& b,
                              const float* inst_grad, int Rcap, const uint32_t* status, const HgsParamBackward* pb = nullptr);
int hgs_launch_mark_visible(hipStream_t s, int P, const float* means3D, const float* viewmatrix, uint8_t* present);
int hgs_launch_dist2(hipStream_t s, int P, const float* points, float* out, void* scratch, size_t scratch_bytes);
size_t hgs_dist2_scratch(int P);

// ---- device helpers ----------------------------------------------------------------------------
#ifdef __HIPCC__
// Agent-scope traffic between workgroups of ONE launch (the eight XCDs have private L2s): relaxed agent-scope atomic
// loads / stores compile to sc1 accesses that go through to the memory side, no cache-wide write-back or invalidate.
// A producer stores its data with hgs_st_agent, drains its stores (hgs_drain_stores + barrier) and then sets its bit in
// the tile's progress mask; a consumer polls the mask (bounded: a timeout raises HGS_ST_TIMEOUT instead of hanging
// the GPU) and reads with hgs_ld_agent.  Waiting is only ever on workgroups within HGS_MAX_PARTS positions of the
// waiter in dispatch order, so the workgroups waited for are resident or finished.
template <typename T> __device__ __forceinline__ void hgs_st_agent(T* p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T> __device__ __forceinline__ T hgs_ld_agent(const T* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void hgs_drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// A pointer read from a struct in memory (HgsViewTargets) has no known address space: the compiler emits FLAT loads for
// it, which count in lgkmcnt as well as vmcnt -- every LDS wait behind such a load then also waits for the HBM round trip
// (measured in the SSIM kernels: the row pass behind a prefetch of the target image took 2.2 us instead of 0.7).  These
// are device-memory pointers by contract (include/hgs.h): say so, and keep the address space up to the load (a cast back to
// a generic pointer is folded away together with the information).
#define HGS_GLOBAL __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ const HGS_GLOBAL T* hgs_global(const T* p) { return (const HGS_GLOBAL T*)p; }
typedef float hgs_float4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 hgs_load4(const HGS_GLOBAL float* p) {      // 16-byte aligned
  return __builtin_bit_cast(float4, *(const HGS_GLOBAL hgs_float4_t*)p);   // (no component-wise copy: the load's registers are the result)
}
__device__ __forceinline__ void hgs_publish_part(unsigned long long* mask, uint32_t part) {
  __hip_atomic_fetch_or(mask, 1ull << part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool hgs_wait_parts(const unsigned long long* mask, unsigned long long want, uint32_t* status) {
#ifndef HGS_WAIT_BACKOFF
#define HGS_WAIT_BACKOFF 0     // experiment (round 5): 1 = the pause between two polls doubles up to ~3 us
#endif
  for (int spin = 0; spin < (1 << 21); spin++) {
    if ((hgs_ld_agent(mask) & want) == want) return true;
#if HGS_WAIT_BACKOFF
    // (s_sleep takes an immediate: 64 x N cycles)
    if (spin < 4) __builtin_amdgcn_s_sleep(8);
    else if (spin < 8) __builtin_amdgcn_s_sleep(16);
    else if (spin < 12) __builtin_amdgcn_s_sleep(32);
    else if (spin < 16) __builtin_amdgcn_s_sleep(64);
    else __builtin_amdgcn_s_sleep(127);
#else
    __builtin_amdgcn_s_sleep(8);
#endif
  }
  status[HGS_ST_TIMEOUT] = 1u;
  // capacity mode: the caller's sticky instance-count maximum (its pointer is parked in the status words) is raised to
  // 0xFFFFFFFF, which the host's next validation cannot miss (include/hgs.h HGS_WAIT_TIMED_OUT) -- a frame blended from
  // unfinished segments must not pass silently through any number of graph replays
  const unsigned long long report = (((unsigned long long)status[HGS_ST_SCANPTR_HI] << 32) | status[HGS_ST_SCANPTR_LO]) & ~1ull;   // (bit 0: HGS_COUNT_ROW_RUNS)
  if (report) atomicMax((unsigned int*)report, 0xFFFFFFFFu);
  return false;
}
// segment length of the split blend for a pass with R instances: long lists are cut so that the pass has on the order
// of `target` segments, within [min_len, max_len] entries (multiples of the 64-entry staging batch); hgs_set_segment_policy
struct HgsSegPolicy { uint32_t min_len, max_len, target; };
__device__ __forceinline__ uint32_t hgs_segment_length(uint32_t R, const HgsSegPolicy& p) {
  const uint32_t s = ((R / p.target) + 63u) & ~63u;
  return s < p.min_len ? p.min_len : (s > p.max_len ? p.max_len : s);
}
#ifndef HGS_SPLIT_QUARTERS
#define HGS_SPLIT_QUARTERS 6u   // a list is split when it is longer than this many quarters of the segment length
#endif
struct HgsSplit { uint32_t nseg, seglen; };   // nseg == 1: the list is walked by one workgroup
__device__ __forceinline__ HgsSplit hgs_split_of(uint32_t n, uint32_t S) {
  if (n <= S * HGS_SPLIT_QUARTERS / 4u) return {1u, n};
  uint32_t seglen = S, nseg = (n + S - 1) / S;
  if (nseg > HGS_MAX_PARTS) { seglen = (((n + HGS_MAX_PARTS - 1) / HGS_MAX_PARTS) + 63u) & ~63u; nseg = (n + seglen - 1) / seglen; }
  return {nseg, seglen};
}
// chunk work items of a long list for the sort kernel (called where the tile counts are scanned)
__device__ __forceinline__ void hgs_emit_sort_items(uint32_t tile, uint32_t n, uint32_t T, const HgsImage& im) {
  if (n <= HGS_SORT_CAP) return;
  const uint32_t nch = (n + HGS_SORT_CAP - 1) / HGS_SORT_CAP;
  if (nch > HGS_MAX_PARTS) return;                       // the tile's own workgroup sorts it chunk after chunk
  const uint32_t base = atomicAdd(&im.status[HGS_ST_SORT_ITEMS], nch);
  const bool fits = base + nch <= T;
  for (uint32_t c = 0; c < nch && base + c < T; c++) im.sort_items[base + c] = fits ? (tile | (c << 24)) : HGS_ITEM_NONE;
  if (fits) im.tile_sortprog[tile] = HGS_PART_FLAG;
}
// float -> int with the hardware's saturating semantics made explicit (NaN -> 0).
__device__ __forceinline__ int hgs_f2i(float v) {
  if (v != v) return 0;
  v = fminf(fmaxf(v, -2147483648.f), 2147483520.f);
  return (int)v;
}
// wave64 inclusive scan / reductions on 32-bit values through ds_bpermute-free DPP-less shuffles.
__device__ __forceinline__ uint32_t hgs_wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}
#endif
