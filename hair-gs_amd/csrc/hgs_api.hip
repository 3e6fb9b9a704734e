// hgs_api.hip -- the extern "C" surface declared in include/hgs.h (host code only).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "hgs_common.h"
#include "hgs_prologue.h"

static thread_local char g_err[512] = "";

void hgs_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- per-kernel timing -------------------------------------------------------------------------
#include <vector>
namespace {
struct ProfRec { hipEvent_t a, b; int id; };
bool g_prof_on = false;
std::vector<ProfRec> g_prof_log;
std::vector<hipEvent_t> g_prof_pool;
hipEvent_t g_prof_open = nullptr;
int g_prof_open_id = -1;
const char* kKernelNames[HGS_K_COUNT] = {"preprocess_fwd_kernel", "scan_kernel", "scatter_kernel", "sort_tiles_kernel",
                                         "blend_fwd_kernel", "blend_bwd_kernel", "preprocess_bwd_kernel", "dist2_kernels",
                                         "ssim_l1_fwd_kernel", "ssim_l1_bwd_kernel", "strand_fwd_kernel",
                                         "strand_bwd_kernel", "ori_fwd_kernel", "ori_bwd_kernel", "adam_kernel",
                                         "smooth_kernels", "head_tail_kernel", "misc_kernels"};
hipEvent_t prof_event() {
  if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace
void hgs_prof_begin(hipStream_t s, int kernel_id) {
  if (!g_prof_on) return;
  g_prof_open = prof_event();
  g_prof_open_id = kernel_id;
  (void)hipEventRecord(g_prof_open, s);
}
void hgs_prof_end(hipStream_t s) {
  if (!g_prof_on || !g_prof_open) return;
  hipEvent_t b = prof_event();
  (void)hipEventRecord(b, s);
  g_prof_log.push_back({g_prof_open, b, g_prof_open_id});
  g_prof_open = nullptr;
}

__global__ __launch_bounds__(256) void hgs_zero_kernel(uint32_t* __restrict__ p, size_t n_words) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_words; i += stride) p[i] = 0u;
}
int hgs_zero_async(hipStream_t s, void* ptr, size_t bytes) {
  if (bytes == 0) return 0;
  if (((size_t)ptr & 3) || (bytes & 3)) { hgs_set_error("hgs_zero_async: pointer/size must be 4-byte multiples"); return 1; }
  const size_t n = bytes / 4;
  const unsigned blocks = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(hgs_zero_kernel, dim3(blocks), dim3(256), 0, s, (uint32_t*)ptr, n);
  HGS_CHECK_LAUNCH();
  return 0;
}

// ---- small per-iteration bookkeeping kernels --------------------------------------------------------------------
// Iteration prologue: workgroup 0 copies the view's row into the slot (and the learning rate); the others clear
// zero_words words at zero_ptr (the counters of the image buffer the coming forward pass bins into, include/hgs.h
// HGS_IMAGE_PREZEROED).  view / lr are by-value arguments: a captured graph is re-pointed at another view by updating this
// node's parameters (hgs_graph_set_prologue), with no launch in between two replays.
__global__ __launch_bounds__(256) void select_view_kernel(HgsPrologue p) { hgs_prologue_block(p, blockIdx.x, gridDim.x); }

struct HgsViewQueueArgs { int v[HGS_VIEW_QUEUE_MAX]; };
__global__ void set_view_queue_kernel(int* __restrict__ queue, int n, HgsViewQueueArgs a, float lr, float* __restrict__ lr_slot) {
  if ((int)threadIdx.x < n) queue[threadIdx.x] = a.v[threadIdx.x];
  if (threadIdx.x == 0 && lr_slot) *lr_slot = lr;
}
__global__ void select_view_queued_kernel(const HgsViewTargets* __restrict__ table, int n_views,
                                          const int* __restrict__ view_index, HgsViewTargets* __restrict__ slot,
                                          const float* __restrict__ lr_slot, float* __restrict__ lr_dst) {
  int view = *view_index;
  if (view < 0 || view >= n_views) view = 0;
  const uint32_t* src = (const uint32_t*)(table + view);
  uint32_t* dst = (uint32_t*)slot;
  for (int i = threadIdx.x; i < (int)(sizeof(HgsViewTargets) / 4); i += 64) dst[i] = src[i];
  if (threadIdx.x == 0 && lr_dst && lr_slot) *lr_dst = *lr_slot;
}

__global__ __launch_bounds__(256) void densify_stats_kernel(int P, const int* __restrict__ radii,
                                                            const float* __restrict__ g, int stride,
                                                            float* __restrict__ max_radii, float* __restrict__ accum,
                                                            float* __restrict__ denom) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const int r = radii[i];
  if (r <= 0) return;                                           // visibility_filter = radii > 0 (train.py:170)
  max_radii[i] = fmaxf(max_radii[i], (float)r);
  const float gx = g[(size_t)i * stride], gy = g[(size_t)i * stride + 1];
  accum[i] += sqrtf(gx * gx + gy * gy);                         // torch.norm(grad[:, :2], dim=-1)
  denom[i] += 1.f;
}

static int g_tile_cull = 1;
extern "C" int hgs_set_tile_cull(int on) {
  const int was = g_tile_cull;
  g_tile_cull = on != 0;
  return was;
}

static int check_aligned(const void* p, const char* what) {
  if (!p || ((size_t)p & (HGS_ALIGN - 1))) {
    hgs_set_error("%s must be a non-null %d-byte aligned device pointer", what, HGS_ALIGN);
    return 1;
  }
  return 0;
}

extern "C" {

int hgs_abi_version(void) { return HGS_ABI_VERSION; }
const char* hgs_last_error(void) { return g_err; }

size_t hgs_geom_bytes(int P) { HgsGeom g; return hgs_geom_carve(nullptr, (size_t)(P > 0 ? P : 0), g, nullptr); }
size_t hgs_image_bytes(int W, int H) { HgsImage im; return hgs_image_carve(nullptr, (size_t)W, (size_t)H, im, nullptr); }
size_t hgs_binning_bytes(int R) { HgsBinning b; return hgs_binning_carve(nullptr, (size_t)(R > 0 ? R : 0), b, nullptr); }
size_t hgs_backward_scratch_bytes(int P, int R) {
  (void)P;
  return hgs_align_up((size_t)(R > 0 ? R : 0) * HGS_INST_GRAD_FLOATS * sizeof(float)) + HGS_ALIGN;
}
int hgs_geom_layout(int P, size_t* offsets) { HgsGeom g; hgs_geom_carve(nullptr, (size_t)P, g, offsets); return 0; }
int hgs_image_zero_range(int W, int H, size_t* offset, size_t* bytes) {
  if (W <= 0 || H <= 0 || !offset || !bytes) { hgs_set_error("hgs_image_zero_range: bad arguments"); return 1; }
  HgsImage im;
  hgs_image_carve(nullptr, (size_t)W, (size_t)H, im, nullptr);
  const size_t T = (size_t)((W + HGS_TILE - 1) / HGS_TILE) * ((H + HGS_TILE - 1) / HGS_TILE);
  *offset = (size_t)((char*)im.tile_count - (char*)nullptr);
  *bytes = hgs_image_zero_words(T) * sizeof(uint32_t);
  return 0;
}
int hgs_image_layout(int W, int H, size_t* offsets) { HgsImage im; hgs_image_carve(nullptr, (size_t)W, (size_t)H, im, offsets); return 0; }
int hgs_binning_layout(int R, size_t* offsets) { HgsBinning b; hgs_binning_carve(nullptr, (size_t)R, b, offsets); return 0; }

static_assert(HGS_FUSED_PREPROCESS_MAX_TILES == HGS_FUSED_SCAN_MAX_T, "include/hgs.h states the scatter kernel's scan limit");
// the strand parameters of hgs_hair_forward_preprocess (NULL: the Gaussians are given)
struct HairSrc {
  const float* endpoints; const long long* pairs; const float* width; float f; const float* opacity_raw; const float* mask_raw;
  float* xyz; float* scale; float* quat; float* opacity; float* extra4; const HgsStrandFusion* fusion;
  const float* scaling_raw; const float* rotation_raw;    // != NULL: a Stage-I cloud (hgs_cloud_forward_preprocess)
};
static int forward_preprocess_impl(void* stream, int P, int D, int M, int W, int H, const float* means3D, const float* shs,
                                   const float* colors_precomp, const float* opacities, const float* scales,
                                   float scale_modifier, const float* rotations, const float* cov3D_precomp,
                                   const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx,
                                   float tan_fovy, int prefiltered, void* geom_buf, void* image_buf, int* radii,
                                   int* num_rendered_host, unsigned int* max_rendered, const HairSrc* hair);

int hgs_forward_preprocess(void* stream, int P, int D, int M, int W, int H, const float* means3D, const float* shs,
                           const float* colors_precomp, const float* opacities, const float* scales,
                           float scale_modifier, const float* rotations, const float* cov3D_precomp,
                           const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx,
                           float tan_fovy, int prefiltered, void* geom_buf, void* image_buf, int* radii,
                           int* num_rendered_host, unsigned int* max_rendered) {
  return forward_preprocess_impl(stream, P, D, M, W, H, means3D, shs, colors_precomp, opacities, scales, scale_modifier,
                                 rotations, cov3D_precomp, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, prefiltered,
                                 geom_buf, image_buf, radii, num_rendered_host, max_rendered, nullptr);
}

int hgs_hair_forward_preprocess(void* stream, int P, int D, int M, int W, int H, const float* endpoints,
                                const long long* endpoint_pairs, const float* width, float dist_to_scale_factor,
                                const float* opacity_raw, const float* mask_raw, const float* shs, float* xyz, float* scale,
                                float* quat, float* opacity, float* extra4, const float* viewmatrix, const float* projmatrix,
                                const float* campos, float tan_fovx, float tan_fovy, int flags, void* geom_buf,
                                void* image_buf, int* radii, unsigned int* max_rendered, const HgsStrandFusion* fusion) {
  if (P > 0 && (!endpoints || !endpoint_pairs || !width || !opacity_raw || !mask_raw || !xyz || !scale || !quat || !opacity ||
                !extra4 || ((size_t)quat & 15) || ((size_t)extra4 & 15))) {
    hgs_set_error("hgs_hair_forward_preprocess: null (or, quat / extra4, not 16-byte aligned) argument");
    return 1;
  }
  if (!max_rendered) { hgs_set_error("hgs_hair_forward_preprocess: capacity mode only (max_rendered must be given)"); return 1; }
  const HairSrc hair = {endpoints, endpoint_pairs, width, dist_to_scale_factor, opacity_raw, mask_raw, xyz, scale, quat, opacity,
                        extra4, fusion, nullptr, nullptr};
  return forward_preprocess_impl(stream, P, D, M, W, H, xyz, shs, nullptr, opacity, scale, 1.f, quat, nullptr, viewmatrix,
                                 projmatrix, campos, tan_fovx, tan_fovy, flags, geom_buf, image_buf, radii, nullptr,
                                 max_rendered, &hair);
}

int hgs_cloud_forward_preprocess(void* stream, int P, int D, int M, int W, int H, const float* means3D, const float* scaling_raw,
                                 const float* rotation_raw, const float* opacity_raw, const float* mask_raw, const float* shs,
                                 float* scale, float* quat, float* opacity, float* extra4, const float* viewmatrix,
                                 const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy, int flags,
                                 void* geom_buf, void* image_buf, int* radii, unsigned int* max_rendered,
                                 const HgsStrandFusion* fusion) {
  if (P > 0 && (!means3D || !scaling_raw || !rotation_raw || !opacity_raw || !mask_raw || !scale || !quat || !opacity || !extra4 ||
                ((size_t)rotation_raw & 15) || ((size_t)quat & 15) || ((size_t)extra4 & 15))) {
    hgs_set_error("hgs_cloud_forward_preprocess: null (or, rotation_raw / quat / extra4, not 16-byte aligned) argument");
    return 1;
  }
  if (!max_rendered) { hgs_set_error("hgs_cloud_forward_preprocess: capacity mode only (max_rendered must be given)"); return 1; }
  const HairSrc cloud = {nullptr, nullptr, nullptr, 0.f, opacity_raw, mask_raw, nullptr, scale, quat, opacity, extra4, fusion,
                         scaling_raw, rotation_raw};
  return forward_preprocess_impl(stream, P, D, M, W, H, means3D, shs, nullptr, opacity, scale, 1.f, quat, nullptr, viewmatrix,
                                 projmatrix, campos, tan_fovx, tan_fovy, flags, geom_buf, image_buf, radii, nullptr,
                                 max_rendered, &cloud);
}

static int forward_preprocess_impl(void* stream, int P, int D, int M, int W, int H, const float* means3D, const float* shs,
                                   const float* colors_precomp, const float* opacities, const float* scales,
                                   float scale_modifier, const float* rotations, const float* cov3D_precomp,
                                   const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx,
                                   float tan_fovy, int prefiltered, void* geom_buf, void* image_buf, int* radii,
                                   int* num_rendered_host, unsigned int* max_rendered, const HairSrc* hair) {
  hipStream_t s = (hipStream_t)stream;
  if (P < 0 || W <= 0 || H <= 0) { hgs_set_error("bad sizes P=%d W=%d H=%d", P, W, H); return 1; }
  if ((size_t)((W + HGS_TILE - 1) / HGS_TILE) * ((H + HGS_TILE - 1) / HGS_TILE) > HGS_MAX_TILES) { hgs_set_error("%dx%d: more than 2^24 tiles", W, H); return 1; }
  if ((unsigned)P > HGS_MAX_GAUSSIANS) { hgs_set_error("P=%d: at most 2^28 Gaussians per pass (instance key layout)", P); return 1; }
  if (D < 0 || D > 3) { hgs_set_error("sh degree %d unsupported (0..3)", D); return 1; }
  if (check_aligned(image_buf, "image_buf")) return 1;
  HgsImage im;
  hgs_image_carve((char*)image_buf, (size_t)W, (size_t)H, im, nullptr);
  const int T = ((W + HGS_TILE - 1) / HGS_TILE) * ((H + HGS_TILE - 1) / HGS_TILE);
  if (!(prefiltered & HGS_IMAGE_PREZEROED) &&
      hgs_zero_async(s, im.tile_count, hgs_image_zero_words((size_t)T) * sizeof(uint32_t))) return 1;
  const HgsPrologue* rider = (hair && hair->fusion && hair->fusion->prologue.table) ? &hair->fusion->prologue : nullptr;
  if (rider && (!rider->slot || rider->view < 0 || ((size_t)rider->zero_ptr & 3) || (rider->zero_bytes & 3) ||
                (rider->zero_bytes && !rider->zero_ptr))) {
    hgs_set_error("hgs_hair/cloud_forward_preprocess: bad prologue group");
    return 1;
  }
  if (hair && T > HGS_FUSED_SCAN_MAX_T) {
    hgs_set_error("hgs_hair_forward_preprocess: %d tiles, at most %d (the tile counters are cleared by the scatter kernel's scan)", T, HGS_FUSED_SCAN_MAX_T);
    return 1;
  }
  if (P == 0 && rider &&     // nothing to ride on: the prologue as a launch of its own
      hgs_iteration_prologue(stream, rider->table, rider->view, rider->slot, rider->lr, rider->lr_dst, rider->zero_ptr, rider->zero_bytes,
                             rider->adam_prep)) return 1;
  if (P == 0) {  // reference short-circuits P == 0 (rasterize_points.cu:81); the scan of all-zero counts writes the
    HgsGeom none = {};  // empty ranges and the tile order the blend kernel (background fill) indexes with
    if (hgs_launch_scan(s, 0, T, none, im, nullptr)) return 1;
    if (num_rendered_host) { HGS_CHECK_HIP(hipStreamSynchronize(s)); *num_rendered_host = 0; }
    return 0;
  }
  if (check_aligned(geom_buf, "geom_buf")) return 1;
  if (!means3D || !opacities || !viewmatrix || !projmatrix || !campos || !radii) { hgs_set_error("null required input"); return 1; }
  if ((shs == nullptr) == (colors_precomp == nullptr)) { hgs_set_error("provide exactly one of shs / colors_precomp"); return 1; }
  if (((scales == nullptr) || (rotations == nullptr)) == (cov3D_precomp == nullptr)) {
    hgs_set_error("provide exactly one of (scales, rotations) / cov3D_precomp"); return 1;
  }
  if (shs && M < (D + 1) * (D + 1)) { hgs_set_error("M=%d SH coefficients < (D+1)^2 for D=%d", M, D); return 1; }
  HgsGeom g;
  hgs_geom_carve((char*)geom_buf, (size_t)P, g, nullptr);
  HgsFwdArgs a;
  a.P = P; a.D = D; a.M = M; a.W = W; a.H = H;
  a.means3D = means3D; a.shs = shs; a.colors_precomp = colors_precomp; a.opacities = opacities; a.scales = scales;
  a.rotations = rotations; a.cov3D_precomp = cov3D_precomp; a.viewmatrix = viewmatrix; a.projmatrix = projmatrix;
  a.campos = campos; a.scale_modifier = scale_modifier; a.tan_fovx = tan_fovx; a.tan_fovy = tan_fovy;
  a.prefiltered = prefiltered & 1;
  a.tile_cull = g_tile_cull;
  a.row_runs = (prefiltered & HGS_COUNT_ROW_RUNS) ? 1 : 0;
  // Capacity mode (nobody waits for num_rendered here) with a place to report the count: the scan is left to the
  // scatter kernel of hgs_forward_render (scatter_kernel, "fused scan").  A blocking caller needs the count NOW.
  // (Every workgroup of the scatter kernel repeats the scan: worth the saved launch only while there are few of them --
  // measured: 100 k Gaussians +0.7 % with the fused scan; 200 k / 500 k / 1 M Gaussians 3 / 4 / 17 us per pass FASTER with
  // the one-workgroup scan kernel in between.)
  const bool fused_scan = !num_rendered_host && max_rendered && T <= HGS_FUSED_SCAN_MAX_T && P <= HGS_FUSED_SCAN_MAX_P;
  a.fused_scan_ptr = fused_scan ? (unsigned long long)(size_t)max_rendered : 0ull;
  if (hair) {
    if (!fused_scan) { hgs_set_error("hgs_hair_forward_preprocess: the scan must be the scatter kernel's (sizes beyond its limits)"); return 1; }
    HgsStrandFusion fu = hair->fusion ? *hair->fusion : HgsStrandFusion{};
    if (!(fu.smooth_pairs && fu.n_smooth > 0 && fu.smooth_partials)) fu.n_smooth = 0;
    if (hair->scaling_raw) {
      if (hgs_launch_cloud_preprocess_fwd(s, a, g, im, radii, hair->scaling_raw, hair->rotation_raw, hair->opacity_raw,
                                          hair->mask_raw, hair->scale, hair->quat, hair->opacity, hair->extra4, fu.prologue)) return 1;
    } else if (hgs_launch_hair_preprocess_fwd(s, a, g, im, radii, hair->endpoints, hair->pairs, hair->width, hair->f,
                                              hair->opacity_raw, hair->mask_raw, hair->xyz, hair->scale, hair->quat,
                                              hair->opacity, hair->extra4, fu)) return 1;
  } else if (hgs_launch_preprocess_fwd(s, a, g, im, radii)) return 1;
  if (!fused_scan && hgs_launch_scan(s, P, T, g, im, max_rendered, a.row_runs)) return 1;
  if (num_rendered_host) {
    uint32_t r = 0;
    HGS_CHECK_HIP(hipMemcpyAsync(&r, im.status, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HGS_CHECK_HIP(hipStreamSynchronize(s));
    *num_rendered_host = (int)r;
  }
  return 0;
}

// How the blend kernels get their records (hgs_common.h, HgsBinning::lazy): 1 always from the templates, 0 always packed by the sort
// kernel, -1 (default) by the pass's entries per tile.
static int g_lazy_mode = -1;
extern "C" int hgs_set_lazy_records(int mode) {
  const int was = g_lazy_mode;
  g_lazy_mode = mode > 0 ? 1 : (mode < 0 ? -1 : 0);
  return was;
}

static int forward_render_impl(void* stream, int P, int W, int H, int R_capacity, const float* bg,
                               const float* colors_precomp, const float* extra, int n_extra, void* geom_buf,
                               void* binning_buf, void* image_buf, float* out_color) {
  hipStream_t s = (hipStream_t)stream;
  if (check_aligned(image_buf, "image_buf")) return 1;
  if (!bg || !out_color) { hgs_set_error("null bg/out_color"); return 1; }
  if (n_extra != 0 && n_extra != 4) { hgs_set_error("n_extra must be 0 or 4 (got %d)", n_extra); return 1; }
  if (n_extra && P > 0 && (!extra || ((size_t)extra & 15))) { hgs_set_error("extra colours must be a 16-byte aligned [P,4] array"); return 1; }
  const int channels = 3 + n_extra;
  HgsImage im;
  hgs_image_carve((char*)image_buf, (size_t)W, (size_t)H, im, nullptr);
  HgsGeom g = {};
  HgsBinning b = {};
  if (P > 0) {
    if (check_aligned(geom_buf, "geom_buf")) return 1;
    hgs_geom_carve((char*)geom_buf, (size_t)P, g, nullptr);
    if (R_capacity > 0) {
      if (check_aligned(binning_buf, "binning_buf")) return 1;
      hgs_binning_carve((char*)binning_buf, (size_t)R_capacity, b, nullptr, channels);
      b.grec = g.grec;
      const long long tiles = (long long)((W + HGS_TILE - 1) / HGS_TILE) * ((H + HGS_TILE - 1) / HGS_TILE);
      b.lazy = g_lazy_mode >= 0 ? g_lazy_mode : ((long long)R_capacity >= (long long)HGS_LAZY_MIN_MEAN_LIST * tiles ? 1 : 0);
    }
    // the scatter also finishes the per-Gaussian instance offsets, so it runs even when nothing is visible
    const float* feat = colors_precomp ? colors_precomp : g.rgb;
    if (hgs_launch_scatter(s, P, W, H, R_capacity > 0 ? R_capacity : 0, feat, extra, n_extra, g, im, b)) return 1;
  }
  // always launched: besides the per-tile sorts (no-ops on empty lists) it computes the blend kernels' work list
  if (hgs_launch_sort_tiles(s, W, H, R_capacity > 0 ? R_capacity : 0, n_extra, g, im, b)) return 1;
  return hgs_launch_blend_fwd(s, W, H, R_capacity > 0 ? R_capacity : 0, channels, bg, im, b, out_color);
}

int hgs_forward_render(void* stream, int P, int W, int H, int R_capacity, const float* bg, const float* colors_precomp,
                       void* geom_buf, void* binning_buf, void* image_buf, float* out_color) {
  return forward_render_impl(stream, P, W, H, R_capacity, bg, colors_precomp, nullptr, 0, geom_buf, binning_buf,
                             image_buf, out_color);
}

int hgs_forward_render_multi(void* stream, int P, int W, int H, int R_capacity, const float* bg7,
                             const float* colors_precomp, const float* extra4, void* geom_buf, void* binning_buf,
                             void* image_buf, float* out_color7) {
  return forward_render_impl(stream, P, W, H, R_capacity, bg7, colors_precomp, extra4, 4, geom_buf, binning_buf,
                             image_buf, out_color7);
}

size_t hgs_binning_bytes_multi(int R) { HgsBinning b; return hgs_binning_carve(nullptr, (size_t)(R > 0 ? R : 0), b, nullptr, 7); }
size_t hgs_backward_scratch_bytes_multi(int P, int R) {
  // R instance rows of 16 floats, then (256-byte aligned) row_reduce_kernel's per-Gaussian sums and per-run partial sums
  const size_t p = (size_t)(P > 0 ? P : 0), r = (size_t)(R > 0 ? R : 0);
  return hgs_align_up(r * 16 * sizeof(float)) + hgs_align_up(hgs_row_reduce_floats(p, r) * sizeof(float)) + HGS_ALIGN;
}
// hgs_set_row_reduce: whether the 7-channel backward sums the instance rows per Gaussian with a launch of its own
// (row_reduce_kernel, balanced by rows): 1 yes, 0 no, -1 (default) where the pass's R is at least HGS_RR_AUTO_RATIO x P
#define HGS_RR_AUTO_RATIO 8
static int g_row_reduce_mode = -1;
extern "C" int hgs_set_row_reduce(int mode) {
  const int was = g_row_reduce_mode;
  g_row_reduce_mode = mode > 0 ? 1 : (mode < 0 ? -1 : 0);
  return was;
}

static int backward_impl(void* stream, int P, int D, int M, int R, int W, int H, const float* bg, const float* means3D,
                         const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                         const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                         const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                         const void* geom_buf, const void* binning_buf, const void* image_buf,
                         const float* const* dL_dpix_planes, void* scratch, int n_extra, float* dL_dextra, float* dL_dmeans2D, float* dL_dconic,
                         float* dL_dopacity, float* dL_dcolors, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh,
                         float* dL_dscales, float* dL_drotations, const HgsParamBackward* pb = nullptr) {
  hipStream_t s = (hipStream_t)stream;
  if (P == 0) return 0;  // rasterize_points.cu:161
  if (check_aligned(geom_buf, "geom_buf") || check_aligned(image_buf, "image_buf")) return 1;
  // (viewmatrix / projmatrix / campos are read at kernel entry whatever the colour source: required, include/hgs.h)
  if (!dL_dpix_planes || !radii || !means3D || !viewmatrix || !projmatrix || !campos) { hgs_set_error("null required input"); return 1; }
  for (int k = 0; k < 3 + n_extra; k++)
    if (!dL_dpix_planes[k]) { hgs_set_error("null dL_dpix plane %d", k); return 1; }
  if (!pb && (!dL_dmeans2D || !dL_dconic || !dL_dopacity || !dL_dcolors || !dL_dmeans3D || !dL_dcov3D || !dL_dscales ||
              !dL_drotations || (n_extra && !dL_dextra))) { hgs_set_error("null gradient output"); return 1; }
  if (shs && !dL_dsh) { hgs_set_error("null gradient output"); return 1; }
  const int channels = 3 + n_extra;
  HgsGeom g;
  HgsImage im;
  HgsBinning b = {};
  hgs_geom_carve((char*)geom_buf, (size_t)P, g, nullptr);
  hgs_image_carve((char*)image_buf, (size_t)W, (size_t)H, im, nullptr);
  float* inst_grad = nullptr;
  // many instances per Gaussian (the states the three-stage workflow lives in: a Stage-I cloud at 1080p, the merged strand
  // model): the per-Gaussian sums of the rows are taken by a launch that is balanced by rows (csrc/hgs_preprocess.hip)
  const bool reduce_rows = n_extra && R > 0 && (g_row_reduce_mode > 0 || (g_row_reduce_mode < 0 && (long long)R >= (long long)HGS_RR_AUTO_RATIO * P));
  if (R > 0) {
    if (check_aligned(binning_buf, "binning_buf") || check_aligned(scratch, "scratch")) return 1;
    hgs_binning_carve((char*)binning_buf, (size_t)R, b, nullptr, channels);
    b.grec = g.grec;
    inst_grad = (float*)scratch;   // not cleared here: blend_bwd writes EVERY row (zeros past a tile's last needed entry)
    if (hgs_launch_blend_bwd(s, W, H, R, channels, bg, im, b, dL_dpix_planes, inst_grad, reduce_rows ? 1 : 0)) return 1;
  }
  float *row_sums = nullptr, *row_partials = nullptr;
  if (reduce_rows) {
    row_sums = (float*)((char*)scratch + hgs_align_up((size_t)R * 16 * sizeof(float)));
    row_partials = row_sums + (size_t)P * 16;
    if (hgs_launch_row_reduce(s, P, R, inst_grad, im.status, row_sums, row_partials)) return 1;
  }
  HgsBwdArgs a;
  a.P = P; a.D = D; a.M = M; a.W = W; a.H = H;
  a.means3D = means3D; a.shs = shs; a.colors_precomp = colors_precomp; a.scales = scales; a.rotations = rotations;
  a.cov3D_precomp = cov3D_precomp; a.viewmatrix = viewmatrix; a.projmatrix = projmatrix; a.campos = campos;
  a.scale_modifier = scale_modifier; a.tan_fovx = tan_fovx; a.tan_fovy = tan_fovy; a.radii = radii;
  a.dL_dmeans2D = dL_dmeans2D; a.dL_dconic = dL_dconic; a.dL_dopacity = dL_dopacity; a.dL_dcolors = dL_dcolors;
  a.dL_dmeans3D = dL_dmeans3D; a.dL_dcov3D = dL_dcov3D; a.dL_dsh = dL_dsh; a.dL_dscales = dL_dscales;
  a.dL_drotations = dL_drotations;
  a.n_extra = n_extra; a.dL_dextra = dL_dextra;
  a.row_sums = row_sums; a.row_partials = row_partials;
  return hgs_launch_preprocess_bwd(s, a, g, b, inst_grad, R, im.status, pb);
}

size_t hgs_param_backward_bytes(void) { return sizeof(HgsParamBackward); }
size_t hgs_adam_prep_bytes(void) { return sizeof(HgsAdamPrep); }
size_t hgs_adam_inline_bytes(void) { return sizeof(HgsAdamInline); }
static int check_adam_inline(const HgsAdamInline& a, int n_slots, const char* who) {
  for (int k = 0; k < 6; k++) {
    const HgsAdamSlot& sl = a.slot[k];
    if (!sl.p) continue;
    if (k >= n_slots || !sl.m || !sl.v || !sl.coef) { hgs_set_error("%s: incomplete Adam slot %d", who, k); return 1; }
  }
  return 0;
}
int hgs_backward_multi_params(void* stream, int P, int D, int M, int R, int W, int H, const float* bg7, const float* means3D,
                              const float* shs, const float* scales, const float* rotations, const float* viewmatrix,
                              const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                              const void* geom_buf, const void* binning_buf, const void* image_buf,
                              const float* const* dL_dpix_planes7, void* scratch, float* dL_dsh, const HgsParamBackward* pb) {
  if (!pb || (pb->kind != HGS_PARAMS_HAIR && pb->kind != HGS_PARAMS_CLOUD)) { hgs_set_error("hgs_backward_multi_params: params->kind must be HGS_PARAMS_HAIR or HGS_PARAMS_CLOUD"); return 1; }
  if (P == 0 && pb->head_tail.out) { hgs_set_error("hgs_backward_multi_params: a deferred loss-head tail needs a launch (P > 0)"); return 1; }
  if (P == 0) return 0;
  if (!shs || !scales || !rotations || !dL_dsh) { hgs_set_error("hgs_backward_multi_params: SH colours and (scales, rotations) are required"); return 1; }
  if (!pb->extra4 || !pb->d_opacity_raw || !pb->d_mask_raw || ((size_t)pb->extra4 & 15)) { hgs_set_error("hgs_backward_multi_params: null (or, extra4, unaligned) argument in params"); return 1; }
  if ((pb->max_radii2D || pb->grad_accum || pb->denom) && !(pb->max_radii2D && pb->grad_accum && pb->denom)) {
    hgs_set_error("hgs_backward_multi_params: incomplete statistics group");
    return 1;
  }
  if (pb->kind == HGS_PARAMS_HAIR) {
    if (!pb->endpoints || !pb->endpoint_pairs || !pb->seg_contrib || !pb->d_width || ((size_t)pb->seg_contrib & 15)) {
      hgs_set_error("hgs_backward_multi_params: null (or, seg_contrib, unaligned) hair argument"); return 1;
    }
    if (pb->head_tail.out) { hgs_set_error("hgs_backward_multi_params: the strand model's deferred tail rides in hgs_hair_endpoint_gather"); return 1; }
  } else if (!pb->rotation_raw || !pb->d_means3D || !pb->d_scaling_raw || !pb->d_rotation_raw || ((size_t)pb->rotation_raw & 15) ||
             ((size_t)pb->d_rotation_raw & 15)) {
    hgs_set_error("hgs_backward_multi_params: null (or, rotation_raw / d_rotation_raw, unaligned) cloud argument"); return 1;
  }
  if (check_adam_inline(pb->adam, pb->kind == HGS_PARAMS_HAIR ? 4 : 6, "hgs_backward_multi_params")) return 1;
  HgsParamBackward p = *pb;
  return backward_impl(stream, P, D, M, R, W, H, bg7, means3D, shs, nullptr, scales, 1.f, rotations, nullptr, viewmatrix,
                       projmatrix, campos, tan_fovx, tan_fovy, radii, geom_buf, binning_buf, image_buf, dL_dpix_planes7, scratch,
                       4, nullptr, p.dL_dmeans2D_rgb, nullptr, nullptr, nullptr, nullptr, nullptr, dL_dsh, nullptr, nullptr, &p);
}

int hgs_backward(void* stream, int P, int D, int M, int R, int W, int H, const float* bg, const float* means3D,
                 const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                 const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                 const float* campos, float tan_fovx, float tan_fovy, const int* radii, const void* geom_buf,
                 const void* binning_buf, const void* image_buf, const float* dL_dpix, void* scratch,
                 float* dL_dmeans2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolors, float* dL_dmeans3D,
                 float* dL_dcov3D, float* dL_dsh, float* dL_dscales, float* dL_drotations) {
  if (!dL_dpix && P > 0) { hgs_set_error("null dL_dpix"); return 1; }
  const size_t HW = (size_t)H * W;
  const float* planes[3] = {dL_dpix, dL_dpix + HW, dL_dpix + 2 * HW};
  return backward_impl(stream, P, D, M, R, W, H, bg, means3D, shs, colors_precomp, scales, scale_modifier, rotations,
                       cov3D_precomp, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii, geom_buf, binning_buf,
                       image_buf, planes, scratch, 0, nullptr, dL_dmeans2D, dL_dconic, dL_dopacity, dL_dcolors,
                       dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations);
}

int hgs_backward_multi(void* stream, int P, int D, int M, int R, int W, int H, const float* bg7, const float* means3D,
                       const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                       const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                       const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                       const void* geom_buf, const void* binning_buf, const void* image_buf,
                       const float* const* dL_dpix_planes7, void* scratch, float* dL_dextra4, float* dL_dmeans2D_rgb, float* dL_dconic, float* dL_dopacity,
                       float* dL_dcolors, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscales,
                       float* dL_drotations) {
  return backward_impl(stream, P, D, M, R, W, H, bg7, means3D, shs, colors_precomp, scales, scale_modifier, rotations,
                       cov3D_precomp, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii, geom_buf, binning_buf,
                       image_buf, dL_dpix_planes7, scratch, 4, dL_dextra4, dL_dmeans2D_rgb, dL_dconic, dL_dopacity, dL_dcolors,
                       dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations);
}

int hgs_mark_visible(void* stream, int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     uint8_t* present) {
  (void)projmatrix;  // the reference computes p_proj but only tests view-space z (auxiliary.h:149-154)
  if (P == 0) return 0;
  if (!means3D || !viewmatrix || !present) { hgs_set_error("null input"); return 1; }
  return hgs_launch_mark_visible((hipStream_t)stream, P, means3D, viewmatrix, present);
}

// An event pair around a launch reads the kernel's duration PLUS a fixed bracket cost (the gap between the first event
// and the kernel's start, and between its end and the second event): about 6 us here, which is most of a small kernel's
// reading.  The cost is measured once with empty kernels -- bracket(1 kernel) minus the marginal cost of one more kernel
// inside the same bracket -- and subtracted from every reading, so that the figures agree with rocprofv3's durations.
__global__ void hgs_empty_kernel() {}
static float g_prof_bracket_ms = -1.f;
static int prof_calibrate() {
  hipEvent_t a = nullptr, b = nullptr;
  HGS_CHECK_HIP(hipEventCreate(&a));
  HGS_CHECK_HIP(hipEventCreate(&b));
  float best1 = 1e9f, best2 = 1e9f;
  for (int rep = 0; rep < 24; rep++) {
    for (int n = 1; n <= 2; n++) {
      HGS_CHECK_HIP(hipDeviceSynchronize());
      HGS_CHECK_HIP(hipEventRecord(a, nullptr));
      for (int k = 0; k < n; k++) hipLaunchKernelGGL(hgs_empty_kernel, dim3(1), dim3(64), 0, nullptr);
      HGS_CHECK_HIP(hipEventRecord(b, nullptr));
      HGS_CHECK_HIP(hipEventSynchronize(b));
      float ms = 0.f;
      HGS_CHECK_HIP(hipEventElapsedTime(&ms, a, b));
      if (rep >= 4) { if (n == 1) best1 = ms < best1 ? ms : best1; else best2 = ms < best2 ? ms : best2; }
    }
  }
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  const float kernel = best2 - best1 > 0.f ? best2 - best1 : 0.f;   // one more empty kernel inside the bracket
  g_prof_bracket_ms = best1 - kernel > 0.f ? best1 - kernel : 0.f;
  return 0;
}
int hgs_prof_enable(int on) {
  if (on && g_prof_bracket_ms < 0.f && prof_calibrate()) return 1;
  g_prof_on = on != 0;
  return 0;
}
double hgs_prof_bracket_overhead_ms(void) { return g_prof_bracket_ms < 0.f ? 0.0 : (double)g_prof_bracket_ms; }
const char* hgs_prof_kernel_name(int id) { return (id >= 0 && id < HGS_K_COUNT) ? kKernelNames[id] : ""; }
int hgs_prof_collect(double* total_ms, long long* launches) {
  for (auto& r : g_prof_log) {
    HGS_CHECK_HIP(hipEventSynchronize(r.b));
    float ms = 0.f;
    HGS_CHECK_HIP(hipEventElapsedTime(&ms, r.a, r.b));
    ms -= g_prof_bracket_ms > 0.f ? g_prof_bracket_ms : 0.f;
    if (ms < 0.f) ms = 0.f;
    if (total_ms) total_ms[r.id] += ms;
    if (launches) launches[r.id] += 1;
    g_prof_pool.push_back(r.a);
    g_prof_pool.push_back(r.b);
  }
  g_prof_log.clear();
  return 0;
}

size_t hgs_dist2_scratch_bytes(int P) { return hgs_dist2_scratch(P > 0 ? P : 0); }
int hgs_dist2(void* stream, int P, const float* points, float* out, void* scratch, size_t scratch_bytes) {
  if (P == 0) return 0;
  if (!points || !out) { hgs_set_error("null input"); return 1; }
  return hgs_launch_dist2((hipStream_t)stream, P, points, out, scratch, scratch_bytes);
}

size_t hgs_view_targets_bytes(void) { return sizeof(HgsViewTargets); }
size_t hgs_head_params_bytes(void) { return sizeof(HgsHeadParams); }
size_t hgs_strand_fusion_bytes(void) { return sizeof(HgsStrandFusion); }

int hgs_iteration_prologue(void* stream, const HgsViewTargets* table, int view, HgsViewTargets* slot, float lr, float* lr_dst,
                           void* zero_ptr, size_t zero_bytes, const HgsAdamPrep* adam_prep) {
  if (!table || !slot || view < 0) { hgs_set_error("hgs_iteration_prologue: bad arguments"); return 1; }
  if (((size_t)zero_ptr & 3) || (zero_bytes & 3) || (zero_bytes && !zero_ptr)) { hgs_set_error("hgs_iteration_prologue: zero range must be 4-byte multiples"); return 1; }
  hipStream_t s = (hipStream_t)stream;
  const HgsPrologue p = {table, view, slot, lr, lr_dst, zero_ptr, zero_bytes, adam_prep};
  {
    HgsProfScope _prof(s, HGS_K_MISC);
    hipLaunchKernelGGL(select_view_kernel, dim3(hgs_prologue_blocks(zero_bytes / 4)), dim3(256), 0, s, p);
  }
  HGS_CHECK_LAUNCH();
  return 0;
}
int hgs_select_view(void* stream, const HgsViewTargets* table, int view, HgsViewTargets* slot, float lr, float* lr_dst) {
  return hgs_iteration_prologue(stream, table, view, slot, lr, lr_dst, nullptr, 0, nullptr);
}

// ---- re-pointing the prologue node of a captured graph ----------------------------------------------------------------
int hgs_graph_find_prologues(void* graph, int max_nodes, void** nodes_out, float* lr_out, int* n_out) {
  if (!graph || !nodes_out || !n_out || max_nodes < 1) { hgs_set_error("hgs_graph_find_prologues: bad arguments"); return 1; }
  size_t n = 0;
  HGS_CHECK_HIP(hipGraphGetNodes((hipGraph_t)graph, nullptr, &n));
  std::vector<hipGraphNode_t> nodes(n);
  if (n) HGS_CHECK_HIP(hipGraphGetNodes((hipGraph_t)graph, nodes.data(), &n));
  int count = 0;
  for (size_t i = 0; i < n; i++) {
    hipGraphNodeType t;
    HGS_CHECK_HIP(hipGraphNodeGetType(nodes[i], &t));
    if (t != hipGraphNodeTypeKernel) continue;
    hipKernelNodeParams kp;
    HGS_CHECK_HIP(hipGraphKernelNodeGetParams(nodes[i], &kp));
    int n_params = 1;                                      // the prologue is the LAST argument of all three kernels
    if (kp.func != (void*)select_view_kernel && !hgs_strands_prologue_kernel(kp.func, &n_params) &&
        !hgs_preprocess_prologue_kernel(kp.func, &n_params)) continue;
    if (!kp.kernelParams) continue;
    const HgsPrologue* pro = (const HgsPrologue*)kp.kernelParams[n_params - 1];
    if (!pro->table) continue;                             // (a parameter launch without a rider)
    if (count < max_nodes) {
      nodes_out[count] = (void*)nodes[i];
      if (lr_out) lr_out[count] = pro->lr;
    }
    count++;
  }
  *n_out = count;
  if (count > max_nodes) { hgs_set_error("hgs_graph_find_prologues: the graph holds %d prologue launches (room for %d)", count, max_nodes); return 1; }
  return 0;
}
int hgs_graph_find_prologue(void* graph, void** node_out) {
  int n = 0;
  if (!node_out) { hgs_set_error("hgs_graph_find_prologue: bad arguments"); return 1; }
  if (hgs_graph_find_prologues(graph, 1, node_out, nullptr, &n)) return 1;
  if (n != 1) { hgs_set_error("hgs_graph_find_prologue: the graph holds %d prologue launches (need exactly 1)", n); return 1; }
  return 0;
}
int hgs_graph_set_prologue(void* graph_exec, void* node, const HgsViewTargets* table, int view, HgsViewTargets* slot, float lr,
                           float* lr_dst, void* zero_ptr, size_t zero_bytes) {
  if (!graph_exec || !node || !table || !slot || view < 0) { hgs_set_error("hgs_graph_set_prologue: bad arguments"); return 1; }
  // the node's own launch (kernel, grid, every other argument) with a new prologue as its last argument
  hipKernelNodeParams kp;
  HGS_CHECK_HIP(hipGraphKernelNodeGetParams((hipGraphNode_t)node, &kp));
  int n_params = 1;
  if (kp.func != (void*)select_view_kernel && !hgs_strands_prologue_kernel(kp.func, &n_params) &&
      !hgs_preprocess_prologue_kernel(kp.func, &n_params)) {
    hgs_set_error("hgs_graph_set_prologue: not a prologue node");
    return 1;
  }
  if (((const HgsPrologue*)kp.kernelParams[n_params - 1])->zero_bytes != zero_bytes) {   // (the grid depends on it)
    hgs_set_error("hgs_graph_set_prologue: the zero range differs from the captured one");
    return 1;
  }
  HgsPrologue p = {table, view, slot, lr, lr_dst, zero_ptr, zero_bytes,
                   ((const HgsPrologue*)kp.kernelParams[n_params - 1])->adam_prep};   // (what the node was captured with)
  void* args[32];
  for (int i = 0; i < n_params - 1; i++) args[i] = kp.kernelParams[i];
  args[n_params - 1] = &p;
  kp.kernelParams = args;
  kp.extra = nullptr;
  HGS_CHECK_HIP(hipGraphExecKernelNodeSetParams((hipGraphExec_t)graph_exec, (hipGraphNode_t)node, &kp));
  return 0;
}

int hgs_set_view_queue(void* stream, int* queue, int n, const int* views_host, float lr, float* lr_slot) {
  if (!queue || !views_host || n < 1 || n > HGS_VIEW_QUEUE_MAX) { hgs_set_error("hgs_set_view_queue: bad arguments (1 <= n <= %d)", HGS_VIEW_QUEUE_MAX); return 1; }
  HgsViewQueueArgs a = {};
  for (int i = 0; i < n; i++) a.v[i] = views_host[i];
  hipStream_t s = (hipStream_t)stream;
  {
    HgsProfScope _prof(s, HGS_K_MISC);
    hipLaunchKernelGGL(set_view_queue_kernel, dim3(1), dim3(64), 0, s, queue, n, a, lr, lr_slot);
  }
  HGS_CHECK_LAUNCH();
  return 0;
}

int hgs_select_view_queued(void* stream, const HgsViewTargets* table, int n_views, const int* view_index,
                           HgsViewTargets* slot, const float* lr_slot, float* lr_dst) {
  if (!table || !slot || !view_index || n_views < 1) { hgs_set_error("hgs_select_view_queued: bad arguments"); return 1; }
  hipStream_t s = (hipStream_t)stream;
  {
    HgsProfScope _prof(s, HGS_K_MISC);
    hipLaunchKernelGGL(select_view_queued_kernel, dim3(1), dim3(64), 0, s, table, n_views, view_index, slot, lr_slot, lr_dst);
  }
  HGS_CHECK_LAUNCH();
  return 0;
}

int hgs_densify_stats(void* stream, int P, const int* radii, const float* dL_dmean2D, int stride, float* max_radii2D,
                      float* xyz_gradient_accum, float* denom) {
  if (P <= 0) return 0;
  if (!radii || !dL_dmean2D || stride < 2 || !max_radii2D || !xyz_gradient_accum || !denom) {
    hgs_set_error("hgs_densify_stats: bad arguments");
    return 1;
  }
  hipStream_t s = (hipStream_t)stream;
  {
    HgsProfScope _prof(s, HGS_K_MISC);
    hipLaunchKernelGGL(densify_stats_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, radii, dL_dmean2D, stride, max_radii2D,
                       xyz_gradient_accum, denom);
  }
  HGS_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
