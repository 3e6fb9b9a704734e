// hgs_strands.hip -- fused strand geometry: shared segment endpoints -> per-Gaussian (mean, scale, quaternion,
// direction), forward and backward, one lane per segment.
//
// Replaces, for the Stage-III model, the ~30 small PyTorch kernels (x3 per iteration, plus their autograd) behind
// HairGaussianModel.get_xyz / get_scaling / get_rotation / get_orientation (reference scene/hair_gaussian_model.py
// :134-201 and utils/transform.py:69-86), computed ONCE per optimizer step and shared by the three raster passes.
//   mean   = (e0 + e1) / 2
//   scale  = (max(|e1-e0|/2 * f, 1e-7), exp(w), exp(w))          f = dist_to_scale_factor
//   quat   = rotation x_hat -> d = (e1-e0)/|e1-e0| :  normalize(1 + d.x, 0, -d.z, d.y)   (w,x,y,z; w >= 0)
//            -- the closed form of matrix_to_quaternion(I + K + K^2/(1+c)), K = skew(x_hat x d), c = x_hat.d;
//            identity for collapsed segments (|e1-e0| <= 1e-7)
//   dir    = d (x_hat for collapsed segments)
// Backward scatters into the shared endpoints with fp32 atomics (an endpoint of a strand has <= 2 segments:
// two-term sums commute, so the result is order-independent for chains).
#include "hgs_common.h"
#include "hgs_smooth.h"
#include "hgs_head_tail.h"
#include "hgs_prologue.h"
#include "hgs_strand_bwd.h"
#include "hgs_strand_fwd.h"

namespace {

#define MINV 1e-7f

__global__ __launch_bounds__(256) void strand_fwd_kernel(int P, const float* __restrict__ ep, const long long* __restrict__ pairs,
                                                         const float* __restrict__ width, float f, float* __restrict__ xyz,
                                                         float* __restrict__ scale, float* __restrict__ quat,
                                                         float* __restrict__ dir, const float* __restrict__ opacity_raw,
                                                         const float* __restrict__ mask_raw, float* __restrict__ opacity,
                                                         float* __restrict__ extra4, HgsStrandFusion fu, HgsPrologue pro) {
  // (the iteration prologue as a rider, include/hgs.h HgsPrologue: the launch's last workgroups; `pro` stays the LAST argument)
  const unsigned npro = pro.table ? hgs_prologue_blocks(pro.zero_bytes / 4) : 0u;
  if (blockIdx.x >= gridDim.x - npro) { hgs_prologue_block(pro, blockIdx.x - (gridDim.x - npro), npro); return; }
  const int nb_seg = (P + 255) / 256;
  if ((int)blockIdx.x >= nb_seg) {   // extra workgroups: smoothness partial sums over the same endpoints
    __shared__ float red[4];
    hgs_smooth_fwd_block((int)blockIdx.x - nb_seg, fu.n_smooth, ep, fu.smooth_pairs, fu.cos_threshold, fu.eps, fu.smooth_partials, red, (float4*)fu.smooth_pair_grads);
    return;
  }
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= P) return;
  const long long i0 = pairs[2 * (size_t)k], i1 = pairs[2 * (size_t)k + 1];
  const HgsStrandGaussian sgn = hgs_strand_gaussian(ep[3 * i0], ep[3 * i0 + 1], ep[3 * i0 + 2], ep[3 * i1], ep[3 * i1 + 1],
                                                    ep[3 * i1 + 2], width[k], f);
  xyz[3 * (size_t)k] = sgn.mx; xyz[3 * (size_t)k + 1] = sgn.my; xyz[3 * (size_t)k + 2] = sgn.mz;
  scale[3 * (size_t)k] = sgn.s0;
  scale[3 * (size_t)k + 1] = sgn.sw;
  scale[3 * (size_t)k + 2] = sgn.sw;
  ((float4*)quat)[k] = make_float4(sgn.q0, sgn.q1, sgn.q2, sgn.q3);
  if (dir) { dir[3 * (size_t)k] = sgn.ux; dir[3 * (size_t)k + 1] = sgn.uy; dir[3 * (size_t)k + 2] = sgn.uz; }
  if (opacity) opacity[k] = hgs_sigmoid(opacity_raw[k]);                                          // gaussian_model.py:93-95
  if (extra4) ((float4*)extra4)[k] = make_float4(hgs_sigmoid(mask_raw[k]), sgn.ux, sgn.uy, sgn.uz);  // :97-99 + direction
}

// (six waves per SIMD = 80 VGPRs without spills instead of 85: nothing at 100 k Gaussians, 22.2 -> 19.9 us at 200 k, 69.9 -> 64.8
// at 1 M, where the launch's ~8000 workgroups run in rounds of resident ones; eight would spill)
#ifndef HGS_SBW_WAVES
#define HGS_SBW_WAVES 6
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HGS_SBW_WAVES))) void strand_bwd_kernel(HgsStrandBwdArgs A) {
  hgs_strand_bwd_block<false>(A, blockIdx.x, gridDim.x);     // (device code: hgs_strand_bwd.h)
}
// the endpoint gather alone (hgs_hair_endpoint_gather): segment contributions and pair gradients are read, not evaluated
__global__ __launch_bounds__(256) void strand_gather_kernel(HgsStrandBwdArgs A) {
  hgs_strand_bwd_block<true>(A, blockIdx.x, gridDim.x);
}

// ---- Stage-I cloud: raw parameters -> rasterizer inputs (scene/gaussian_model.py:118-157) -------------------------------
__global__ __launch_bounds__(256) void cloud_fwd_kernel(int P, const float* __restrict__ s_raw, const float* __restrict__ r_raw,
                                                        const float* __restrict__ o_raw, const float* __restrict__ m_raw,
                                                        float* __restrict__ scale, float* __restrict__ quat,
                                                        float* __restrict__ opacity, float* __restrict__ extra4, HgsPrologue pro) {
  const unsigned npro = pro.table ? hgs_prologue_blocks(pro.zero_bytes / 4) : 0u;      // (as strand_fwd_kernel)
  if (blockIdx.x >= gridDim.x - npro) { hgs_prologue_block(pro, blockIdx.x - (gridDim.x - npro), npro); return; }
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= P) return;
  const HgsCloudGaussian c = hgs_cloud_gaussian(s_raw[3 * (size_t)k], s_raw[3 * (size_t)k + 1], s_raw[3 * (size_t)k + 2],
                                                ((const float4*)r_raw)[k], o_raw[k], m_raw[k]);
  scale[3 * (size_t)k] = c.s0; scale[3 * (size_t)k + 1] = c.s1; scale[3 * (size_t)k + 2] = c.s2;
  ((float4*)quat)[k] = c.q;
  opacity[k] = c.opacity;
  ((float4*)extra4)[k] = c.extra;
}

__global__ __launch_bounds__(256) void cloud_bwd_kernel(int P, const float* __restrict__ s_raw, const float* __restrict__ r_raw,
                                                        const float* __restrict__ opacity, const float* __restrict__ extra4,
                                                        const float* __restrict__ g_scale, const float* __restrict__ g_quat,
                                                        const float* __restrict__ g_opacity, const float* __restrict__ g_extra4,
                                                        float* __restrict__ d_s, float* __restrict__ d_r,
                                                        float* __restrict__ d_o, float* __restrict__ d_m, HgsStrandFusion fu) {
  if (fu.head_tail.out && blockIdx.x == gridDim.x - 1) { hgs_head_tail_block(fu.head_tail); return; }   // (as strand_bwd_kernel)
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= P) return;
  if (fu.radii)                      // densification statistics of this Gaussian (hgs_densify_stats)
    hgs_densify_stats_lane(k, fu.radii[k], fu.dmean2D[(size_t)k * fu.dmean2D_stride], fu.dmean2D[(size_t)k * fu.dmean2D_stride + 1],
                           fu.max_radii2D, fu.grad_accum, fu.denom);
  // (the arithmetic: hgs_cloud_param_grads, hgs_strand_bwd.h -- shared with the rasterizer backward's cloud lanes)
  const float gs[3] = {g_scale[3 * (size_t)k], g_scale[3 * (size_t)k + 1], g_scale[3 * (size_t)k + 2]};
  const HgsCloudParamGrads o = hgs_cloud_param_grads(expf(s_raw[3 * (size_t)k]), expf(s_raw[3 * (size_t)k + 1]), expf(s_raw[3 * (size_t)k + 2]),
                                                     ((const float4*)r_raw)[k], opacity[k], extra4[4 * (size_t)k], gs,
                                                     ((const float4*)g_quat)[k], g_opacity[k], ((const float4*)g_extra4)[k]);
  d_s[3 * (size_t)k] = o.d_s[0]; d_s[3 * (size_t)k + 1] = o.d_s[1]; d_s[3 * (size_t)k + 2] = o.d_s[2];
  d_o[k] = o.d_o;
  d_m[k] = o.d_m;
  ((float4*)d_r)[k] = o.d_r;
}

}  // namespace

// ---- the iteration prologue as a rider of the forward launches (include/hgs.h HgsPrologue) ------------------------------
bool hgs_strands_prologue_kernel(const void* func, int* n_params) {
  if (func == (const void*)strand_fwd_kernel) { *n_params = 15; return true; }
  if (func == (const void*)cloud_fwd_kernel) { *n_params = 10; return true; }
  return false;
}
static inline unsigned rider_blocks(const HgsPrologue& p) { return p.table ? hgs_prologue_blocks(p.zero_bytes / 4) : 0u; }
// validates the rider; with nothing to ride on (P == 0) it is launched on its own
static int prologue_rider(void* stream, int P, const HgsPrologue& p, const char* who) {
  if (!p.table) return 0;
  if (!p.slot || p.view < 0 || ((size_t)p.zero_ptr & 3) || (p.zero_bytes & 3) || (p.zero_bytes && !p.zero_ptr)) {
    hgs_set_error("%s: bad prologue group", who);
    return 1;
  }
  if (P == 0) return hgs_iteration_prologue(stream, p.table, p.view, p.slot, p.lr, p.lr_dst, p.zero_ptr, p.zero_bytes, p.adam_prep);
  return 0;
}

extern "C" {

int hgs_strand_geometry_forward(void* stream, int P, const float* endpoints, const long long* endpoint_pairs,
                                const float* width, float dist_to_scale_factor, float* xyz, float* scale, float* quat,
                                float* dir) {
  if (P == 0) return 0;
  if (!endpoints || !endpoint_pairs || !width || !xyz || !scale || !quat || !dir) { hgs_set_error("hgs_strand_geometry_forward: null argument"); return 1; }
  hipStream_t s = (hipStream_t)stream;
  {
    HgsProfScope _prof(s, HGS_K_STRAND_FWD);
    hipLaunchKernelGGL(strand_fwd_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, endpoints, endpoint_pairs, width,
                       dist_to_scale_factor, xyz, scale, quat, dir, (const float*)nullptr, (const float*)nullptr,
                       (float*)nullptr, (float*)nullptr, HgsStrandFusion{}, HgsPrologue{});
  }
  HGS_CHECK_LAUNCH();
  return 0;
}

int hgs_strand_geometry_backward(void* stream, int P, int E, const float* endpoints, const long long* endpoint_pairs,
                                 const float* width, float dist_to_scale_factor, const float* g_xyz, const float* g_scale,
                                 const float* g_quat, const float* g_dir, float* d_endpoints, float* d_width) {
  if (!d_endpoints || !d_width) { hgs_set_error("hgs_strand_geometry_backward: null output"); return 1; }
  hipStream_t s = (hipStream_t)stream;
  if (hgs_zero_async(s, d_endpoints, (size_t)E * 3 * sizeof(float))) return 1;
  if (P == 0) return 0;
  {
    HgsProfScope _prof(s, HGS_K_STRAND_BWD);
    const HgsStrandBwdArgs A = {P, endpoints, endpoint_pairs, width, dist_to_scale_factor, g_xyz, g_scale, g_quat, g_dir,
                                d_endpoints, d_width, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, HgsStrandFusion{}, nullptr, HgsAdamInline{}};
    hipLaunchKernelGGL(strand_bwd_kernel, dim3((P + 255) / 256), dim3(256), 0, s, A);
  }
  HGS_CHECK_LAUNCH();
  return 0;
}

int hgs_hair_params_forward(void* stream, int P, const float* endpoints, const long long* endpoint_pairs,
                            const float* width, float dist_to_scale_factor, const float* opacity_raw,
                            const float* mask_raw, float* xyz, float* scale, float* quat, float* dir, float* opacity,
                            float* extra4, const HgsStrandFusion* fusion) {
  HgsStrandFusion fu = fusion ? *fusion : HgsStrandFusion{};
  if (prologue_rider(stream, P, fu.prologue, "hgs_hair_params_forward")) return 1;
  if (P == 0) return 0;
  const bool smooth = fu.smooth_pairs && fu.n_smooth > 0 && fu.smooth_partials;
  if (!smooth) fu.n_smooth = 0;
  if (!endpoints || !endpoint_pairs || !width || !opacity_raw || !mask_raw || !xyz || !scale || !quat || !opacity || !extra4) {
    hgs_set_error("hgs_hair_params_forward: null argument");
    return 1;
  }
  hipStream_t s = (hipStream_t)stream;
  {
    HgsProfScope _prof(s, HGS_K_STRAND_FWD);
    const HgsPrologue pro = fu.prologue;
    fu.prologue = HgsPrologue{};     // (handed over as the kernel's last argument, where the graph functions find it)
    hipLaunchKernelGGL(strand_fwd_kernel, dim3((P + 255) / 256 + (fu.n_smooth + 255) / 256 + rider_blocks(pro)), dim3(256), 0, s, P,
                       endpoints, endpoint_pairs, width, dist_to_scale_factor, xyz, scale, quat, dir, opacity_raw, mask_raw,
                       opacity, extra4, fu, pro);
  }
  HGS_CHECK_LAUNCH();
  return 0;
}

int hgs_hair_params_backward(void* stream, int P, int E, const float* endpoints, const long long* endpoint_pairs,
                             const float* width, float dist_to_scale_factor, const float* opacity, const float* extra4,
                             const float* g_xyz, const float* g_scale, const float* g_quat, const float* g_dir,
                             const float* g_opacity, const float* g_extra4, int accumulate_endpoints,
                             float* d_endpoints, float* d_width, float* d_opacity_raw, float* d_mask_raw,
                             const HgsStrandFusion* fusion) {
  HgsStrandFusion fu = fusion ? *fusion : HgsStrandFusion{};
  const bool smooth = fu.smooth_pairs && fu.n_smooth > 0 && fu.head_out && fu.grad_out;
  if (!smooth) fu.n_smooth = 0;
  const bool gather = fu.ep_segments != nullptr;
  if (gather && fu.n_endpoints != E) { hgs_set_error("hgs_hair_params_backward: HgsStrandFusion.n_endpoints != E"); return 1; }
  if (!gather) { fu.ep_pairs = nullptr; fu.n_endpoints = 0; }
  if (fu.radii && (!fu.dmean2D || fu.dmean2D_stride < 2 || !fu.max_radii2D || !fu.grad_accum || !fu.denom)) {
    hgs_set_error("hgs_hair_params_backward: incomplete statistics group in HgsStrandFusion");
    return 1;
  }
  if (!d_endpoints || !d_width || !d_opacity_raw || !d_mask_raw || !opacity || !extra4 || !g_opacity || !g_extra4) {
    hgs_set_error("hgs_hair_params_backward: null argument");
    return 1;
  }
  hipStream_t s = (hipStream_t)stream;
  if (!gather && !accumulate_endpoints && hgs_zero_async(s, d_endpoints, (size_t)E * 3 * sizeof(float))) return 1;
  if (P == 0 && fu.head_tail.out) { hgs_set_error("hgs_hair_params_backward: a deferred loss-head tail needs a launch (P > 0)"); return 1; }
  if (P == 0) return gather ? hgs_zero_async(s, d_endpoints, (size_t)E * 3 * sizeof(float)) : 0;
  {
    HgsProfScope _prof(s, HGS_K_STRAND_BWD);
    const int extra = gather ? (E + 255) / 256 : (fu.n_smooth + 255) / 256;
    const HgsStrandBwdArgs A = {P, endpoints, endpoint_pairs, width, dist_to_scale_factor, g_xyz, g_scale, g_quat, g_dir,
                                d_endpoints, d_width, opacity, extra4, g_opacity, g_extra4, d_opacity_raw, d_mask_raw, fu, nullptr, HgsAdamInline{}};
    hipLaunchKernelGGL(strand_bwd_kernel, dim3((P + 255) / 256 + extra + (fu.head_tail.out ? 1 : 0)), dim3(256), 0, s, A);
  }
  HGS_CHECK_LAUNCH();
  return 0;
}

int hgs_hair_endpoint_gather(void* stream, int E, const float* seg_contrib, const float* endpoints, float* d_endpoints,
                             const HgsStrandFusion* fusion, const HgsAdamInline* adam) {
  if (!fusion || !fusion->ep_segments) { hgs_set_error("hgs_hair_endpoint_gather: HgsStrandFusion.ep_segments is required (gather mode)"); return 1; }
  HgsStrandFusion fu = *fusion;
  const bool smooth = fu.smooth_pairs && fu.n_smooth > 0 && fu.head_out && fu.grad_out && fu.ep_pairs;
  if (!smooth) { fu.n_smooth = 0; fu.ep_pairs = nullptr; }
  if (fu.smooth_pair_grads && ((size_t)fu.smooth_pair_grads & 15)) { hgs_set_error("hgs_hair_endpoint_gather: smooth_pair_grads must be 16-byte aligned"); return 1; }
  fu.n_endpoints = E;
  fu.radii = nullptr;                                       // (the statistics belong to hgs_backward_multi_params)
  if (E == 0 && !fu.head_tail.out) return 0;
  if (E > 0 && (!seg_contrib || !endpoints || !d_endpoints || ((size_t)seg_contrib & 15))) {
    hgs_set_error("hgs_hair_endpoint_gather: null (or, seg_contrib, unaligned) argument");
    return 1;
  }
  hipStream_t s = (hipStream_t)stream;
  {
    HgsProfScope _prof(s, HGS_K_STRAND_BWD);
    // P = 0 per-segment workgroups: the launch is the endpoint lanes (+ the tail's workgroup)
    static const float4 kNone = {0.f, 0.f, 0.f, 0.f};
    const HgsStrandBwdArgs A = {0, endpoints, nullptr, nullptr, 0.f, nullptr, nullptr, nullptr, nullptr, d_endpoints, nullptr,
                                nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, fu,
                                E > 0 ? (const float4*)seg_contrib : &kNone, adam ? *adam : HgsAdamInline{}};
    if (A.adam.slot[0].p && (!A.adam.slot[0].m || !A.adam.slot[0].v || !A.adam.slot[0].coef)) { hgs_set_error("hgs_hair_endpoint_gather: incomplete Adam slot"); return 1; }
    // the in-lane update writes the endpoints: no lane of the launch may read them, i.e. a smoothness term needs its pair
    // gradients precomputed (HgsStrandFusion.smooth_pair_grads) -- evaluating the pairs here would race with the updates
    if (A.adam.slot[0].p && fu.n_smooth > 0 && !fu.smooth_pair_grads) {
      hgs_set_error("hgs_hair_endpoint_gather: the in-lane Adam update with a smoothness term needs HgsStrandFusion.smooth_pair_grads");
      return 1;
    }
    hipLaunchKernelGGL(strand_gather_kernel, dim3((E + 255) / 256 + (fu.head_tail.out ? 1 : 0)), dim3(256), 0, s, A);
  }
  HGS_CHECK_LAUNCH();
  return 0;
}

int hgs_cloud_params_forward(void* stream, int P, const float* scaling_raw, const float* rotation_raw,
                             const float* opacity_raw, const float* mask_raw, float* scale, float* quat, float* opacity,
                             float* extra4, const HgsStrandFusion* fusion) {
  const HgsPrologue pro = fusion ? fusion->prologue : HgsPrologue{};
  if (prologue_rider(stream, P, pro, "hgs_cloud_params_forward")) return 1;
  if (P == 0) return 0;
  if (!scaling_raw || !rotation_raw || !opacity_raw || !mask_raw || !scale || !quat || !opacity || !extra4) {
    hgs_set_error("hgs_cloud_params_forward: null argument");
    return 1;
  }
  hipStream_t s = (hipStream_t)stream;
  {
    HgsProfScope _prof(s, HGS_K_STRAND_FWD);
    hipLaunchKernelGGL(cloud_fwd_kernel, dim3((P + 255) / 256 + rider_blocks(pro)), dim3(256), 0, s, P, scaling_raw, rotation_raw,
                       opacity_raw, mask_raw, scale, quat, opacity, extra4, pro);
  }
  HGS_CHECK_LAUNCH();
  return 0;
}

int hgs_cloud_params_backward(void* stream, int P, const float* scaling_raw, const float* rotation_raw,
                              const float* opacity, const float* extra4, const float* g_scale, const float* g_quat,
                              const float* g_opacity, const float* g_extra4, float* d_scaling_raw, float* d_rotation_raw,
                              float* d_opacity_raw, float* d_mask_raw, const HgsStrandFusion* fusion) {
  if (P == 0 && fusion && fusion->head_tail.out) { hgs_set_error("hgs_cloud_params_backward: a deferred loss-head tail needs a launch (P > 0)"); return 1; }
  if (P == 0) return 0;
  if (!scaling_raw || !rotation_raw || !opacity || !extra4 || !g_scale || !g_quat || !g_opacity || !g_extra4 ||
      !d_scaling_raw || !d_rotation_raw || !d_opacity_raw || !d_mask_raw) {
    hgs_set_error("hgs_cloud_params_backward: null argument");
    return 1;
  }
  HgsStrandFusion fu = fusion ? *fusion : HgsStrandFusion{};
  if (fu.radii && (!fu.dmean2D || fu.dmean2D_stride < 2 || !fu.max_radii2D || !fu.grad_accum || !fu.denom)) {
    hgs_set_error("hgs_cloud_params_backward: incomplete statistics group in HgsStrandFusion");
    return 1;
  }
  hipStream_t s = (hipStream_t)stream;
  {
    HgsProfScope _prof(s, HGS_K_STRAND_BWD);
    hipLaunchKernelGGL(cloud_bwd_kernel, dim3((P + 255) / 256 + (fu.head_tail.out ? 1 : 0)), dim3(256), 0, s, P, scaling_raw, rotation_raw, opacity, extra4,
                       g_scale, g_quat, g_opacity, g_extra4, d_scaling_raw, d_rotation_raw, d_opacity_raw, d_mask_raw, fu);
  }
  HGS_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
