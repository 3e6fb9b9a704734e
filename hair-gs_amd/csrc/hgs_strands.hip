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
    hgs_smooth_fwd_block((int)blockIdx.x - nb_seg, fu.n_smooth, ep, fu.smooth_pairs, fu.cos_threshold, fu.eps, fu.smooth_partials, red);
    return;
  }
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= P) return;
  const long long i0 = pairs[2 * (size_t)k], i1 = pairs[2 * (size_t)k + 1];
  const float ax = ep[3 * i0], ay = ep[3 * i0 + 1], az = ep[3 * i0 + 2];
  const float bx = ep[3 * i1], by = ep[3 * i1 + 1], bz = ep[3 * i1 + 2];
  xyz[3 * (size_t)k] = (ax + bx) / 2.f; xyz[3 * (size_t)k + 1] = (ay + by) / 2.f; xyz[3 * (size_t)k + 2] = (az + bz) / 2.f;
  const float dx = bx - ax, dy = by - ay, dz = bz - az;
  const float L = sqrtf(dx * dx + dy * dy + dz * dz);
  const float ew = expf(width[k]);
  scale[3 * (size_t)k] = fmaxf(L / 2.f * f, MINV);
  scale[3 * (size_t)k + 1] = ew;
  scale[3 * (size_t)k + 2] = ew;
  float q0 = 1.f, q1 = 0.f, q2 = 0.f, q3 = 0.f, ux = 1.f, uy = 0.f, uz = 0.f;
  if (L > MINV) {
    const float il = 1.f / L;
    const float vx = dx * il, vy = dy * il, vz = dz * il;
    const float n0 = 1.f + vx;
    if (n0 > MINV) {
      const float in = 1.f / sqrtf(n0 * n0 + vz * vz + vy * vy);
      q0 = n0 * in; q1 = 0.f; q2 = -vz * in; q3 = vy * in;
    } else {  // d = -x_hat: half turn about z
      q0 = 0.f; q3 = 1.f;
    }
    if (L >= MINV) { ux = vx; uy = vy; uz = vz; }
  }
  ((float4*)quat)[k] = make_float4(q0, q1, q2, q3);
  if (dir) { dir[3 * (size_t)k] = ux; dir[3 * (size_t)k + 1] = uy; dir[3 * (size_t)k + 2] = uz; }
  if (opacity) opacity[k] = 1.f / (1.f + expf(-opacity_raw[k]));              // gaussian_model.py:93-95
  if (extra4) ((float4*)extra4)[k] = make_float4(1.f / (1.f + expf(-mask_raw[k])), ux, uy, uz);  // :97-99 + direction
}

// Gradient of segment k w.r.t. its two endpoints: endpoint 0 receives h - gD, endpoint 1 receives h + gD
// (h = half the gradient of the midpoint, gD = gradient w.r.t. delta = e1 - e0 from direction, quaternion and length).
struct SegGrads { const float* g_xyz; const float* g_scale; const float* g_quat; const float* g_dir; const float* g_extra4; };
__device__ __forceinline__ void segment_endpoint_grads(int k, const float* __restrict__ ep, const long long* __restrict__ pairs,
                                                       float f, const SegGrads& sg, float* h, float* gD) {
  // every load first and unconditional (the pointer tests are uniform): a lane that evaluates several segments then has
  // all of them in flight together instead of one dependent chain after the other
  const long long i0 = pairs[2 * (size_t)k], i1 = pairs[2 * (size_t)k + 1];
  float4 ge = make_float4(0.f, 0.f, 0.f, 0.f), gq = make_float4(0.f, 0.f, 0.f, 0.f);
  float gx[3] = {0.f, 0.f, 0.f}, gd[3] = {0.f, 0.f, 0.f}, gs0 = 0.f;
  if (sg.g_extra4) ge = ((const float4*)sg.g_extra4)[k];
  if (sg.g_quat) gq = ((const float4*)sg.g_quat)[k];
  if (sg.g_xyz) { gx[0] = sg.g_xyz[3 * (size_t)k]; gx[1] = sg.g_xyz[3 * (size_t)k + 1]; gx[2] = sg.g_xyz[3 * (size_t)k + 2]; }
  if (sg.g_dir) { gd[0] = sg.g_dir[3 * (size_t)k]; gd[1] = sg.g_dir[3 * (size_t)k + 1]; gd[2] = sg.g_dir[3 * (size_t)k + 2]; }
  if (sg.g_scale) gs0 = sg.g_scale[3 * (size_t)k];
  const float dx = ep[3 * i1] - ep[3 * i0], dy = ep[3 * i1 + 1] - ep[3 * i0 + 1], dz = ep[3 * i1 + 2] - ep[3 * i0 + 2];
  const float L = sqrtf(dx * dx + dy * dy + dz * dz);
  h[0] = 0.5f * gx[0]; h[1] = 0.5f * gx[1]; h[2] = 0.5f * gx[2];
  gD[0] = gD[1] = gD[2] = 0.f;
  if (L > MINV) {
    const float il = 1.f / L;
    const float vx = dx * il, vy = dy * il, vz = dz * il;
    float gvx = gd[0], gvy = gd[1], gvz = gd[2];  // gradient w.r.t. the unit direction (L > MINV implies L >= MINV)
    if (sg.g_extra4) { gvx += ge.y; gvy += ge.z; gvz += ge.w; }
    const float n0 = 1.f + vx;
    if (sg.g_quat && n0 > MINV) {
      const float in = 1.f / sqrtf(n0 * n0 + vz * vz + vy * vy);
      const float q0 = n0 * in, q2 = -vz * in, q3 = vy * in;
      const float dot = q0 * gq.x + q2 * gq.z + q3 * gq.w;  // q1 = 0
      const float gn0 = (gq.x - q0 * dot) * in, gn2 = (gq.z - q2 * dot) * in, gn3 = (gq.w - q3 * dot) * in;
      gvx += gn0; gvy += gn3; gvz -= gn2;
    }
    const float vd = vx * gvx + vy * gvy + vz * gvz;
    gD[0] = (gvx - vx * vd) * il; gD[1] = (gvy - vy * vd) * il; gD[2] = (gvz - vz * vd) * il;
    if (sg.g_scale && L / 2.f * f > MINV) {
      const float gs = gs0 * (0.5f * f);
      gD[0] += gs * vx; gD[1] += gs * vy; gD[2] += gs * vz;
    }
  }
}

__global__ __launch_bounds__(256) void strand_bwd_kernel(int P, const float* __restrict__ ep, const long long* __restrict__ pairs,
                                                         const float* __restrict__ width, float f,
                                                         const float* __restrict__ g_xyz, const float* __restrict__ g_scale,
                                                         const float* __restrict__ g_quat, const float* __restrict__ g_dir,
                                                         float* __restrict__ d_ep, float* __restrict__ d_width,
                                                         const float* __restrict__ opacity, const float* __restrict__ extra4,
                                                         const float* __restrict__ g_opacity, const float* __restrict__ g_extra4,
                                                         float* __restrict__ d_opacity_raw, float* __restrict__ d_mask_raw,
                                                         HgsStrandFusion fu) {
  // (the loss head's deferred tail, include/hgs.h HgsHeadTail: one spare workgroup behind the launch's own)
  if (fu.head_tail.out && blockIdx.x == gridDim.x - 1) { hgs_head_tail_block(fu.head_tail); return; }
  const int nb_seg = (P + 255) / 256;
  const SegGrads sg = {g_xyz, g_scale, g_quat, g_dir, g_extra4};
  const float smooth_scale = fu.n_smooth > 0
      ? fu.head_out[HGS_HEAD_G_SMOOTH] * fu.grad_out[0] / fmaxf(fu.head_out[HGS_HEAD_SMOOTH_COUNT], 1.f) : 0.f;
  if ((int)blockIdx.x >= nb_seg) {
    const int i = ((int)blockIdx.x - nb_seg) * 256 + threadIdx.x;
    if (fu.ep_segments) {
      // gather mode: one lane per ENDPOINT sums the contributions of its (<= 2) segments and (<= 4) smoothness pair
      // roles in a fixed order and stores once: no float atomics (each segment / pair is simply evaluated by every
      // endpoint it touches: ~500 flops per endpoint against 18 L2 atomics per segment)
      if (i >= fu.n_endpoints) return;
      // The (<= 2 + 4) evaluations are independent: absent slots (code < 0) evaluate item 0 and are masked out afterwards,
      // so that nothing branches between the loads of one evaluation and the next -- the lane's six dependent chains
      // (code -> index row -> endpoints) overlap instead of running one after the other (14.5 -> 11.6 us for the launch).
      float acc[3] = {0.f, 0.f, 0.f};
      const int2 cs = *(const int2*)(fu.ep_segments + 2 * (size_t)i);
      const bool with_smooth = fu.ep_pairs && fu.n_smooth > 0;
      int4 cp = make_int4(-1, -1, -1, -1);
      if (with_smooth) cp = *(const int4*)(fu.ep_pairs + 4 * (size_t)i);
      const int seg_code[2] = {cs.x, cs.y};
      float sh[2][3], sD[2][3];
#pragma unroll
      for (int s = 0; s < 2; s++) segment_endpoint_grads(seg_code[s] >= 0 ? seg_code[s] >> 1 : 0, ep, pairs, f, sg, sh[s], sD[s]);
      const int pair_code[4] = {cp.x, cp.y, cp.z, cp.w};
      float pg0[4][3], pg1[4][3];
      bool pok[4] = {false, false, false, false};
      if (with_smooth) {
#pragma unroll
        for (int s = 0; s < 4; s++)
          pok[s] = hgs_smooth_pair_grads(pair_code[s] >= 0 ? pair_code[s] >> 2 : 0, ep, fu.smooth_pairs, fu.cos_threshold, fu.eps,
                                         smooth_scale, pg0[s], pg1[s]) && pair_code[s] >= 0;
      }
#pragma unroll
      for (int s = 0; s < 2; s++) {
        const float sign = (seg_code[s] & 1) ? 1.f : -1.f;
        if (seg_code[s] >= 0) {
#pragma unroll
          for (int c = 0; c < 3; c++) acc[c] += sh[s][c] + sign * sD[s][c];
        }
      }
#pragma unroll
      for (int s = 0; s < 4; s++) {
        const int role = pair_code[s] & 3;                       // a0: -g0, a1: +g0, b0: -g1, b1: +g1
        const float sign = (role & 1) ? 1.f : -1.f;
        if (pok[s]) {
#pragma unroll
          for (int c = 0; c < 3; c++) acc[c] += sign * (role < 2 ? pg0[s][c] : pg1[s][c]);
        }
      }
      d_ep[3 * (size_t)i] = acc[0]; d_ep[3 * (size_t)i + 1] = acc[1]; d_ep[3 * (size_t)i + 2] = acc[2];
    } else if (i < fu.n_smooth) {   // scatter mode: smoothness gradient added into d_ep with atomics
      hgs_smooth_bwd_pair(i, ep, fu.smooth_pairs, fu.cos_threshold, fu.eps, smooth_scale, d_ep);
    }
    return;
  }
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= P) return;
  if (fu.radii) {                    // densification statistics of this Gaussian (hgs_densify_stats)
    const int r = fu.radii[k];
    if (r > 0) {
      fu.max_radii2D[k] = fmaxf(fu.max_radii2D[k], (float)r);
      const float gx = fu.dmean2D[(size_t)k * fu.dmean2D_stride], gy = fu.dmean2D[(size_t)k * fu.dmean2D_stride + 1];
      fu.grad_accum[k] += sqrtf(gx * gx + gy * gy);
      fu.denom[k] += 1.f;
    }
  }
  if (d_opacity_raw) { const float o = opacity[k]; d_opacity_raw[k] = g_opacity[k] * o * (1.f - o); }   // sigmoid'
  if (d_mask_raw) { const float m = extra4[4 * (size_t)k]; d_mask_raw[k] = g_extra4[4 * (size_t)k] * m * (1.f - m); }
  float gw = 0.f;
  if (g_scale) gw = (g_scale[3 * (size_t)k + 1] + g_scale[3 * (size_t)k + 2]) * expf(width[k]);
  if (!fu.ep_segments) {             // scatter mode: this segment's contribution to its two endpoints
    float h[3], gD[3];
    segment_endpoint_grads(k, ep, pairs, f, sg, h, gD);
    const long long i0 = pairs[2 * (size_t)k], i1 = pairs[2 * (size_t)k + 1];
    atomicAdd(&d_ep[3 * i0], h[0] - gD[0]); atomicAdd(&d_ep[3 * i0 + 1], h[1] - gD[1]); atomicAdd(&d_ep[3 * i0 + 2], h[2] - gD[2]);
    atomicAdd(&d_ep[3 * i1], h[0] + gD[0]); atomicAdd(&d_ep[3 * i1 + 1], h[1] + gD[1]); atomicAdd(&d_ep[3 * i1 + 2], h[2] + gD[2]);
  }
  d_width[k] = gw;
}

// ---- Stage-I cloud: raw parameters -> rasterizer inputs (scene/gaussian_model.py:118-157) -------------------------------
__device__ __forceinline__ int argmax3(float a, float b, float c) { return (b > a) ? ((c > b) ? 2 : 1) : ((c > a) ? 2 : 0); }
// column `ax` of the rotation matrix of the UNIT quaternion (w, x, y, z)
__device__ __forceinline__ void rot_column(int ax, float w, float x, float y, float z, float& d0, float& d1, float& d2) {
  if (ax == 0)      { d0 = 1.f - 2.f * (y * y + z * z); d1 = 2.f * (x * y + w * z);       d2 = 2.f * (x * z - w * y); }
  else if (ax == 1) { d0 = 2.f * (x * y - w * z);       d1 = 1.f - 2.f * (x * x + z * z); d2 = 2.f * (y * z + w * x); }
  else              { d0 = 2.f * (x * z + w * y);       d1 = 2.f * (y * z - w * x);       d2 = 1.f - 2.f * (x * x + y * y); }
}

__global__ __launch_bounds__(256) void cloud_fwd_kernel(int P, const float* __restrict__ s_raw, const float* __restrict__ r_raw,
                                                        const float* __restrict__ o_raw, const float* __restrict__ m_raw,
                                                        float* __restrict__ scale, float* __restrict__ quat,
                                                        float* __restrict__ opacity, float* __restrict__ extra4, HgsPrologue pro) {
  const unsigned npro = pro.table ? hgs_prologue_blocks(pro.zero_bytes / 4) : 0u;      // (as strand_fwd_kernel)
  if (blockIdx.x >= gridDim.x - npro) { hgs_prologue_block(pro, blockIdx.x - (gridDim.x - npro), npro); return; }
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= P) return;
  const float s0 = expf(s_raw[3 * (size_t)k]), s1 = expf(s_raw[3 * (size_t)k + 1]), s2 = expf(s_raw[3 * (size_t)k + 2]);
  scale[3 * (size_t)k] = s0; scale[3 * (size_t)k + 1] = s1; scale[3 * (size_t)k + 2] = s2;
  const float4 r = ((const float4*)r_raw)[k];
  const float n = sqrtf(r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w);
  const float inq = 1.f / fmaxf(n, 1e-12f);                              // F.normalize (get_rotation)
  ((float4*)quat)[k] = make_float4(r.x * inq, r.y * inq, r.z * inq, r.w * inq);
  const float ib = 1.f / n;                                              // build_rotation normalises without the clamp
  float d0, d1, d2;
  rot_column(argmax3(s0, s1, s2), r.x * ib, r.y * ib, r.z * ib, r.w * ib, d0, d1, d2);
  opacity[k] = 1.f / (1.f + expf(-o_raw[k]));
  ((float4*)extra4)[k] = make_float4(1.f / (1.f + expf(-m_raw[k])), d0, d1, d2);
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
  if (fu.radii) {                    // densification statistics of this Gaussian (hgs_densify_stats)
    const int rr = fu.radii[k];
    if (rr > 0) {
      fu.max_radii2D[k] = fmaxf(fu.max_radii2D[k], (float)rr);
      const float gx = fu.dmean2D[(size_t)k * fu.dmean2D_stride], gy = fu.dmean2D[(size_t)k * fu.dmean2D_stride + 1];
      fu.grad_accum[k] += sqrtf(gx * gx + gy * gy);
      fu.denom[k] += 1.f;
    }
  }
  const float s0 = expf(s_raw[3 * (size_t)k]), s1 = expf(s_raw[3 * (size_t)k + 1]), s2 = expf(s_raw[3 * (size_t)k + 2]);
  d_s[3 * (size_t)k] = g_scale[3 * (size_t)k] * s0;
  d_s[3 * (size_t)k + 1] = g_scale[3 * (size_t)k + 1] * s1;
  d_s[3 * (size_t)k + 2] = g_scale[3 * (size_t)k + 2] * s2;
  const float4 ge = ((const float4*)g_extra4)[k];
  { const float o = opacity[k]; d_o[k] = g_opacity[k] * o * (1.f - o); }
  { const float m = extra4[4 * (size_t)k]; d_m[k] = ge.x * m * (1.f - m); }
  const float4 r = ((const float4*)r_raw)[k];
  const float n = sqrtf(r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w);
  const float in = 1.f / n;
  const float w = r.x * in, x = r.y * in, y = r.z * in, z = r.w * in;
  // dL/dq (q = unit quaternion) = the rasterizer's gradient of `quat` + J^T of the direction column
  const float4 gq = ((const float4*)g_quat)[k];
  float qw = gq.x, qx = gq.y, qy = gq.z, qz = gq.w;
  if (!(n > 1e-12f)) { qw = 0.f; qx = 0.f; qy = 0.f; qz = 0.f; }           // clamped branch of F.normalize: q = r / 1e-12
  const int ax = argmax3(s0, s1, s2);
  const float a = ge.y, b = ge.z, c = ge.w;                                // dL/d(direction)
  if (ax == 0) {        // (1-2(y^2+z^2), 2(xy+wz), 2(xz-wy))
    qw += 2.f * (b * z - c * y); qx += 2.f * (b * y + c * z); qy += 2.f * (-2.f * a * y + b * x - c * w); qz += 2.f * (-2.f * a * z + b * w + c * x);
  } else if (ax == 1) { // (2(xy-wz), 1-2(x^2+z^2), 2(yz+wx))
    qw += 2.f * (-a * z + c * x); qx += 2.f * (a * y - 2.f * b * x + c * w); qy += 2.f * (a * x + c * z); qz += 2.f * (-a * w - 2.f * b * z + c * y);
  } else {              // (2(xz+wy), 2(yz-wx), 1-2(x^2+y^2))
    qw += 2.f * (a * y - b * x); qx += 2.f * (a * z - b * w - 2.f * c * x); qy += 2.f * (a * w + b * z - 2.f * c * y); qz += 2.f * (a * x + b * y);
  }
  const float dot = w * qw + x * qx + y * qy + z * qz;                     // through r -> r / |r|
  ((float4*)d_r)[k] = make_float4((qw - w * dot) * in, (qx - x * dot) * in, (qy - y * dot) * in, (qz - z * dot) * in);
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
  if (P == 0) return hgs_iteration_prologue(stream, p.table, p.view, p.slot, p.lr, p.lr_dst, p.zero_ptr, p.zero_bytes);
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
    hipLaunchKernelGGL(strand_bwd_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, endpoints, endpoint_pairs, width,
                       dist_to_scale_factor, g_xyz, g_scale, g_quat, g_dir, d_endpoints, d_width, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr,
                       HgsStrandFusion{});
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
    hipLaunchKernelGGL(strand_bwd_kernel, dim3((P + 255) / 256 + extra + (fu.head_tail.out ? 1 : 0)), dim3(256), 0, s, P, endpoints,
                       endpoint_pairs, width, dist_to_scale_factor, g_xyz, g_scale, g_quat, g_dir, d_endpoints, d_width,
                       opacity, extra4, g_opacity, g_extra4, d_opacity_raw, d_mask_raw, fu);
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
