// hgs_strand_bwd.h -- device code of the strand parameters' backward (hgs_hair_params_backward): gradients of the
// rasterizer inputs -> endpoints / width / raw opacity / raw mask (+ smoothness gradient, densification statistics, the
// loss head's tail).  Kept apart from its kernel (hgs_strands.hip) with the loads that do not depend on the rasterizer's
// gradients -- adjacency codes, index rows, endpoints, the smoothness pairs -- in front of those that do.
// (Round 2 ran this code as a RIDER of the rasterizer's preprocess-backward launch -- strand workgroups dispatched behind
// the preprocess workgroups, prefetching, then waiting for a ticket before reading the gradients through an agent-scope
// acquire: correct, bit-identical, and slower, DESIGN.md section 6: 800 workgroups polling one word next to the 1200
// atomics on it cost 50 us, and without any wait the merged launch still took 26 us against 9.8 + 12.4 as two launches,
// because at the preprocess kernel's 128 registers the 1200 workgroups are not resident together.)
#pragma once
#include "hgs_common.h"
#include "hgs_smooth.h"
#include "hgs_head_tail.h"
#include "hgs_strand_fwd.h"
#include "hgs_adam.h"

#define HGS_STRAND_MINV 1e-7f


// Gradient of segment k w.r.t. its two endpoints: endpoint 0 receives h - gD, endpoint 1 receives h + gD
// (h = half the gradient of the midpoint, gD = gradient w.r.t. delta = e1 - e0 from direction, quaternion and length).
struct HgsSegGrads { const float* g_xyz; const float* g_scale; const float* g_quat; const float* g_dir; const float* g_extra4; };
struct HgsSegGeom { float dx, dy, dz; };   // e1 - e0 of the segment (independent of the rasterizer's gradients)
__device__ __forceinline__ HgsSegGeom hgs_segment_geom(int k, const float* __restrict__ ep, const long long* __restrict__ pairs) {
  const long long i0 = pairs[2 * (size_t)k], i1 = pairs[2 * (size_t)k + 1];
  return {ep[3 * i1] - ep[3 * i0], ep[3 * i1 + 1] - ep[3 * i0 + 1], ep[3 * i1 + 2] - ep[3 * i0 + 2]};
}
// The gradient values of one segment (what the rasterizer backward leaves for Gaussian k): gx = dL/dmean, gs0 = dL/dscale.x,
// gq = dL/dquaternion, gd = dL/ddirection, ge = dL/dextra4 (mask channel, then the blended direction).
struct HgsSegGradVals { float gx[3]; float gs0; float4 gq; float gd[3]; float4 ge; bool has_quat, has_extra, has_scale; };
// Evaluated operation by operation (no contraction) so that every translation unit that inlines it -- strand_bwd_kernel
// (hgs_strands.hip, built with contraction on) and the rasterizer's hair_preprocess_bwd lanes (hgs_preprocess.hip, built without)
// -- produces the same bits: what makes the fused backward equal to the two-launch one bit for bit.
__device__ __forceinline__ void hgs_segment_endpoint_grads_v(const HgsSegGeom& sgm, float f, const HgsSegGradVals& g, float* h, float* gD) {
#pragma clang fp contract(off)
  const float dx = sgm.dx, dy = sgm.dy, dz = sgm.dz;
  const float L = sqrtf(dx * dx + dy * dy + dz * dz);
  h[0] = 0.5f * g.gx[0]; h[1] = 0.5f * g.gx[1]; h[2] = 0.5f * g.gx[2];
  gD[0] = gD[1] = gD[2] = 0.f;
  if (L > HGS_STRAND_MINV) {
    const float il = 1.f / L;
    const float vx = dx * il, vy = dy * il, vz = dz * il;
    float gvx = g.gd[0], gvy = g.gd[1], gvz = g.gd[2];  // gradient w.r.t. the unit direction (L > HGS_STRAND_MINV implies L >= HGS_STRAND_MINV)
    if (g.has_extra) { gvx += g.ge.y; gvy += g.ge.z; gvz += g.ge.w; }
    const float n0 = 1.f + vx;
    if (g.has_quat && n0 > HGS_STRAND_MINV) {
      const float in = 1.f / sqrtf(n0 * n0 + vz * vz + vy * vy);
      const float q0 = n0 * in, q2 = -vz * in, q3 = vy * in;
      const float dot = q0 * g.gq.x + q2 * g.gq.z + q3 * g.gq.w;  // q1 = 0
      const float gn0 = (g.gq.x - q0 * dot) * in, gn2 = (g.gq.z - q2 * dot) * in, gn3 = (g.gq.w - q3 * dot) * in;
      gvx += gn0; gvy += gn3; gvz -= gn2;
    }
    const float vd = vx * gvx + vy * gvy + vz * gvz;
    gD[0] = (gvx - vx * vd) * il; gD[1] = (gvy - vy * vd) * il; gD[2] = (gvz - vz * vd) * il;
    if (g.has_scale && L / 2.f * f > HGS_STRAND_MINV) {
      const float gs = g.gs0 * (0.5f * f);
      gD[0] += gs * vx; gD[1] += gs * vy; gD[2] += gs * vz;
    }
  }
}
__device__ __forceinline__ void hgs_segment_endpoint_grads(int k, const HgsSegGeom& sgm, float f, const HgsSegGrads& sg, float* h, float* gD) {
  // every load first and unconditional (the pointer tests are uniform): a lane that evaluates several segments then has
  // all of them in flight together instead of one dependent chain after the other
  HgsSegGradVals g;
  g.ge = make_float4(0.f, 0.f, 0.f, 0.f); g.gq = make_float4(0.f, 0.f, 0.f, 0.f);
  g.gx[0] = g.gx[1] = g.gx[2] = 0.f; g.gd[0] = g.gd[1] = g.gd[2] = 0.f; g.gs0 = 0.f;
  if (sg.g_extra4) g.ge = *(const float4*)(sg.g_extra4 + 4 * (size_t)k);
  if (sg.g_quat) g.gq = *(const float4*)(sg.g_quat + 4 * (size_t)k);
  if (sg.g_xyz) { g.gx[0] = *(sg.g_xyz + 3 * (size_t)k); g.gx[1] = *(sg.g_xyz + 3 * (size_t)k + 1); g.gx[2] = *(sg.g_xyz + 3 * (size_t)k + 2); }
  if (sg.g_dir) { g.gd[0] = *(sg.g_dir + 3 * (size_t)k); g.gd[1] = *(sg.g_dir + 3 * (size_t)k + 1); g.gd[2] = *(sg.g_dir + 3 * (size_t)k + 2); }
  if (sg.g_scale) g.gs0 = *(sg.g_scale + 3 * (size_t)k);
  g.has_quat = sg.g_quat != nullptr; g.has_extra = sg.g_extra4 != nullptr; g.has_scale = sg.g_scale != nullptr;
  hgs_segment_endpoint_grads_v(sgm, f, g, h, gD);
}
// the densification statistics of one Gaussian (hgs_densify_stats; scene/hair_gaussian_model.py:1401-1408, train.py:170-171),
// the norm evaluated without contraction for the same reason
__device__ __forceinline__ void hgs_densify_stats_lane(int k, int radius, float gx, float gy, float* __restrict__ max_radii2D,
                                                       float* __restrict__ grad_accum, float* __restrict__ denom) {
#pragma clang fp contract(off)
  if (radius <= 0) return;
  max_radii2D[k] = fmaxf(max_radii2D[k], (float)radius);
  grad_accum[k] += sqrtf(gx * gx + gy * gy);
  denom[k] += 1.f;
}


// Stage-I cloud: gradients of the rasterizer inputs (scale, unit quaternion, opacity, extra4 = [mask, direction]) -> raw
// parameters (scene/gaussian_model.py:118-157 and their autograd).  s = exp(scaling_raw) as the forward stored it, r = the raw
// rotation.  Shared by cloud_bwd_kernel and the rasterizer backward's cloud lanes; no contraction (see above).
struct HgsCloudParamGrads { float d_s[3]; float4 d_r; float d_o, d_m; };
__device__ __forceinline__ HgsCloudParamGrads hgs_cloud_param_grads(float s0, float s1, float s2, float4 r, float o, float m,
                                                                   const float* g_scale, float4 gq, float g_opacity, float4 ge) {
#pragma clang fp contract(off)
  HgsCloudParamGrads out;
  out.d_s[0] = g_scale[0] * s0; out.d_s[1] = g_scale[1] * s1; out.d_s[2] = g_scale[2] * s2;
  out.d_o = g_opacity * o * (1.f - o);
  out.d_m = ge.x * m * (1.f - m);
  const float n = sqrtf(r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w);
  const float in = 1.f / n;
  const float w = r.x * in, x = r.y * in, y = r.z * in, z = r.w * in;
  // dL/dq (q = unit quaternion) = the rasterizer's gradient of `quat` + J^T of the direction column
  float qw = gq.x, qx = gq.y, qy = gq.z, qz = gq.w;
  if (!(n > 1e-12f)) { qw = 0.f; qx = 0.f; qy = 0.f; qz = 0.f; }           // clamped branch of F.normalize: q = r / 1e-12
  const int ax = hgs_argmax3(s0, s1, s2);
  const float a = ge.y, b = ge.z, c = ge.w;                                // dL/d(direction)
  if (ax == 0) {        // (1-2(y^2+z^2), 2(xy+wz), 2(xz-wy))
    qw += 2.f * (b * z - c * y); qx += 2.f * (b * y + c * z); qy += 2.f * (-2.f * a * y + b * x - c * w); qz += 2.f * (-2.f * a * z + b * w + c * x);
  } else if (ax == 1) { // (2(xy-wz), 1-2(x^2+z^2), 2(yz+wx))
    qw += 2.f * (-a * z + c * x); qx += 2.f * (a * y - 2.f * b * x + c * w); qy += 2.f * (a * x + c * z); qz += 2.f * (-a * w - 2.f * b * z + c * y);
  } else {              // (2(xz+wy), 2(yz-wx), 1-2(x^2+y^2))
    qw += 2.f * (a * y - b * x); qx += 2.f * (a * z - b * w - 2.f * c * x); qy += 2.f * (a * w + b * z - 2.f * c * y); qz += 2.f * (a * x + b * y);
  }
  const float dot = w * qw + x * qx + y * qy + z * qz;                     // through r -> r / |r|
  out.d_r = make_float4((qw - w * dot) * in, (qx - x * dot) * in, (qy - y * dot) * in, (qz - z * dot) * in);
  return out;
}

struct HgsStrandBwdArgs {
  int P; const float* ep; const long long* pairs; const float* width; float f;
  const float* g_xyz; const float* g_scale; const float* g_quat; const float* g_dir;
  float* d_ep; float* d_width;
  const float* opacity; const float* extra4; const float* g_opacity; const float* g_extra4;
  float* d_opacity_raw; float* d_mask_raw;
  HgsStrandFusion fu;
  // != NULL: the per-segment work is done -- the rasterizer backward's own per-Gaussian lanes left, for segment k, the gradient
  // of its first endpoint in seg_contrib[2 k] and of its second in seg_contrib[2 k + 1] (h - gD, h + gD; hgs_backward_multi_params)
  // -- and this launch is the endpoint gather alone (gather mode; no per-segment workgroups)
  const float4* seg_contrib;
  HgsAdamInline adam;   // gather-only launch: slot 0 = the endpoints' Adam state (p != NULL: the lane applies endpoint i's update)
};

// Workgroup `blk` of `nblk` (256 threads): [0, ceil(P / 256)) one lane per Gaussian; then, gather mode, one lane per
// endpoint (scatter mode: per smoothness pair); the last one runs the loss head's deferred tail when that is asked for.
// CONTRIB: compiled for the gather-only launch (A.seg_contrib given): none of the per-segment code, fewer registers, more waves
template <bool CONTRIB = false>
__device__ __forceinline__ void hgs_strand_bwd_block(const HgsStrandBwdArgs& A_in, unsigned blk, unsigned nblk) {
  HgsStrandBwdArgs A = A_in;
  if (CONTRIB) { A.P = 0; } else { A.seg_contrib = nullptr; }
  const HgsStrandFusion& fu = A.fu;
  const int P = A.P;
  const float* __restrict__ ep = A.ep;
  const long long* __restrict__ pairs = A.pairs;
  const float f = A.f;
  // (the loss head's deferred tail, include/hgs.h HgsHeadTail: one spare workgroup behind the launch's own)
  if (fu.head_tail.out && blk == nblk - 1) { hgs_head_tail_block(fu.head_tail); return; }
  const int nb_seg = CONTRIB ? 0 : (P + 255) / 256;
  const HgsSegGrads sg = {A.g_xyz, A.g_scale, A.g_quat, A.g_dir, A.g_extra4};
  const float smooth_scale = fu.n_smooth > 0
      ? fu.head_out[HGS_HEAD_G_SMOOTH] * fu.grad_out[0] / fmaxf(fu.head_out[HGS_HEAD_SMOOTH_COUNT], 1.f) : 0.f;
  if ((int)blk >= nb_seg) {
    const int i = ((int)blk - nb_seg) * 256 + threadIdx.x;
    if (fu.ep_segments) {
      // gather mode: one lane per ENDPOINT sums the contributions of its (<= 2) segments and (<= 4) smoothness pair
      // roles in a fixed order and stores once: no float atomics (each segment / pair is simply evaluated by every
      // endpoint it touches: ~500 flops per endpoint against 18 L2 atomics per segment)
      if (i >= fu.n_endpoints) return;
      const int ic = i;
      // The (<= 2 + 4) evaluations are independent: absent slots (code < 0) evaluate item 0 and are masked out afterwards,
      // so that nothing branches between the loads of one evaluation and the next -- the lane's six dependent chains
      // (code -> index row -> endpoints) overlap instead of running one after the other (14.5 -> 11.6 us for the launch).
      float acc[3] = {0.f, 0.f, 0.f};
      const int2 cs = *(const int2*)(fu.ep_segments + 2 * (size_t)ic);
      const bool with_smooth = fu.ep_pairs && fu.n_smooth > 0;
      int4 cp = make_int4(-1, -1, -1, -1);
      if (with_smooth) cp = *(const int4*)(fu.ep_pairs + 4 * (size_t)ic);
      const int seg_code[2] = {cs.x, cs.y};
      HgsSegGeom geo[2] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
      if (!CONTRIB) {
#pragma unroll
        for (int s = 0; s < 2; s++) geo[s] = hgs_segment_geom(seg_code[s] >= 0 ? seg_code[s] >> 1 : 0, ep, pairs);
      }
      const int pair_code[4] = {cp.x, cp.y, cp.z, cp.w};
      float pg0[4][3], pg1[4][3];
      bool pok[4] = {false, false, false, false};
      if (CONTRIB && with_smooth && fu.smooth_pair_grads) {
        // the pairs' unit gradients as the forward's spare workgroups left them (HgsStrandFusion.smooth_pair_grads): the role's
        // 16 bytes instead of pair -> index row -> four endpoints; times the scale, as hgs_smooth_pair_grads forms it
        float4 u[4];
#pragma unroll
        for (int s = 0; s < 4; s++) {
          const int code = pair_code[s] >= 0 ? pair_code[s] : 0;
          u[s] = ((const float4*)fu.smooth_pair_grads)[2 * (size_t)(code >> 2) + ((code & 3) >> 1)];
        }
#pragma unroll
        for (int s = 0; s < 4; s++) {
          pok[s] = u[s].w != 0.f && pair_code[s] >= 0;
          pg0[s][0] = pg1[s][0] = smooth_scale * u[s].x; pg0[s][1] = pg1[s][1] = smooth_scale * u[s].y;
          pg0[s][2] = pg1[s][2] = smooth_scale * u[s].z;
        }
      } else if (with_smooth) {
#pragma unroll
        for (int s = 0; s < 4; s++)
          pok[s] = hgs_smooth_pair_grads(pair_code[s] >= 0 ? pair_code[s] >> 2 : 0, ep, fu.smooth_pairs, fu.cos_threshold, fu.eps,
                                         smooth_scale, pg0[s], pg1[s]) && pair_code[s] >= 0;
      }
      // (everything above is independent of the rasterizer's gradients)
      if (CONTRIB) {   // the segments' contributions as the rasterizer backward left them: the same two terms, same order
        float4 ct[2];
#pragma unroll
        for (int s = 0; s < 2; s++) ct[s] = A.seg_contrib[seg_code[s] >= 0 ? seg_code[s] : 0];
#pragma unroll
        for (int s = 0; s < 2; s++)
          if (seg_code[s] >= 0) { acc[0] += ct[s].x; acc[1] += ct[s].y; acc[2] += ct[s].z; }
      } else {
        float sh[2][3], sD[2][3];
#pragma unroll
        for (int s = 0; s < 2; s++) hgs_segment_endpoint_grads(seg_code[s] >= 0 ? seg_code[s] >> 1 : 0, geo[s], f, sg, sh[s], sD[s]);
#pragma unroll
        for (int s = 0; s < 2; s++) {
          const float sign = (seg_code[s] & 1) ? 1.f : -1.f;
          if (seg_code[s] >= 0) {
#pragma unroll
            for (int c = 0; c < 3; c++) acc[c] += sh[s][c] + sign * sD[s][c];
          }
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
      A.d_ep[3 * (size_t)i] = acc[0]; A.d_ep[3 * (size_t)i + 1] = acc[1]; A.d_ep[3 * (size_t)i + 2] = acc[2];
      // Adam in the lane (include/hgs.h HgsAdamSlot): possible because nothing of this launch reads the endpoints any more --
      // segment contributions and pair gradients arrive precomputed (round 4 still noted that the neighbouring lanes re-read them)
      if (CONTRIB) hgs_adam_lane<3>(A.adam.slot[0], (size_t)i, acc, A.adam.beta1, A.adam.beta2, A.adam.eps);
    } else if (i < fu.n_smooth) {   // scatter mode: smoothness gradient added into d_ep with atomics
      hgs_smooth_bwd_pair(i, ep, fu.smooth_pairs, fu.cos_threshold, fu.eps, smooth_scale, A.d_ep);
    }
    return;
  }
  if (CONTRIB) return;
  const int k = blk * 256 + threadIdx.x;
  if (k >= P) return;
  if (fu.radii)                      // densification statistics of this Gaussian (hgs_densify_stats)
    hgs_densify_stats_lane(k, fu.radii[k], fu.dmean2D[(size_t)k * fu.dmean2D_stride], fu.dmean2D[(size_t)k * fu.dmean2D_stride + 1],
                           fu.max_radii2D, fu.grad_accum, fu.denom);
  if (A.d_opacity_raw) { const float o = A.opacity[k]; A.d_opacity_raw[k] = A.g_opacity[k] * o * (1.f - o); }   // sigmoid'
  if (A.d_mask_raw) { const float m = A.extra4[4 * (size_t)k]; A.d_mask_raw[k] = A.g_extra4[4 * (size_t)k] * m * (1.f - m); }
  float gw = 0.f;
  if (A.g_scale) gw = (A.g_scale[3 * (size_t)k + 1] + A.g_scale[3 * (size_t)k + 2]) * expf(A.width[k]);
  if (!fu.ep_segments) {             // scatter mode: this segment's contribution to its two endpoints
    float h[3], gD[3];
    hgs_segment_endpoint_grads(k, hgs_segment_geom(k, ep, pairs), f, sg, h, gD);
    const long long i0 = pairs[2 * (size_t)k], i1 = pairs[2 * (size_t)k + 1];
    float* d_ep = A.d_ep;
    atomicAdd(&d_ep[3 * i0], h[0] - gD[0]); atomicAdd(&d_ep[3 * i0 + 1], h[1] - gD[1]); atomicAdd(&d_ep[3 * i0 + 2], h[2] - gD[2]);
    atomicAdd(&d_ep[3 * i1], h[0] + gD[0]); atomicAdd(&d_ep[3 * i1 + 1], h[1] + gD[1]); atomicAdd(&d_ep[3 * i1 + 2], h[2] + gD[2]);
  }
  A.d_width[k] = gw;
}
