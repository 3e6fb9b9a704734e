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

#define HGS_STRAND_MINV 1e-7f

// Gradient of segment k w.r.t. its two endpoints: endpoint 0 receives h - gD, endpoint 1 receives h + gD
// (h = half the gradient of the midpoint, gD = gradient w.r.t. delta = e1 - e0 from direction, quaternion and length).
struct HgsSegGrads { const float* g_xyz; const float* g_scale; const float* g_quat; const float* g_dir; const float* g_extra4; };
struct HgsSegGeom { float dx, dy, dz; };   // e1 - e0 of the segment (independent of the rasterizer's gradients)
__device__ __forceinline__ HgsSegGeom hgs_segment_geom(int k, const float* __restrict__ ep, const long long* __restrict__ pairs) {
  const long long i0 = pairs[2 * (size_t)k], i1 = pairs[2 * (size_t)k + 1];
  return {ep[3 * i1] - ep[3 * i0], ep[3 * i1 + 1] - ep[3 * i0 + 1], ep[3 * i1 + 2] - ep[3 * i0 + 2]};
}
__device__ __forceinline__ void hgs_segment_endpoint_grads(int k, const HgsSegGeom& sgm, float f, const HgsSegGrads& sg, float* h, float* gD) {
  // every load first and unconditional (the pointer tests are uniform): a lane that evaluates several segments then has
  // all of them in flight together instead of one dependent chain after the other
  float4 ge = make_float4(0.f, 0.f, 0.f, 0.f), gq = make_float4(0.f, 0.f, 0.f, 0.f);
  float gx[3] = {0.f, 0.f, 0.f}, gd[3] = {0.f, 0.f, 0.f}, gs0 = 0.f;
  if (sg.g_extra4) ge = *(const float4*)(sg.g_extra4 + 4 * (size_t)k);
  if (sg.g_quat) gq = *(const float4*)(sg.g_quat + 4 * (size_t)k);
  if (sg.g_xyz) { gx[0] = *(sg.g_xyz + 3 * (size_t)k); gx[1] = *(sg.g_xyz + 3 * (size_t)k + 1); gx[2] = *(sg.g_xyz + 3 * (size_t)k + 2); }
  if (sg.g_dir) { gd[0] = *(sg.g_dir + 3 * (size_t)k); gd[1] = *(sg.g_dir + 3 * (size_t)k + 1); gd[2] = *(sg.g_dir + 3 * (size_t)k + 2); }
  if (sg.g_scale) gs0 = *(sg.g_scale + 3 * (size_t)k);
  const float dx = sgm.dx, dy = sgm.dy, dz = sgm.dz;
  const float L = sqrtf(dx * dx + dy * dy + dz * dz);
  h[0] = 0.5f * gx[0]; h[1] = 0.5f * gx[1]; h[2] = 0.5f * gx[2];
  gD[0] = gD[1] = gD[2] = 0.f;
  if (L > HGS_STRAND_MINV) {
    const float il = 1.f / L;
    const float vx = dx * il, vy = dy * il, vz = dz * il;
    float gvx = gd[0], gvy = gd[1], gvz = gd[2];  // gradient w.r.t. the unit direction (L > HGS_STRAND_MINV implies L >= HGS_STRAND_MINV)
    if (sg.g_extra4) { gvx += ge.y; gvy += ge.z; gvz += ge.w; }
    const float n0 = 1.f + vx;
    if (sg.g_quat && n0 > HGS_STRAND_MINV) {
      const float in = 1.f / sqrtf(n0 * n0 + vz * vz + vy * vy);
      const float q0 = n0 * in, q2 = -vz * in, q3 = vy * in;
      const float dot = q0 * gq.x + q2 * gq.z + q3 * gq.w;  // q1 = 0
      const float gn0 = (gq.x - q0 * dot) * in, gn2 = (gq.z - q2 * dot) * in, gn3 = (gq.w - q3 * dot) * in;
      gvx += gn0; gvy += gn3; gvz -= gn2;
    }
    const float vd = vx * gvx + vy * gvy + vz * gvz;
    gD[0] = (gvx - vx * vd) * il; gD[1] = (gvy - vy * vd) * il; gD[2] = (gvz - vz * vd) * il;
    if (sg.g_scale && L / 2.f * f > HGS_STRAND_MINV) {
      const float gs = gs0 * (0.5f * f);
      gD[0] += gs * vx; gD[1] += gs * vy; gD[2] += gs * vz;
    }
  }
}


struct HgsStrandBwdArgs {
  int P; const float* ep; const long long* pairs; const float* width; float f;
  const float* g_xyz; const float* g_scale; const float* g_quat; const float* g_dir;
  float* d_ep; float* d_width;
  const float* opacity; const float* extra4; const float* g_opacity; const float* g_extra4;
  float* d_opacity_raw; float* d_mask_raw;
  HgsStrandFusion fu;
};

// Workgroup `blk` of `nblk` (256 threads): [0, ceil(P / 256)) one lane per Gaussian; then, gather mode, one lane per
// endpoint (scatter mode: per smoothness pair); the last one runs the loss head's deferred tail when that is asked for.
__device__ __forceinline__ void hgs_strand_bwd_block(const HgsStrandBwdArgs& A, unsigned blk, unsigned nblk) {
  const HgsStrandFusion& fu = A.fu;
  const int P = A.P;
  const float* __restrict__ ep = A.ep;
  const long long* __restrict__ pairs = A.pairs;
  const float f = A.f;
  // (the loss head's deferred tail, include/hgs.h HgsHeadTail: one spare workgroup behind the launch's own)
  if (fu.head_tail.out && blk == nblk - 1) { hgs_head_tail_block(fu.head_tail); return; }
  const int nb_seg = (P + 255) / 256;
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
      HgsSegGeom geo[2];
#pragma unroll
      for (int s = 0; s < 2; s++) geo[s] = hgs_segment_geom(seg_code[s] >= 0 ? seg_code[s] >> 1 : 0, ep, pairs);
      const int pair_code[4] = {cp.x, cp.y, cp.z, cp.w};
      float pg0[4][3], pg1[4][3];
      bool pok[4] = {false, false, false, false};
      if (with_smooth) {
#pragma unroll
        for (int s = 0; s < 4; s++)
          pok[s] = hgs_smooth_pair_grads(pair_code[s] >= 0 ? pair_code[s] >> 2 : 0, ep, fu.smooth_pairs, fu.cos_threshold, fu.eps,
                                         smooth_scale, pg0[s], pg1[s]) && pair_code[s] >= 0;
      }
      // (everything above is independent of the rasterizer's gradients)
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
    } else if (i < fu.n_smooth) {   // scatter mode: smoothness gradient added into d_ep with atomics
      hgs_smooth_bwd_pair(i, ep, fu.smooth_pairs, fu.cos_threshold, fu.eps, smooth_scale, A.d_ep);
    }
    return;
  }
  const int k = blk * 256 + threadIdx.x;
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
