// hgs_smooth.h -- device code of the strand smoothness term (loss/losses.py:175-221), shared by its own kernels
// (hgs_optim.hip) and by the strand parameter kernels that run it in extra workgroups of the same launch (hgs_strands.hip).
#pragma once
#include "hgs_common.h"

__device__ __forceinline__ float hgs_block_sum256(float v, float* red4) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red4[0] + red4[1]) + (red4[2] + red4[3]);
}

struct HgsSmoothEval { float d0[3], d1[3], l0, l1, dot, ang; bool sel; };
// (evaluated without contraction, like the parameter arithmetic of hgs_strand_bwd.h: the term and its gradient are then the
// same bits whichever translation unit's kernel runs them -- the forward rider of hair_preprocess_fwd_kernel, strand_fwd_kernel,
// the endpoint lanes of strand_bwd_kernel)
__device__ __forceinline__ HgsSmoothEval hgs_smooth_eval(const float* __restrict__ ep, const long long* __restrict__ q,
                                                          float cos_th, float eps) {
#pragma clang fp contract(off)
  HgsSmoothEval s;
  float a[3], b[3];
#pragma unroll
  for (int c = 0; c < 3; c++) { a[c] = ep[3 * q[1] + c] - ep[3 * q[0] + c]; b[c] = ep[3 * q[3] + c] - ep[3 * q[2] + c]; }
  s.l0 = sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
  s.l1 = sqrtf(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
#pragma unroll
  for (int c = 0; c < 3; c++) { s.d0[c] = a[c] / s.l0; s.d1[c] = b[c] / s.l1; }
  s.dot = s.d0[0] * s.d1[0] + s.d0[1] * s.d1[1] + s.d0[2] * s.d1[2];
  s.sel = s.dot <= cos_th;                                      // losses.py:211-213
  const float dc = fminf(fmaxf(s.dot, -1.f + eps), 1.f - eps);  // :216-218
  s.ang = acosf(dc);
  return s;
}

// UNSCALED gradients of a pair from its evaluation (dL/d(term) / count = 1): g0 w.r.t. delta a = q[1] - q[0], g1 w.r.t. delta b;
// false (and zeros) if the pair contributes nothing (not selected, or clamp saturated)
__device__ __forceinline__ bool hgs_smooth_unit_grads(const HgsSmoothEval& e, float eps, float* g0, float* g1) {
#pragma clang fp contract(off)
  g0[0] = g0[1] = g0[2] = 0.f; g1[0] = g1[1] = g1[2] = 0.f;
  if (!e.sel) return false;
  if (!(e.dot > -1.f + eps && e.dot < 1.f - eps)) return false;  // clamp saturated: zero gradient
  // d(ang^2)/d(dot) = 2 ang * (-1/sqrt(1-dot^2))
  const float gdot = 2.f * e.ang * (-1.f / sqrtf(1.f - e.dot * e.dot));
#pragma unroll
  for (int c = 0; c < 3; c++) {
    g0[c] = gdot * (e.d1[c] - e.d0[c] * e.dot) / e.l0;          // (I - d0 d0^T) d1 / |a|
    g1[c] = gdot * (e.d0[c] - e.d1[c] * e.dot) / e.l1;
  }
  return true;
}

// forward: block `blk` of 256 pairs -> partials[2*blk] = sum of squared angles of the selected pairs, [2*blk+1] = their count
// pair_grads (may be NULL; [N][2] float4): the pair's unit gradients (g0, ok), (g1, ok) for the endpoint gather of the backward
// (hgs_hair_endpoint_gather: they do not depend on the rasterizer, so the forward's spare workgroups compute them here and the
// backward's endpoint lanes read 16 bytes per role instead of walking pair -> index row -> four endpoints)
__device__ __forceinline__ void hgs_smooth_fwd_block(int blk, int N, const float* __restrict__ ep,
                                                     const long long* __restrict__ idx, float cos_th, float eps,
                                                     float* __restrict__ partials, float* red4, float4* __restrict__ pair_grads = nullptr) {
  const int i = blk * 256 + threadIdx.x;
  float s = 0.f, c = 0.f;
  if (i < N) {
    const HgsSmoothEval e = hgs_smooth_eval(ep, idx + 4 * (size_t)i, cos_th, eps);
    if (e.sel) { s = e.ang * e.ang; c = 1.f; }
    if (pair_grads) {
      float g0[3], g1[3];
      const float ok = hgs_smooth_unit_grads(e, eps, g0, g1) ? 1.f : 0.f;
      pair_grads[2 * (size_t)i] = make_float4(g0[0], g0[1], g0[2], ok);
      pair_grads[2 * (size_t)i + 1] = make_float4(g1[0], g1[1], g1[2], ok);
    }
  }
  const float bs = hgs_block_sum256(s, red4), bc = hgs_block_sum256(c, red4);
  if (threadIdx.x == 0) { partials[2 * blk] = bs; partials[2 * blk + 1] = bc; }
}

// gradients of pair i w.r.t. its two segment deltas (g0 for delta a = q[1]-q[0], g1 for delta b = q[3]-q[2]); false if the
// pair contributes nothing (not selected, or clamp saturated).  scale = dL/d(term) / max(count, 1)
__device__ __forceinline__ bool hgs_smooth_pair_grads(int i, const float* __restrict__ ep, const long long* __restrict__ idx,
                                                      float cos_th, float eps, float scale, float* g0, float* g1) {
  const long long* q = idx + 4 * (size_t)i;
  const HgsSmoothEval e = hgs_smooth_eval(ep, q, cos_th, eps);
  const bool ok = hgs_smooth_unit_grads(e, eps, g0, g1);
#pragma unroll
  for (int c = 0; c < 3; c++) { g0[c] = scale * g0[c]; g1[c] = scale * g1[c]; }   // (the unit gradient times the scale: what a reader of pair_grads forms too)
  return ok;
}

// backward of pair i: scatter into d_ep with fp32 atomics; scale = dL/d(term) / max(count, 1)
__device__ __forceinline__ void hgs_smooth_bwd_pair(int i, const float* __restrict__ ep, const long long* __restrict__ idx,
                                                    float cos_th, float eps, float scale, float* __restrict__ d_ep) {
  const long long* q = idx + 4 * (size_t)i;
  float g0[3], g1[3];
  if (!hgs_smooth_pair_grads(i, ep, idx, cos_th, eps, scale, g0, g1)) return;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    atomicAdd(&d_ep[3 * q[0] + c], -g0[c]); atomicAdd(&d_ep[3 * q[1] + c], g0[c]);
    atomicAdd(&d_ep[3 * q[2] + c], -g1[c]); atomicAdd(&d_ep[3 * q[3] + c], g1[c]);
  }
}
