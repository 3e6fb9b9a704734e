// hgs_adam.h -- device code of the Adam update (torch.optim.Adam as the reference builds it: scene/gaussian_model.py:250,
// scene/hair_gaussian_model.py:246; eps 1e-15, betas (0.9, 0.999), no weight decay), shared by
//   * adam_kernel (hgs_optim.hip, hgs_adam_step): every parameter tensor in one launch, and
//   * the backward's own lanes (round 5, include/hgs.h HgsAdamSlot): the lane that has just finished the gradient of a parameter
//     element applies that element's update at once -- preprocess_bwd_kernel's per-Gaussian lanes (width / opacity / mask / SH DC of
//     a strand model; xyz / scaling / rotation / opacity / mask / SH DC of a cloud) and the endpoint lanes of strand_gather_kernel --
//     so that a single-rank iteration has no optimizer launch at all.
// Evaluated operation by operation (no contraction) in every translation unit: the two forms give the same bits, a run may mix
// them (captured replays with the in-lane update, eager topology iterations with the launch).
#pragma once
#include "hgs_common.h"

// what one optimizer step needs of a tensor's scalars: lr / (1 - beta1^t) and 1 / sqrt(1 - beta2^t), t = the step being taken
struct HgsAdamCoef { float step_size, inv_sqrt_bc2; };
__device__ __forceinline__ HgsAdamCoef hgs_adam_coef(float lr, float step, float beta1, float beta2) {
#pragma clang fp contract(off)
  const float bc1 = 1.f - powf(beta1, step), bc2 = 1.f - powf(beta2, step);
  HgsAdamCoef c;
  c.step_size = lr / bc1;
  c.inv_sqrt_bc2 = 1.f / sqrtf(bc2);
  return c;
}
// m = b1 m + (1 - b1) g (as a lerp, like torch); v = b2 v + (1 - b2) g^2; p -= step_size * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
__device__ __forceinline__ void hgs_adam_one(float& p, float g, float& m, float& v, float one_m_b1, float beta2, float step_size,
                                             float inv_sqrt_bc2, float eps) {
#pragma clang fp contract(off)
  m = m + (g - m) * one_m_b1;
  v = beta2 * v + (1.f - beta2) * g * g;
  p -= step_size * (m / (sqrtf(v) * inv_sqrt_bc2 + eps));
}

// The in-lane update of element(s) [i * N, (i + 1) * N) of the tensor behind `s` with the gradient values g[0..N).
// (p, m, v of the element are loaded here, behind the gradient arithmetic: one more round trip at the end of the lane, no
// registers held across it.  Requested EARLY instead -- with the lane's other independent loads, 18 more live registers in
// preprocess_bwd_kernel -- the step was slower at every size, same box: north_star 4196 -> 4181 it/s, C3 3102 -> 3061, C4 1006 -> 992.)
template <int N>
__device__ __forceinline__ void hgs_adam_lane(const HgsAdamSlot& s, size_t i, const float* g, float beta1, float beta2, float eps) {
  if (!s.p) return;
  const float step_size = s.coef[0], inv_sqrt_bc2 = s.coef[1], one_m_b1 = 1.f - beta1;
  float p[N], m[N], v[N];
#pragma unroll
  for (int k = 0; k < N; k++) { p[k] = s.p[i * N + k]; m[k] = s.m[i * N + k]; v[k] = s.v[i * N + k]; }
#pragma unroll
  for (int k = 0; k < N; k++) hgs_adam_one(p[k], g[k], m[k], v[k], one_m_b1, beta2, step_size, inv_sqrt_bc2, eps);
#pragma unroll
  for (int k = 0; k < N; k++) { s.p[i * N + k] = p[k]; s.m[i * N + k] = m[k]; s.v[i * N + k] = v[k]; }
}

// Start of an iteration whose backward updates in its lanes (rides with the prologue, hgs_prologue.h): thread k advances tensor
// k's step counter and leaves its two coefficients where the lanes read them.  `lr` of the position group is written by the
// same workgroup just before (HgsPrologue.lr_dst): the caller puts a barrier in between.
__device__ __forceinline__ void hgs_adam_prepare_block(const HgsAdamPrep* prep) {
  const int k = (int)threadIdx.x;
  if (!prep || k >= prep->n) return;
  const float step = *prep->step[k] + 1.0f;
  *prep->step[k] = step;
  const HgsAdamCoef c = hgs_adam_coef(*prep->lr[k], step, prep->beta1, prep->beta2);
  prep->coef[2 * k] = c.step_size;
  prep->coef[2 * k + 1] = c.inv_sqrt_bc2;
}
