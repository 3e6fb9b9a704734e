// hgs_strand_fwd.h -- device code of the strand parameters' forward: one segment's Gaussian (mean, scale, quaternion,
// direction) from its two endpoints.  Shared by strand_fwd_kernel (hgs_strands.hip) and the fused parameters + preprocess
// kernel (hgs_preprocess.hip).  The two translation units are built with different contraction flags; the body opts out of
// contraction, so both evaluate it operation by operation and give the same bits (what makes a frame of the fused kernel
// equal to render() on HairGaussianModel.derived_gaussians(), tests/test_gpu_frames.py).
// Formulas: hgs_strands.hip (reference scene/hair_gaussian_model.py:134-201, utils/transform.py:69-86).
#pragma once
#include "hgs_common.h"

struct HgsStrandGaussian {
  float mx, my, mz;        // mean
  float s0, sw;            // scale = (s0, sw, sw)
  float q0, q1, q2, q3;    // rotation (w, x, y, z)
  float ux, uy, uz;        // unit direction
};

__device__ __forceinline__ HgsStrandGaussian hgs_strand_gaussian(float ax, float ay, float az, float bx, float by, float bz,
                                                                 float width_raw, float f) {
#pragma clang fp contract(off)
  constexpr float kMinV = 1e-7f;
  HgsStrandGaussian o;
  o.mx = (ax + bx) / 2.f; o.my = (ay + by) / 2.f; o.mz = (az + bz) / 2.f;
  const float dx = bx - ax, dy = by - ay, dz = bz - az;
  const float L = sqrtf(dx * dx + dy * dy + dz * dz);
  o.sw = expf(width_raw);
  o.s0 = fmaxf(L / 2.f * f, kMinV);
  o.q0 = 1.f; o.q1 = 0.f; o.q2 = 0.f; o.q3 = 0.f; o.ux = 1.f; o.uy = 0.f; o.uz = 0.f;
  if (L > kMinV) {
    const float il = 1.f / L;
    const float vx = dx * il, vy = dy * il, vz = dz * il;
    const float n0 = 1.f + vx;
    if (n0 > kMinV) {
      const float in = 1.f / sqrtf(n0 * n0 + vz * vz + vy * vy);
      o.q0 = n0 * in; o.q1 = 0.f; o.q2 = -vz * in; o.q3 = vy * in;
    } else {  // d = -x_hat: half turn about z
      o.q0 = 0.f; o.q3 = 1.f;
    }
    if (L >= kMinV) { o.ux = vx; o.uy = vy; o.uz = vz; }
  }
  return o;
}

__device__ __forceinline__ float hgs_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }   // gaussian_model.py:93-99

// ---- Stage-I cloud: raw parameters -> rasterizer inputs (scene/gaussian_model.py:118-157); shared by cloud_fwd_kernel and the
// fused parameters + preprocess kernel, evaluated operation by operation in both (see above) ------------------------------
__device__ __forceinline__ int hgs_argmax3(float a, float b, float c) { return (b > a) ? ((c > b) ? 2 : 1) : ((c > a) ? 2 : 0); }
// column `ax` of the rotation matrix of the UNIT quaternion (w, x, y, z)
__device__ __forceinline__ void hgs_rot_column(int ax, float w, float x, float y, float z, float& d0, float& d1, float& d2) {
#pragma clang fp contract(off)
  if (ax == 0)      { d0 = 1.f - 2.f * (y * y + z * z); d1 = 2.f * (x * y + w * z);       d2 = 2.f * (x * z - w * y); }
  else if (ax == 1) { d0 = 2.f * (x * y - w * z);       d1 = 1.f - 2.f * (x * x + z * z); d2 = 2.f * (y * z + w * x); }
  else              { d0 = 2.f * (x * z + w * y);       d1 = 2.f * (y * z - w * x);       d2 = 1.f - 2.f * (x * x + y * y); }
}
struct HgsCloudGaussian { float s0, s1, s2; float4 q; float opacity; float4 extra; };
__device__ __forceinline__ HgsCloudGaussian hgs_cloud_gaussian(float sr0, float sr1, float sr2, float4 r, float o_raw, float m_raw) {
#pragma clang fp contract(off)
  HgsCloudGaussian o;
  o.s0 = expf(sr0); o.s1 = expf(sr1); o.s2 = expf(sr2);
  const float n = sqrtf(r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w);
  const float inq = 1.f / fmaxf(n, 1e-12f);                              // F.normalize (get_rotation)
  o.q = make_float4(r.x * inq, r.y * inq, r.z * inq, r.w * inq);
  const float ib = 1.f / n;                                              // build_rotation normalises without the clamp
  float d0, d1, d2;
  hgs_rot_column(hgs_argmax3(o.s0, o.s1, o.s2), r.x * ib, r.y * ib, r.z * ib, r.w * ib, d0, d1, d2);
  o.opacity = hgs_sigmoid(o_raw);
  o.extra = make_float4(hgs_sigmoid(m_raw), d0, d1, d2);
  return o;
}
