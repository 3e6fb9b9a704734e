// hgs_pixel_head.h -- the loss head's per-pixel terms (loss/losses.py:224-289 orientation term, :240-248 mask BCE), as device
// functions shared by the two places that evaluate them: pix_fwd_kernel (hgs_losses.hip: a pass over the rendered planes) and the
// blend forward's epilogue (hgs_blend.hip, HgsPixelHead: the pixel's seven channels are still in the registers that blended them).
// One definition, so both forms produce the same bits.
#pragma once
#include "hgs_common.h"

// ---- orientation loss (reference loss/losses.py:224-289) ---------------------------------------------------------
// per pixel: world-space direction image -> view space (x,y) -> unit 2-vector -> angle in [0,pi) w.r.t. the image
// y axis -> bidirectional difference to the GT angle, confidence-weighted, averaged over the mask.
struct OriParams { const float* view; float bg0, bg1, bg2; float min_val; int has_mask; };

__device__ __forceinline__ bool ori_pixel(const OriParams& p, float o0, float o1, float o2, float& px, float& py, float& r,
                                          float& n, float& x, float& y, float& yq, float& theta) {
  const HGS_CONSTANT float* v = hgs_constant(p.view);   // world_view_transform, row-major 4x4 (device, wave-uniform, not written by any launch in flight)
  px = o0 * v[0] + o1 * v[4] + o2 * v[8];      // (flat @ world_view[:3,:3])[:, :2]
  py = o0 * v[1] + o1 * v[5] + o2 * v[9];
  r = sqrtf(px * px + py * py);
  n = r + p.min_val;
  const float in = __builtin_amdgcn_rcpf(n);    // (hardware reciprocals, 1 ulp, here and in the gradient: the per-pixel kernel is
  x = px * in;                                  //  bound by its vector instructions -- 357 per wavefront, 63 % of the pipe -- and an
  y = py * in;                                  //  IEEE division is ten of them)
  yq = y < p.min_val ? y + p.min_val : y;
  theta = atan2f(x, yq);
  if (theta < 0.f) theta += 3.14159265358979323846f;
  return true;
}

struct HeadFlags { int bce, ori; };

// gradient of the orientation term w.r.t. the direction image at one masked pixel, `scale` = dL/d(term) / mask count
__device__ __forceinline__ void ori_pixel_grad(const OriParams& p, float px, float py, float r, float n, float x, float yq,
                                               float th, float gt, float conf, float scale, float& g0, float& g1, float& g2) {
  const float hp = 1.57079632679489661923f;
  const float e = th - gt;
  const float u = fabsf(e) - hp;
  const float sg = (u > 0.f ? 1.f : (u < 0.f ? -1.f : 0.f)) * (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f));
  const float dth = -sg * conf * scale;                               // dL/dtheta
  const float iden = __builtin_amdgcn_rcpf(x * x + yq * yq);
  const float dx = dth * (yq * iden), dy = dth * (-x * iden);         // atan2(x, yq)
  // x = px/n, y = py/n, n = r + eps.  r = 0 (a masked pixel nothing was blended into): torch's norm has the subgradient 0
  // there and the direct 1 / n path stays -- the reference's gradient at such a pixel is ~conf / (count eps^2), and so is this one
  const float inv_n = __builtin_amdgcn_rcpf(n), inv_n2 = inv_n * inv_n, ir = r > 0.f ? __builtin_amdgcn_rcpf(r) : 0.f;
  const float dn = -(dx * px + dy * py) * inv_n2;
  const float dpx = dx * inv_n + dn * px * ir, dpy = dy * inv_n + dn * py * ir;
  const HGS_CONSTANT float* v = hgs_constant(p.view);
  g0 = dpx * v[0] + dpy * v[1]; g1 = dpx * v[4] + dpy * v[5]; g2 = dpx * v[8] + dpy * v[9];
}

// One pixel of the head: in = the blended mask logit and direction, the view's targets at the pixel; out = the pixel's
// share of the three sums (orientation term, its pixel count, BCE) and, with `grad`, dL/d(mask logit, direction) for an upstream
// gradient of 1 (g_mask = l_mask / HW, ori_scale = l_orientation / mask count).
struct HgsPixelIn { float xm, ym, o0, o1, o2, gt, cf; unsigned char mk; };
struct HgsPixelOut { float s, cnt, b, gm, g0, g1, g2; };
__device__ __forceinline__ void hgs_pixel_terms(const HeadFlags& fl, const OriParams& p, const HgsPixelIn& in, bool grad,
                                                float g_mask, float ori_scale, HgsPixelOut& o) {
  o.s = 0.f; o.cnt = 0.f; o.b = 0.f; o.gm = 0.f; o.g0 = 0.f; o.g1 = 0.f; o.g2 = 0.f;
  if (fl.bce) {
    const float x = in.xm, y = in.ym;
    // (hardware exp2 / log2: en in (0, 1], so log(1 + en) is within 1e-7 absolute of log1p(en) -- of a term of order 0.1-1)
    const float en = __expf(-fabsf(x));
    o.b = fmaxf(x, 0.f) - x * y + __logf(1.f + en);
    if (grad) { const float r1 = __builtin_amdgcn_rcpf(1.f + en); o.gm = g_mask * ((x >= 0.f ? r1 : en * r1) - y); }   // sigmoid(x) - y
  }
  if (fl.ori) {
    const bool m = p.has_mask ? in.mk != 0 : (in.o0 != p.bg0 || in.o1 != p.bg1 || in.o2 != p.bg2);
    if (m) {
      float px, py, r, n, x, y, yq, th;
      ori_pixel(p, in.o0, in.o1, in.o2, px, py, r, n, x, y, yq, th);
      const float hp = 1.57079632679489661923f;
      o.s = (hp - fabsf(fabsf(th - in.gt) - hp)) * in.cf;
      o.cnt = 1.f;
      if (grad) ori_pixel_grad(p, px, py, r, n, x, yq, th, in.gt, in.cf, ori_scale, o.g0, o.g1, o.g2);
    }
  }
}
