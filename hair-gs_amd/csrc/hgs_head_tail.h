// hgs_head_tail.h -- device code of the loss head's tail (include/hgs.h HgsHeadTail): the sums over the per-pixel kernel's
// partials and the entries of `out` that depend on them.  Shared by the head's own one-workgroup launch (hgs_losses.hip)
// and by the parameter backward kernels that run it in a spare workgroup of their launch (hgs_strands.hip), off the
// iteration's critical path.  256 threads; the same arithmetic wherever it runs.
#pragma once
#include "hgs_common.h"

__device__ __forceinline__ void hgs_head_tail_block(const HgsHeadTail& t) {
  __shared__ float tail_red[3][4];
  float a[3] = {0.f, 0.f, 0.f};
  // (all loads of a thread are independent: in flight together)
#pragma unroll 8
  for (int i = threadIdx.x; i < t.nb_pix; i += 256) {
#pragma unroll
    for (int c = 0; c < 3; c++) a[c] += t.pix_partials[3 * (size_t)i + c];
  }
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) a[c] += __shfl_xor(a[c], d, 64);
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int c = 0; c < 3; c++) tail_red[c][threadIdx.x >> 6] = a[c];
  __syncthreads();
  if (threadIdx.x != 0) return;
  float s[3];
#pragma unroll
  for (int c = 0; c < 3; c++) s[c] = (tail_red[c][0] + tail_red[c][1]) + (tail_red[c][2] + tail_red[c][3]);
  const float ori_s = s[0], ori_c = s[1], bce_s = s[2];
  float total = t.out[HGS_HEAD_TOTAL_FWD];      // (1 - lambda_dssim) L1 + lambda_dssim DSSIM, from the head's forward (the tail
                                                // may run twice: it must not read what it writes)
  float mask = 0.f, ori = 0.f;
  if (t.bce) { mask = bce_s * t.inv_hw; total = fmaf(t.l_mask, mask, total); }     // (explicit: the same bits in every host kernel)
  if (t.ori) { ori = ori_s / ori_c; total = fmaf(t.l_ori, ori, total); }            // empty mask -> NaN, as the reference
  if (t.smooth) total = fmaf(t.l_smooth, t.out[HGS_HEAD_SMOOTH], total);
  t.out[HGS_HEAD_TOTAL] = total; t.out[HGS_HEAD_MASK] = mask; t.out[HGS_HEAD_ORIENTATION] = ori;
  t.out[HGS_HEAD_ORI_COUNT] = ori_c;
}
