// hgs_optim.hip -- (1) Adam over all parameter groups in ONE launch; (2) strand smoothness loss forward/backward.
//
// (1) replaces torch.optim.Adam's per-group multi-tensor kernels (6 groups x ~42 us on MI355X for ~1 MB of state)
//     for the optimizer the reference builds at scene/gaussian_model.py:250 / scene/hair_gaussian_model.py:246
//     (Adam, eps = 1e-15, betas (0.9, 0.999), no weight decay, one group per tensor, per-group lr).
//     Update rule = torch's: m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).
//     lr and the step counter t live in device memory (one fp32 scalar each per tensor) so the launch is graph-replayable.
// (2) replaces the gather / normalise / acos chain and, above all, its index_put backward (sort-based, ~0.25 ms per
//     iteration) of loss/losses.py:175-221 angle_smoothness_loss.
#include "hgs_common.h"
#include "hgs_smooth.h"

namespace {

#define ADAM_MAX_TENSORS 8

struct AdamTensors {
  float* p[ADAM_MAX_TENSORS];
  const float* g[ADAM_MAX_TENSORS];
  float* m[ADAM_MAX_TENSORS];
  float* v[ADAM_MAX_TENSORS];
  const float* lr[ADAM_MAX_TENSORS];
  float* step[ADAM_MAX_TENSORS];
  unsigned long long start[ADAM_MAX_TENSORS + 1];  // prefix of element counts
  int n;
};

__global__ __launch_bounds__(256) void adam_kernel(AdamTensors t, float beta1, float beta2, float eps) {
  const unsigned long long total = t.start[t.n];
  const unsigned long long stride = (unsigned long long)gridDim.x * 256;
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    int k = 0;
#pragma unroll
    for (int j = 1; j < ADAM_MAX_TENSORS; j++) k += (j < t.n && i >= t.start[j]) ? 1 : 0;
    const unsigned long long e = i - t.start[k];
    const float step = *t.step[k] + 1.0f;  // the counter itself is advanced by adam_step_kernel afterwards
    const float g = t.g[k][e];
    const float m = t.m[k][e] + (g - t.m[k][e]) * (1.f - beta1);        // lerp, like torch
    const float v = beta2 * t.v[k][e] + (1.f - beta2) * g * g;
    const float bc1 = 1.f - powf(beta1, step), bc2 = 1.f - powf(beta2, step);
    const float step_size = *t.lr[k] / bc1;
    const float denom = sqrtf(v) / sqrtf(bc2) + eps;
    t.m[k][e] = m;
    t.v[k][e] = v;
    t.p[k][e] -= step_size * (m / denom);
  }
}
__global__ void adam_step_kernel(AdamTensors t) {
  if (threadIdx.x < t.n) *t.step[threadIdx.x] += 1.0f;
}

// ---- smoothness (device code in hgs_smooth.h) -------------------------------------------------------------------------
__global__ __launch_bounds__(256) void smooth_fwd_kernel(int N, const float* __restrict__ ep, const long long* __restrict__ idx,
                                                         float cos_th, float eps, float* __restrict__ partials) {
  __shared__ float red[4];
  hgs_smooth_fwd_block(blockIdx.x, N, ep, idx, cos_th, eps, partials, red);
}

__global__ __launch_bounds__(256) void smooth_bwd_kernel(int N, const float* __restrict__ ep, const long long* __restrict__ idx,
                                                         float cos_th, float eps, const float* __restrict__ g_loss,
                                                         const float* __restrict__ count, const float* __restrict__ go,
                                                         float* __restrict__ d_ep) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  hgs_smooth_bwd_pair(i, ep, idx, cos_th, eps, *g_loss * (go ? *go : 1.f) / fmaxf(*count, 1.f), d_ep);
}

}  // namespace

// launchers shared with the loss head (hgs_losses.hip)
int hgs_launch_smooth_fwd(hipStream_t s, int N, const float* endpoints, const long long* index_pairs, float cos_th, float eps,
                          float* partials) {
  HgsProfScope _prof(s, HGS_K_SMOOTH);
  hipLaunchKernelGGL(smooth_fwd_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, endpoints, index_pairs, cos_th, eps, partials);
  return 0;
}
int hgs_launch_smooth_bwd(hipStream_t s, int N, const float* endpoints, const long long* index_pairs, float cos_th, float eps,
                          const float* g_loss, const float* count, const float* go, float* d_endpoints) {
  HgsProfScope _prof(s, HGS_K_SMOOTH);
  hipLaunchKernelGGL(smooth_bwd_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, endpoints, index_pairs, cos_th, eps, g_loss,
                     count, go, d_endpoints);
  return 0;
}

extern "C" {

int hgs_adam_step(void* stream, int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                  float* const* exp_avg_sq, const float* const* lr, float* const* step, const long long* numel,
                  float beta1, float beta2, float eps) {
  if (n_tensors <= 0) return 0;
  if (n_tensors > ADAM_MAX_TENSORS) { hgs_set_error("hgs_adam_step: at most %d tensors per call", ADAM_MAX_TENSORS); return 1; }
  AdamTensors t;
  t.n = n_tensors;
  t.start[0] = 0;
  for (int k = 0; k < n_tensors; k++) {
    if (!params[k] || !grads[k] || !exp_avg[k] || !exp_avg_sq[k] || !lr[k] || !step[k] || numel[k] <= 0) {
      hgs_set_error("hgs_adam_step: null pointer or empty tensor %d", k);
      return 1;
    }
    t.p[k] = params[k]; t.g[k] = grads[k]; t.m[k] = exp_avg[k]; t.v[k] = exp_avg_sq[k]; t.lr[k] = lr[k]; t.step[k] = step[k];
    t.start[k + 1] = t.start[k] + (unsigned long long)numel[k];
  }
  for (int k = n_tensors; k < ADAM_MAX_TENSORS; k++) { t.p[k] = nullptr; t.g[k] = nullptr; t.m[k] = nullptr; t.v[k] = nullptr; t.lr[k] = nullptr; t.step[k] = nullptr; t.start[k + 1] = t.start[n_tensors]; }
  hipStream_t s = (hipStream_t)stream;
  const unsigned long long total = t.start[n_tensors];
  const unsigned blocks = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  {
    HgsProfScope _prof(s, HGS_K_ADAM);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, s, t, beta1, beta2, eps);
    hipLaunchKernelGGL(adam_step_kernel, dim3(1), dim3(64), 0, s, t);
  }
  HGS_CHECK_LAUNCH();
  return 0;
}

int hgs_smoothness_num_blocks(int N) { return (N + 255) / 256; }

int hgs_smoothness_forward(void* stream, int N, const float* endpoints, const long long* index_pairs, float cos_threshold,
                           float eps, float* partials) {
  if (N <= 0) return 0;
  if (!endpoints || !index_pairs || !partials) { hgs_set_error("hgs_smoothness_forward: null argument"); return 1; }
  hipStream_t s = (hipStream_t)stream;
  {
    HgsProfScope _prof(s, HGS_K_SMOOTH);
    hipLaunchKernelGGL(smooth_fwd_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, endpoints, index_pairs, cos_threshold, eps, partials);
  }
  HGS_CHECK_LAUNCH();
  return 0;
}

int hgs_smoothness_backward(void* stream, int N, int E, const float* endpoints, const long long* index_pairs,
                            float cos_threshold, float eps, const float* g_loss, const float* count, float* d_endpoints) {
  if (!d_endpoints || !g_loss || !count) { hgs_set_error("hgs_smoothness_backward: null argument"); return 1; }
  hipStream_t s = (hipStream_t)stream;
  if (hgs_zero_async(s, d_endpoints, (size_t)E * 3 * sizeof(float))) return 1;
  if (N <= 0) return 0;
  {
    HgsProfScope _prof(s, HGS_K_SMOOTH);
    hipLaunchKernelGGL(smooth_bwd_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, endpoints, index_pairs, cos_threshold, eps, g_loss, count, (const float*)nullptr, d_endpoints);
  }
  HGS_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
