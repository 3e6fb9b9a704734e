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
#include "hgs_adam.h"

namespace {

#define ADAM_MAX_TENSORS 8

// Elements per workgroup: every workgroup ends with a returning atomic on its tensor's ticket, and those serialise -- measured
// (same box) 1024 / 2048 / 4096 / 8192 / 16384 elements: 15.1 / 10.3 / 9.6 / 11.7 / 18.5 us at 100 k strand-Gaussians (0.9 M
// elements: 8192 leaves CUs idle), 109 / 62.6 / 45.2 / 42.7 / 47.2 us at 1 M (9 M elements)
static inline unsigned adam_elems_per_block(unsigned long long total) { return total > (6ull << 20) ? 8192u : 4096u; }

struct AdamTensors {
  float* p[ADAM_MAX_TENSORS];
  const float* g[ADAM_MAX_TENSORS];
  float* m[ADAM_MAX_TENSORS];
  float* v[ADAM_MAX_TENSORS];
  const float* lr[ADAM_MAX_TENSORS];
  float* step[ADAM_MAX_TENSORS];
  unsigned long long numel[ADAM_MAX_TENSORS];
  unsigned int blk_start[ADAM_MAX_TENSORS + 1];   // prefix of the tensors' workgroup counts
  unsigned int vec4;                              // bit k: tensor k's four arrays are 16-byte aligned
  unsigned int elems;                             // elements per workgroup (multiple of 1024)
  int n;
  unsigned int* ticket;                           // [ADAM_MAX_TENSORS] words owned by the optimizer (zero between launches)
};

// fallback ticket words for callers that pass none (tickets == NULL): launches that share them must be stream-ordered
__device__ unsigned int g_adam_ticket[ADAM_MAX_TENSORS] = {};

// (the update rule itself: hgs_adam.h -- shared with the backward kernels that apply it in their own lanes)
#define adam_one hgs_adam_one

// One tensor per workgroup (uniform index: the bias corrections -- two powf -- and the learning rate are evaluated once
// per thread, not once per element), 128-bit accesses where the arrays are aligned.
__global__ __launch_bounds__(256) void adam_kernel(AdamTensors t, float beta1, float beta2, float eps) {
  int k = 0;
#pragma unroll
  for (int j = 1; j < ADAM_MAX_TENSORS; j++) k += (j < t.n && blockIdx.x >= t.blk_start[j]) ? 1 : 0;
  const float step = *t.step[k] + 1.0f;  // (the counters themselves are advanced by the last workgroup to finish, below)
  const HgsAdamCoef cf = hgs_adam_coef(*t.lr[k], step, beta1, beta2);
  const float step_size = cf.step_size, inv_sqrt_bc2 = cf.inv_sqrt_bc2, one_m_b1 = 1.f - beta1;
  const unsigned long long n = t.numel[k];
  const unsigned long long base = (unsigned long long)(blockIdx.x - t.blk_start[k]) * t.elems;
  float* __restrict__ P = t.p[k];
  const float* __restrict__ G = t.g[k];
  float* __restrict__ M = t.m[k];
  float* __restrict__ V = t.v[k];
  if ((t.vec4 >> k) & 1u) {
    // (two float4 of each array in flight per thread and trip; more workgroups of fewer elements cost more than they hide:
    // every workgroup ends with a returning atomic on its tensor's ticket)
#pragma unroll 2
    for (unsigned u = 0; u < t.elems / 1024u; u++) {
      const unsigned long long e = base + (unsigned long long)u * 1024 + threadIdx.x * 4;
      if (e >= n) break;
      if (e + 3 < n) {
        float4 p = *(float4*)(P + e), m = *(float4*)(M + e), v = *(float4*)(V + e);
        const float4 g = *(const float4*)(G + e);
        adam_one(p.x, g.x, m.x, v.x, one_m_b1, beta2, step_size, inv_sqrt_bc2, eps);
        adam_one(p.y, g.y, m.y, v.y, one_m_b1, beta2, step_size, inv_sqrt_bc2, eps);
        adam_one(p.z, g.z, m.z, v.z, one_m_b1, beta2, step_size, inv_sqrt_bc2, eps);
        adam_one(p.w, g.w, m.w, v.w, one_m_b1, beta2, step_size, inv_sqrt_bc2, eps);
        *(float4*)(P + e) = p; *(float4*)(M + e) = m; *(float4*)(V + e) = v;
      } else {
        for (unsigned long long i = e; i < n && i < e + 4; i++) {
          float p = P[i], m = M[i], v = V[i];
          adam_one(p, G[i], m, v, one_m_b1, beta2, step_size, inv_sqrt_bc2, eps);
          P[i] = p; M[i] = m; V[i] = v;
        }
      }
    }
  } else {
    for (unsigned long long i = base + threadIdx.x; i < n && i < base + t.elems; i += 256) {
      float p = P[i], m = M[i], v = V[i];
      adam_one(p, G[i], m, v, one_m_b1, beta2, step_size, inv_sqrt_bc2, eps);
      P[i] = p; M[i] = m; V[i] = v;
    }
  }
  // Step counter of this tensor: the workgroup that takes the tensor's last ticket advances it (a second launch only for
  // that cost 4 us per iteration; one ticket per tensor: a single counter for the whole grid serialised ~1200 atomics on
  // one address, 5 us).  The barrier makes "took a ticket" imply "EVERY wavefront of this workgroup has read the counter"
  // (thread 0 could otherwise run ahead of a sibling wavefront that has not loaded it yet), so the last ticket implies
  // that every workgroup of the tensor has; the next reader is the next launch.  atomicInc wraps the ticket back to 0:
  // nothing to reset between launches.
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned nblk = t.blk_start[k + 1] - t.blk_start[k];
    if (atomicInc(&t.ticket[k], nblk - 1) == nblk - 1) *t.step[k] = step;
  }
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
                  float beta1, float beta2, float eps, unsigned int* tickets) {
  if (n_tensors <= 0) return 0;
  if (n_tensors > ADAM_MAX_TENSORS) { hgs_set_error("hgs_adam_step: at most %d tensors per call", ADAM_MAX_TENSORS); return 1; }
  AdamTensors t;
  t.n = n_tensors;
  t.blk_start[0] = 0;
  t.vec4 = 0;
  unsigned long long total = 0;
  for (int k = 0; k < n_tensors; k++) total += numel[k] > 0 ? (unsigned long long)numel[k] : 0ull;
  t.elems = adam_elems_per_block(total);
  for (int k = 0; k < n_tensors; k++) {
    if (!params[k] || !grads[k] || !exp_avg[k] || !exp_avg_sq[k] || !lr[k] || !step[k] || numel[k] <= 0) {
      hgs_set_error("hgs_adam_step: null pointer or empty tensor %d", k);
      return 1;
    }
    t.p[k] = params[k]; t.g[k] = grads[k]; t.m[k] = exp_avg[k]; t.v[k] = exp_avg_sq[k]; t.lr[k] = lr[k]; t.step[k] = step[k];
    t.numel[k] = (unsigned long long)numel[k];
    t.blk_start[k + 1] = t.blk_start[k] + (unsigned int)((t.numel[k] + t.elems - 1) / t.elems);
    if (!(((size_t)params[k] | (size_t)grads[k] | (size_t)exp_avg[k] | (size_t)exp_avg_sq[k]) & 15)) t.vec4 |= 1u << k;
  }
  for (int k = n_tensors; k < ADAM_MAX_TENSORS; k++) {
    t.p[k] = nullptr; t.g[k] = nullptr; t.m[k] = nullptr; t.v[k] = nullptr; t.lr[k] = nullptr; t.step[k] = nullptr;
    t.numel[k] = 0; t.blk_start[k + 1] = t.blk_start[n_tensors];
  }
  t.ticket = tickets;
  if (!tickets && hipGetSymbolAddress((void**)&t.ticket, HIP_SYMBOL(g_adam_ticket)) != hipSuccess) {
    hgs_set_error("hgs_adam_step: no ticket words");
    return 1;
  }
  hipStream_t s = (hipStream_t)stream;
  {
    HgsProfScope _prof(s, HGS_K_ADAM);
    hipLaunchKernelGGL(adam_kernel, dim3(t.blk_start[n_tensors]), dim3(256), 0, s, t, beta1, beta2, eps);
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
