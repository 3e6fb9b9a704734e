// Which float formula gives torch.sigmoid's bits on this PyTorch-ROCm build?  (tools/probes/sigmoid_variants.py)
#include <hip/hip_runtime.h>
extern "C" __global__ void sig_variants(int n, const float* __restrict__ x, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  out[0 * (size_t)n + i] = 1.f / (1.f + expf(-v));
  out[1 * (size_t)n + i] = __builtin_amdgcn_rcpf(1.f + expf(-v));
  out[2 * (size_t)n + i] = 1.f / (1.f + __expf(-v));
  out[3 * (size_t)n + i] = __builtin_amdgcn_rcpf(1.f + __expf(-v));
  out[4 * (size_t)n + i] = __fdividef(1.f, 1.f + expf(-v));
  out[5 * (size_t)n + i] = __fdividef(1.f, 1.f + __expf(-v));
  out[6 * (size_t)n + i] = 1.f / (1.f + exp2f(-v * 1.44269504088896340736f));
  out[7 * (size_t)n + i] = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-v * 1.44269504088896340736f));
}
extern "C" int sig_run(void* stream, int n, const float* x, float* out) {
  hipLaunchKernelGGL(sig_variants, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, x, out);
  return (int)hipGetLastError();
}
