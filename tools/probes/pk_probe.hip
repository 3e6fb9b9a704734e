// pk_probe.hip -- does v_pk_fma_f32 issue at the rate of v_fma_f32 on this part?  (tools/probes: measurement aids, not product)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/pk_probe.hip -o /tmp/pk_probe && /tmp/pk_probe
// Every wave runs N dependent-free FMA streams (8 accumulators) for ITER iterations; occupancy 8 waves/SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
#define ITER 4096
__global__ void k_scalar(float* out, float a, float b) {
  float x[16];
  for (int i = 0; i < 16; i++) x[i] = threadIdx.x * 1e-3f + i;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = __builtin_fmaf(x[i], a, b);
  }
  float s = 0;
  for (int i = 0; i < 16; i++) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_packed(float* out, float a, float b) {
  v2f x[8];
  for (int i = 0; i < 8; i++) x[i] = v2f{threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f + i};
  const v2f av = {a, a}, bv = {b, b};
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(av), "v"(bv));
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += x[i].x + x[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_scalar_asm(float* out, float a, float b) {
  float x[16];
  for (int i = 0; i < 16; i++) x[i] = threadIdx.x * 1e-3f + i;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
  }
  float s = 0;
  for (int i = 0; i < 16; i++) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float* out;
  hipMalloc(&out, 256 * 8 * 256 * 4 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 8;  // 8 workgroups of 256 threads per CU: 8 waves per SIMD
  for (int variant = 0; variant < 3; variant++) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
      hipEventRecord(e0);
      if (variant == 0) hipLaunchKernelGGL(k_scalar_asm, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
      if (variant == 1) hipLaunchKernelGGL(k_packed, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
      if (variant == 2) hipLaunchKernelGGL(k_scalar, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const double fma = (double)blocks * 256 * ITER * 16;
    printf("%s: %.3f ms  %.1f TFLOP/s (fma = 2 flop)\n", variant == 0 ? "v_fma_f32   " : variant == 1 ? "v_pk_fma_f32" : "compiler    ", best, 2 * fma / best / 1e9);
  }
  return 0;
}
