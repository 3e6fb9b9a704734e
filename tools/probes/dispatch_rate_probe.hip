// Workgroup dispatch rate: how long does a launch of N workgroups that do nothing (or one dependent load) take, by
// workgroup size and static LDS?   hipcc --offload-arch=gfx950 -O3 dispatch_rate_probe.hip -o dispatch_probe && ./dispatch_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int LDS_BYTES>
__global__ void k_empty(const int* p, int* out, int do_load) {
  __shared__ char lds[LDS_BYTES > 0 ? LDS_BYTES : 1];
  if (LDS_BYTES > 0 && threadIdx.x == 0 && p[0] == 12345) lds[p[1] & (LDS_BYTES - 1)] = 1;   // (keeps the allocation)
  if (do_load && threadIdx.x == 0 && p[blockIdx.x & 1023] == -7) out[0] = LDS_BYTES > 0 ? lds[0] : 1;
}
template <int LDS_BYTES>
static void run(int wgs, int block, int do_load, const int* p, int* out) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9f;
  for (int rep = 0; rep < 20; rep++) {
    hipEventRecord(a, nullptr);
    for (int k = 0; k < 10; k++) hipLaunchKernelGGL(k_empty<LDS_BYTES>, dim3(wgs), dim3(block), 0, nullptr, p, out, do_load);
    hipEventRecord(b, nullptr);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  printf("wgs %6d  block %4d  lds %6d  load %d : %7.2f us per launch  -> %6.0f workgroups/us  %6.0f waves/us\n", wgs, block, LDS_BYTES,
         do_load, best * 100.f, wgs / (best * 100.f), wgs * ((block + 63) / 64) / (best * 100.f));
}
int main() {
  int *p, *out;
  hipMalloc(&p, 4096); hipMemset(p, 0, 4096); hipMalloc(&out, 64);
  for (int block : {64, 128, 256, 1024})
    for (int wgs : {8192, 32768}) { run<0>(wgs, block, 0, p, out); }
  run<0>(8192, 256, 1, p, out); run<0>(8192, 64, 1, p, out);
  run<4096>(8192, 256, 0, p, out); run<16384>(8192, 256, 0, p, out); run<16384>(8192, 64, 0, p, out); run<32768>(8192, 256, 0, p, out);
  return 0;
}
