// atomic_scope_probe.hip -- cost of counter atomics as the binning kernels issue them (tools/probes: measurement aid).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/atomic_scope_probe.hip -o /tmp/atomic_probe && /tmp/atomic_probe
// G workgroups each add to K pseudo-random counters of a T-entry table (one add per lane and instruction, no return):
//   agent : device-scope atomics on ONE table (what preprocess_fwd_kernel's count publish does)
//   wg    : workgroup-scope atomics on a table per XCD (index = HW_REG_XCC_ID), summed afterwards
// Prints the time per variant and whether the per-XCD tables add up to the same totals.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }
__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int MODE>
__global__ void k(unsigned* tab, int T, int K, int hot) {
  const unsigned x = MODE == 1 ? xcc_id() : 0u;
  unsigned* t = tab + (size_t)x * T;
  for (int j = 0; j < K; j++) {
    // a workgroup's tiles: a window of `hot` consecutive tiles somewhere in the table (strand-like locality)
    const unsigned base = hash(blockIdx.x * 977u) % (unsigned)(T - hot);
    const unsigned i = base + hash(blockIdx.x * 131071u + threadIdx.x * 31u + j) % (unsigned)hot;
    if (MODE == 1) __hip_atomic_fetch_add(&t[i], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_fetch_add(&t[i], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// the publish loop of preprocess_fwd_kernel: a 1024-slot workgroup table with `valid` occupied slots (hash order), four
// slots per thread, one atomic per occupied slot
__global__ void k_sparse(unsigned* tab, int T, int hot, int valid) {
  __shared__ unsigned key[1024], cnt[1024];
  for (int i = threadIdx.x; i < 1024; i += 256) { key[i] = 0xFFFFFFFFu; cnt[i] = 0u; }
  __syncthreads();
  const unsigned base = hash(blockIdx.x * 977u) % (unsigned)(T - hot);
  if ((int)threadIdx.x < valid) {
    const unsigned t = base + hash(blockIdx.x * 131071u + threadIdx.x) % (unsigned)hot;
    unsigned s = (t * 2654435761u) >> 22;
    for (int k = 0; k < 8; k++) { const unsigned prev = atomicCAS(&key[s], 0xFFFFFFFFu, t); if (prev == 0xFFFFFFFFu || prev == t) { atomicAdd(&cnt[s], 1u); break; } s = (s + 1) & 1023; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 256)
    if (key[i] != 0xFFFFFFFFu) atomicAdd(&tab[key[i]], cnt[i]);
}
int main() {
  const int T = 8160, G = 3906, K = 1;
  unsigned* tab;
  CHECK(hipMalloc(&tab, sizeof(unsigned) * T * 16));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  for (int hot : {40, 400, 8000}) {
    std::vector<unsigned> ref(T), got(T * 16);
    for (int mode = 0; mode < 2; mode++) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; rep++) {
        CHECK(hipMemset(tab, 0, sizeof(unsigned) * T * 16));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(a));
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(G), dim3(64), 0, 0, tab, T, K, hot);
        else hipLaunchKernelGGL(k<1>, dim3(G), dim3(64), 0, 0, tab, T, K, hot);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
      }
      CHECK(hipMemcpy(got.data(), tab, sizeof(unsigned) * T * 16, hipMemcpyDeviceToHost));
      bool ok = true;
      if (mode == 0) for (int i = 0; i < T; i++) ref[i] = got[i];
      else for (int i = 0; i < T; i++) { unsigned s = 0; for (int x = 0; x < 16; x++) s += got[x * T + i]; ok &= s == ref[i]; }
      printf("window %5d tiles  %s: %7.2f us for %d wavefront atomics (64 lanes each)%s\n", hot, mode ? "workgroup scope, table per XCD" : "agent scope, one table        ",
             best * 1e3f, G * K, mode ? (ok ? "  [sums agree]" : "  [SUMS DIFFER]") : "");
    }
  }
  for (int hot : {40, 400}) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
      CHECK(hipMemset(tab, 0, sizeof(unsigned) * T * 16));
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(a));
      hipLaunchKernelGGL(k_sparse, dim3(G), dim3(256), 0, 0, tab, T, hot, 40);
      CHECK(hipEventRecord(b));
      CHECK(hipEventSynchronize(b));
      float ms; CHECK(hipEventElapsedTime(&ms, a, b));
      if (ms < best) best = ms;
    }
    printf("window %5d tiles  sparse publish from a 1024-slot table (40 tiles per workgroup, %d workgroups): %7.2f us\n", hot, G, best * 1e3f);
  }
  return 0;
}
