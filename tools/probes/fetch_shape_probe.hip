// FETCH_SIZE calibration for the blend kernels' access shapes (MI355X_MICROARCH.md, HBM: "on gfx950 FETCH_SIZE reports exactly
// 1/2 of the bytes of a wide coalesced streaming read (16 B/lane) ... other access widths are uncalibrated: calibrate on a
// known byte count in your own access pattern").  Every kernel below reads a KNOWN number of bytes, once, from buffers far
// larger than the 32 MB of L2 in total; run under
//     rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- ./fetch_shape_probe
// (tools/probes/fetch_shape_probe.sh) and divide the counter by the byte count the program prints per kernel.
//   k_quadrant_dword  the blend backward's per-pixel planes: 9 planar [H, W] fp32 images, one workgroup per 16x16 tile, a
//                     wavefront = an 8x8 pixel quadrant, ONE DWORD PER LANE (eight 32-byte row segments per wave instruction)
//   k_quadrant_sparse the same on every second tile only (37 % of a north_star frame's tiles are empty and never read)
//   k_row_dword       one dword per lane, 64 consecutive floats per wave instruction (256 contiguous bytes)
//   k_stream16        16 bytes per lane, consecutive (the guide's calibrated case: the instance records are read like this)
//   k_record_gather   64-byte records gathered by index, 16 bytes per lane, 4 lanes per record (the sort kernel's template gather)
//   hipcc --offload-arch=gfx950 -O3 fetch_shape_probe.hip -o fetch_shape_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define W 1920
#define H 1080
#define PLANES 9
#define GX (W / 16)
#define GY ((H + 15) / 16)

__global__ __launch_bounds__(256) void k_quadrant_dword(const float* __restrict__ img, float* __restrict__ sink, int every) {
  const int tile = blockIdx.x * every;
  const int tx = tile % GX, ty = tile / GX;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int px = tx * 16 + (wave & 1) * 8 + (lane & 7), py = ty * 16 + (wave >> 1) * 8 + (lane >> 3);
  float s = 0.f;
  if (py < H) {
#pragma unroll
    for (int k = 0; k < PLANES; k++) s += img[(size_t)k * W * H + (size_t)py * W + px];
  }
  if (s == 12345.678f) sink[0] = s;
}
__global__ __launch_bounds__(256) void k_row_dword(const float* __restrict__ img, float* __restrict__ sink, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  float s = 0.f;
  if (i < n) s = img[i];
  if (s == 12345.678f) sink[0] = s;
}
__global__ __launch_bounds__(256) void k_stream16(const float4* __restrict__ img, float* __restrict__ sink, size_t n4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) v = img[i];
  if (v.x + v.y + v.z + v.w == 12345.678f) sink[0] = v.x;
}
__global__ __launch_bounds__(256) void k_record_gather(const float4* __restrict__ rec, const unsigned* __restrict__ idx, float* __restrict__ sink, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;   // float4 i of the output = quarter (i & 3) of record idx[i >> 2]
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if ((i >> 2) < n) v = rec[(size_t)idx[i >> 2] * 4 + (i & 3)];
  if (v.x + v.y + v.z + v.w == 12345.678f) sink[0] = v.x;
}

int main() {
  const size_t npix = (size_t)W * H, nimg = npix * PLANES;
  float *img, *sink, *big;
  hipMalloc(&img, nimg * 4); hipMalloc(&sink, 64);
  const size_t nbig = (size_t)64 << 20;            // 256 MB of floats for the streaming kernels
  hipMalloc(&big, nbig * 4);
  hipMemset(img, 0, nimg * 4); hipMemset(big, 0, nbig * 4);
  const size_t nrec = 1 << 20;                     // 64 MB of records, every one gathered once in random order
  std::vector<unsigned> perm(nrec);
  for (size_t i = 0; i < nrec; i++) perm[i] = (unsigned)i;
  srand(1);
  for (size_t i = nrec - 1; i > 0; i--) { size_t j = (size_t)rand() % (i + 1); unsigned t = perm[i]; perm[i] = perm[j]; perm[j] = t; }
  unsigned* idx; hipMalloc(&idx, nrec * 4); hipMemcpy(idx, perm.data(), nrec * 4, hipMemcpyHostToDevice);
  hipDeviceSynchronize();
  const size_t flush_n4 = nbig / 4;
  for (int rep = 0; rep < 3; rep++) {
    // a 256 MB stream between the measured kernels pushes the previous one's lines out of L2 and most of the Infinity Cache
    hipLaunchKernelGGL(k_stream16, dim3((unsigned)((flush_n4 + 255) / 256)), dim3(256), 0, nullptr, (const float4*)big, sink, flush_n4);
    hipLaunchKernelGGL(k_quadrant_dword, dim3(GX * GY), dim3(256), 0, nullptr, img, sink, 1);
    hipLaunchKernelGGL(k_stream16, dim3((unsigned)((flush_n4 + 255) / 256)), dim3(256), 0, nullptr, (const float4*)big, sink, flush_n4);
    hipLaunchKernelGGL(k_quadrant_dword, dim3(GX * GY / 2), dim3(256), 0, nullptr, img, sink, 2);
    hipLaunchKernelGGL(k_stream16, dim3((unsigned)((flush_n4 + 255) / 256)), dim3(256), 0, nullptr, (const float4*)big, sink, flush_n4);
    hipLaunchKernelGGL(k_row_dword, dim3((unsigned)((nimg + 255) / 256)), dim3(256), 0, nullptr, img, sink, nimg);
    hipLaunchKernelGGL(k_stream16, dim3((unsigned)((flush_n4 + 255) / 256)), dim3(256), 0, nullptr, (const float4*)big, sink, flush_n4);
    hipLaunchKernelGGL(k_record_gather, dim3((unsigned)((nrec * 4 + 255) / 256)), dim3(256), 0, nullptr, (const float4*)big, idx, sink, nrec);
  }
  hipDeviceSynchronize();
  // bytes each kernel reads per launch (rows of the last tile row beyond H are not read: H = 1080 = 67.5 tiles)
  printf("k_quadrant_dword %zu\n", npix * PLANES * 4);
  {
    size_t pixels = 0;
    for (int t = 0; t < GX * GY; t += 2) { const int ty = t / GX; const int rows = (ty * 16 + 16 <= H) ? 16 : H - ty * 16; pixels += (size_t)rows * 16; }
    printf("k_quadrant_sparse %zu\n", pixels * PLANES * 4);
  }
  printf("k_row_dword %zu\n", nimg * 4);
  printf("k_stream16 %zu\n", nbig * 4);
  printf("k_record_gather %zu (+ %zu of indices)\n", nrec * 64, nrec * 4);
  return 0;
}
