// hgs_prologue.h -- device code of the iteration prologue (include/hgs.h hgs_iteration_prologue / HgsPrologue): the view
// select and the clearing of the image buffer's counters.  Shared by the stand-alone launch (hgs_api.hip) and by the
// parameter forward kernels that run it in spare workgroups of the iteration's first launch (hgs_strands.hip).
#pragma once
#include "hgs_common.h"
#include "hgs_adam.h"

__host__ __device__ static inline unsigned hgs_prologue_blocks(size_t zero_words) {
  const size_t b = (zero_words + 1023) / 1024;           // 4 words per thread
  return 1u + (unsigned)(b < 1024 ? b : 1024);
}

// workgroup `wg` of `nwg` (256 threads each): wg 0 copies the view (and the learning rate), the others clear
// `keep`: two consecutive words inside the zero range that are NOT cleared (NULL: none)
__device__ __forceinline__ void hgs_prologue_block(const HgsPrologue& p, unsigned wg, unsigned nwg, const uint32_t* keep = nullptr) {
  if (wg == 0) {
    const uint32_t* src = (const uint32_t*)(p.table + p.view);
    uint32_t* dst = (uint32_t*)p.slot;
    for (int i = threadIdx.x; i < (int)(sizeof(HgsViewTargets) / 4); i += 256) dst[i] = src[i];
    if (threadIdx.x == 0 && p.lr_dst) *p.lr_dst = p.lr;
    if (p.adam_prep) {            // an iteration whose backward updates in its lanes: step counters + coefficients (hgs_adam.h)
      __syncthreads();            // (the position learning rate written above is one of the values read)
      hgs_adam_prepare_block(p.adam_prep);
    }
    return;
  }
  uint32_t* z = (uint32_t*)p.zero_ptr;
  const size_t words = p.zero_bytes / 4, stride = (size_t)(nwg - 1) * 256;
  const uint32_t* keep1 = keep ? keep + 1 : nullptr;
  for (size_t i = (size_t)(wg - 1) * 256 + threadIdx.x; i < words; i += stride)
    if (z + i != keep && z + i != keep1) z[i] = 0u;
}

// hgs_strands.hip: is `func` one of its forward kernels (which take an HgsPrologue as their LAST of *n_params arguments)?
bool hgs_strands_prologue_kernel(const void* func, int* n_params);
// hgs_preprocess.hip: the fused parameters + preprocess kernel (same convention)
bool hgs_preprocess_prologue_kernel(const void* func, int* n_params);
