/*
 * knn_oracle.c -- CPU restatement of simple-knn's distCUDA2 (mean squared distance to the 3
 * nearest neighbours, Morton-sorted boxes of 1024 points).
 *
 * TEST INFRASTRUCTURE ONLY (see raster_oracle.c header).  "Parity unpinned": the reference
 * ships no test or golden vector for this function and its CUDA source cannot be built here;
 * the independent pin is a brute-force O(P^2) 3-NN (tests/test_oracle_checks.py) -- the box
 * pruning of the reference is exact, so both must agree to the last bit of the fp32 distances.
 *
 * Follows /root/reference/submodules/simple-knn/simple_knn.cu (SK/) line by line:
 *   prepMorton / coord2Morton  SK/simple_knn.cu:46-71
 *   boxMinMax                  SK/simple_knn.cu:79-118
 *   distBoxPoint / updateKBest SK/simple_knn.cu:120-146
 *   boxMeanDist                SK/simple_knn.cu:148-184
 *   SimpleKNN::knn (host)      SK/simple_knn.cu:186-222  (min/max reduce with init (0,0,0) :192)
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BOX_SIZE 1024 /* SK/simple_knn.cu:12 */

static uint32_t prepMorton(uint32_t x) {
  x = (x | (x << 16)) & 0x030000FF;
  x = (x | (x << 8)) & 0x0300F00F;
  x = (x | (x << 4)) & 0x030C30C3;
  x = (x | (x << 2)) & 0x09249249;
  return x;
}
static uint32_t f2u(float v) { /* GPU float->uint: saturating, NaN -> 0 */
  if (v != v || v <= 0.f) return 0;
  if (v >= 4294967040.f) return 0xFFFFFFFFu;
  return (uint32_t)v;
}
static uint32_t coord2Morton(const float* c, const float* mn, const float* mx) {
  uint32_t x = prepMorton(f2u(((c[0] - mn[0]) / (mx[0] - mn[0])) * ((1 << 10) - 1)));
  uint32_t y = prepMorton(f2u(((c[1] - mn[1]) / (mx[1] - mn[1])) * ((1 << 10) - 1)));
  uint32_t z = prepMorton(f2u(((c[2] - mn[2]) / (mx[2] - mn[2])) * ((1 << 10) - 1)));
  return x | (y << 1) | (z << 2);
}
typedef struct { float mn[3], mx[3]; } MinMax;

static float distBoxPoint(const MinMax* b, const float* p) {
  float d[3] = {0, 0, 0};
  for (int k = 0; k < 3; k++)
    if (p[k] < b->mn[k] || p[k] > b->mx[k]) d[k] = fminf(fabsf(p[k] - b->mn[k]), fabsf(p[k] - b->mx[k]));
  return d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
}
static void updateKBest3(const float* ref, const float* pt, float* knn) {
  float d[3] = {pt[0] - ref[0], pt[1] - ref[1], pt[2] - ref[2]};
  float dist = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
  for (int j = 0; j < 3; j++)
    if (knn[j] > dist) { float t = knn[j]; knn[j] = dist; dist = t; }
}

/* out_codes / out_indices may be NULL; when given they receive the Morton codes (input order)
 * and the Morton-sorted index permutation (stable), for bit-exact checks of the GPU stages. */
void hgs_oracle_dist2(int P, const float* points, float* mean_dists, uint32_t* out_codes, uint32_t* out_indices) {
  if (P <= 0) return;
  float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0}; /* init = {0,0,0}: the box always contains the origin */
  for (int i = 0; i < P; i++)
    for (int k = 0; k < 3; k++) {
      mn[k] = fminf(mn[k], points[3 * i + k]);
      mx[k] = fmaxf(mx[k], points[3 * i + k]);
    }
  uint32_t* codes = (uint32_t*)malloc((size_t)P * 4);
  uint32_t* idx = (uint32_t*)malloc((size_t)P * 4);
  uint32_t* tmpc = (uint32_t*)malloc((size_t)P * 4);
  uint32_t* tmpi = (uint32_t*)malloc((size_t)P * 4);
  for (int i = 0; i < P; i++) { codes[i] = coord2Morton(points + 3 * i, mn, mx); idx[i] = (uint32_t)i; }
  if (out_codes) memcpy(out_codes, codes, (size_t)P * 4);
  /* stable LSD radix sort over all 32 key bits (cub SortPairs default range) */
  uint32_t *ci = codes, *co = tmpc, *ii = idx, *io = tmpi;
  for (int shift = 0; shift < 32; shift += 8) {
    size_t hist[257];
    memset(hist, 0, sizeof(hist));
    for (int i = 0; i < P; i++) hist[((ci[i] >> shift) & 255) + 1]++;
    for (int b = 0; b < 256; b++) hist[b + 1] += hist[b];
    for (int i = 0; i < P; i++) { size_t d = hist[(ci[i] >> shift) & 255]++; co[d] = ci[i]; io[d] = ii[i]; }
    uint32_t* t = ci; ci = co; co = t;
    t = ii; ii = io; io = t;
  }
  const uint32_t* sorted = ii; /* 4 passes: back in `idx` */
  if (out_indices) memcpy(out_indices, sorted, (size_t)P * 4);
  int num_boxes = (P + BOX_SIZE - 1) / BOX_SIZE;
  MinMax* boxes = (MinMax*)malloc((size_t)num_boxes * sizeof(MinMax));
  for (int b = 0; b < num_boxes; b++) {
    MinMax me = {{FLT_MAX, FLT_MAX, FLT_MAX}, {-FLT_MAX, -FLT_MAX, -FLT_MAX}};
    int hi = (b + 1) * BOX_SIZE < P ? (b + 1) * BOX_SIZE : P;
    for (int i = b * BOX_SIZE; i < hi; i++)
      for (int k = 0; k < 3; k++) {
        float v = points[3 * sorted[i] + k];
        me.mn[k] = fminf(me.mn[k], v);
        me.mx[k] = fmaxf(me.mx[k], v);
      }
    boxes[b] = me;
  }
#pragma omp parallel for schedule(dynamic, 256)
  for (int i0 = 0; i0 < P; i0++) {
    const float* point = points + 3 * sorted[i0];
    float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    int lo = i0 - 3 > 0 ? i0 - 3 : 0, hi = i0 + 3 < P - 1 ? i0 + 3 : P - 1;
    for (int i = lo; i <= hi; i++) {
      if (i == i0) continue;
      updateKBest3(point, points + 3 * sorted[i], best);
    }
    float reject = best[2];
    best[0] = best[1] = best[2] = FLT_MAX;
    for (int b = 0; b < num_boxes; b++) {
      float dist = distBoxPoint(&boxes[b], point);
      if (dist > reject || dist > best[2]) continue;
      int bhi = (b + 1) * BOX_SIZE < P ? (b + 1) * BOX_SIZE : P;
      for (int i = b * BOX_SIZE; i < bhi; i++) {
        if (i == i0) continue;
        updateKBest3(point, points + 3 * sorted[i], best);
      }
    }
    mean_dists[sorted[i0]] = (best[0] + best[1] + best[2]) / 3.0f;
  }
  free(boxes);
  free(codes); free(idx); free(tmpc); free(tmpi);
}
