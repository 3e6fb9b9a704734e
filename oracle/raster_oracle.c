/*
 * raster_oracle.c -- CPU restatement of the hair-gs differentiable Gaussian rasterizer.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (hair-gs_amd/) may import, link or
 * call this file; it is the checker the HIP kernels are compared against (tests/, smoke(),
 * and bench.py's cpu_baseline leg).
 *
 * PARITY STATUS: "parity unpinned" for the rasterizer arithmetic as a whole -- the reference
 * (CUDA + CUB + cooperative_groups) cannot be built in this image and ships no tests/golden
 * vectors for this path.  What IS pinned against the reference itself: SH colour evaluation
 * (utils/sh.py::eval_sh) and the camera matrices (utils/graphics.py), through
 * tests/golden/ref_python_pins.npz (generator: tests/golden/make_ref_python_pins.py).
 * glm's mat3 operator* / transpose operand order (what every bit-exact key depends on) is pinned
 * against the reference's vendored glm itself: oracle/build_ref.py compiles a small TU against
 * DGR/third_party/glm into oracle/_ref/, tests/golden/make_glm_pins.py records its outputs and
 * tests/test_oracle_pins.py asserts mat3_mul / mat3_tr / the computeCov3D and cov2D chains bit for bit.
 * Independent cross-checks: fp64 build of this file + central finite differences
 * (tests/test_oracle_checks.py).
 *
 * Each function cites the reference lines it follows.  Paths are relative to
 * /root/reference/submodules/diff-gaussian-rasterization/cuda_rasterizer/ (CR/).
 *
 * Arithmetic discipline: compiled with -ffp-contract=off; glm's column-major mat3 product is
 * restated with glm 0.9.9.9's operand order (third_party/glm/glm/detail/type_mat3x3.inl:486-520)
 * so that the fp32 build reproduces an un-contracted evaluation of the reference expressions.
 * Build with -DORACLE_F64 for a double-precision variant (finite-difference checks).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef ORACLE_F64
typedef double real;
#define R_SQRT sqrt
#define R_EXP exp
#define R_CEIL ceil
#define SYM(name) name##_f64
#else
typedef float real;
#define R_SQRT sqrtf
#define R_EXP expf
#define R_CEIL ceilf
#define SYM(name) name##_f32
#endif

#define BLOCK_X 16 /* CR/config.h:16 */
#define BLOCK_Y 16 /* CR/config.h:17 */
#define NCH 3      /* CR/config.h:15 */

/* CR/auxiliary.h:22-39 */
static const real SH_C0 = (real)0.28209479177387814f;
static const real SH_C1 = (real)0.4886025119029199f;
static const real SH_C2[5] = {(real)1.0925484305920792f, (real)-1.0925484305920792f, (real)0.31539156525252005f,
                              (real)-1.0925484305920792f, (real)0.5462742152960396f};
static const real SH_C3[7] = {(real)-0.5900435899266435f, (real)2.890611442640554f, (real)-0.4570457994644658f,
                              (real)0.3731763325901154f,  (real)-0.4570457994644658f, (real)1.445305721320277f,
                              (real)-0.5900435899266435f};

static real rmin(real a, real b) { return a < b ? a : b; }
static real rmax(real a, real b) { return a > b ? a : b; }

/* float -> int the way the GPU converts (saturating, NaN -> 0) */
static int f2i(real v) {
  if (v != v) return 0;
  if (v >= (real)2147483520.0) return 2147483647;
  if (v <= (real)-2147483648.0) return (int)-2147483647 - 1;
  return (int)v;
}

/* glm-style column-major 3x3: m[col][row] */
typedef struct { real m[3][3]; } mat3;

/* glm::mat3(x0,y0,z0, x1,y1,z1, x2,y2,z2): consecutive triples are COLUMNS */
static mat3 mat3_cols(real x0, real y0, real z0, real x1, real y1, real z1, real x2, real y2, real z2) {
  mat3 r;
  r.m[0][0] = x0; r.m[0][1] = y0; r.m[0][2] = z0;
  r.m[1][0] = x1; r.m[1][1] = y1; r.m[1][2] = z1;
  r.m[2][0] = x2; r.m[2][1] = y2; r.m[2][2] = z2;
  return r;
}
/* glm operator*(mat3, mat3): type_mat3x3.inl:486-520 */
static mat3 mat3_mul(mat3 A, mat3 B) {
  mat3 r;
  for (int c = 0; c < 3; c++)
    for (int w = 0; w < 3; w++)
      r.m[c][w] = A.m[0][w] * B.m[c][0] + A.m[1][w] * B.m[c][1] + A.m[2][w] * B.m[c][2];
  return r;
}
static mat3 mat3_tr(mat3 A) {
  mat3 r;
  for (int c = 0; c < 3; c++)
    for (int w = 0; w < 3; w++) r.m[c][w] = A.m[w][c];
  return r;
}

/* Exports for the glm pin (tests/test_oracle_pins.py): the two helpers above and the two chains the rasterizer builds
 * from them, on 9 floats in glm memory order (m[col][row]). */
void SYM(hgs_oracle_mat3_mul)(const real* a, const real* b, real* out) {
  mat3 A, B;
  memcpy(A.m, a, sizeof(A.m));
  memcpy(B.m, b, sizeof(B.m));
  mat3 r = mat3_mul(A, B);
  memcpy(out, r.m, sizeof(r.m));
}
void SYM(hgs_oracle_mat3_tr)(const real* a, real* out) {
  mat3 A;
  memcpy(A.m, a, sizeof(A.m));
  mat3 r = mat3_tr(A);
  memcpy(out, r.m, sizeof(r.m));
}
void SYM(hgs_oracle_mat3_gram)(const real* m, real* out) { /* Sigma = transpose(M) * M, CR/forward.cu:143 */
  mat3 M;
  memcpy(M.m, m, sizeof(M.m));
  mat3 r = mat3_mul(mat3_tr(M), M);
  memcpy(out, r.m, sizeof(r.m));
}
void SYM(hgs_oracle_mat3_sandwich)(const real* t, const real* v, real* out) { /* CR/forward.cu:108 */
  mat3 T, V;
  memcpy(T.m, t, sizeof(T.m));
  memcpy(V.m, v, sizeof(V.m));
  mat3 r = mat3_mul(mat3_mul(mat3_tr(T), mat3_tr(V)), T);
  memcpy(out, r.m, sizeof(r.m));
}

/* CR/auxiliary.h:58-66 */
static void transformPoint4x3(const real p[3], const real* M, real out[3]) {
  out[0] = M[0] * p[0] + M[4] * p[1] + M[8] * p[2] + M[12];
  out[1] = M[1] * p[0] + M[5] * p[1] + M[9] * p[2] + M[13];
  out[2] = M[2] * p[0] + M[6] * p[1] + M[10] * p[2] + M[14];
}
/* CR/auxiliary.h:68-77 */
static void transformPoint4x4(const real p[3], const real* M, real out[4]) {
  out[0] = M[0] * p[0] + M[4] * p[1] + M[8] * p[2] + M[12];
  out[1] = M[1] * p[0] + M[5] * p[1] + M[9] * p[2] + M[13];
  out[2] = M[2] * p[0] + M[6] * p[1] + M[10] * p[2] + M[14];
  out[3] = M[3] * p[0] + M[7] * p[1] + M[11] * p[2] + M[15];
}
/* CR/auxiliary.h:89-97 */
static void transformVec4x3Transpose(const real p[3], const real* M, real out[3]) {
  out[0] = M[0] * p[0] + M[1] * p[1] + M[2] * p[2];
  out[1] = M[4] * p[0] + M[5] * p[1] + M[6] * p[2];
  out[2] = M[8] * p[0] + M[9] * p[1] + M[10] * p[2];
}
/* CR/auxiliary.h:41-44 -- evaluated in double whatever `real` is */
static real ndc2Pix(real v, int S) { return (real)(((v + 1.0) * S - 1.0) * 0.5); }

/* CR/auxiliary.h:46-56 */
static void getRect(real px, real py, int max_radius, uint32_t rmin_[2], uint32_t rmax_[2], uint32_t gx, uint32_t gy) {
  int a;
  a = f2i((px - max_radius) / BLOCK_X); if (a < 0) a = 0; rmin_[0] = (uint32_t)a < gx ? (uint32_t)a : gx;
  a = f2i((py - max_radius) / BLOCK_Y); if (a < 0) a = 0; rmin_[1] = (uint32_t)a < gy ? (uint32_t)a : gy;
  a = f2i((px + max_radius + BLOCK_X - 1) / BLOCK_X); if (a < 0) a = 0; rmax_[0] = (uint32_t)a < gx ? (uint32_t)a : gx;
  a = f2i((py + max_radius + BLOCK_Y - 1) / BLOCK_Y); if (a < 0) a = 0; rmax_[1] = (uint32_t)a < gy ? (uint32_t)a : gy;
}

/* CR/forward.cu:20-71 */
static void computeColorFromSH(int idx, int deg, int max_coeffs, const real* means, const real* campos, const real* shs,
                               uint8_t* clamped, real out[3]) {
  real dir[3] = {means[3 * idx] - campos[0], means[3 * idx + 1] - campos[1], means[3 * idx + 2] - campos[2]};
  real len = R_SQRT(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]); /* glm::length = sqrt(dot) */
  dir[0] = dir[0] / len; dir[1] = dir[1] / len; dir[2] = dir[2] / len;
  const real* sh = shs + (size_t)idx * max_coeffs * 3;
#define SHC(k, ch) sh[3 * (k) + (ch)]
  real x = dir[0], y = dir[1], z = dir[2];
  for (int ch = 0; ch < 3; ch++) {
    real result = SH_C0 * SHC(0, ch);
    if (deg > 0) {
      result = result - SH_C1 * y * SHC(1, ch) + SH_C1 * z * SHC(2, ch) - SH_C1 * x * SHC(3, ch);
      if (deg > 1) {
        real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        result = result + SH_C2[0] * xy * SHC(4, ch) + SH_C2[1] * yz * SHC(5, ch) +
                 SH_C2[2] * ((real)2.0 * zz - xx - yy) * SHC(6, ch) + SH_C2[3] * xz * SHC(7, ch) +
                 SH_C2[4] * (xx - yy) * SHC(8, ch);
        if (deg > 2) {
          result = result + SH_C3[0] * y * ((real)3.0 * xx - yy) * SHC(9, ch) + SH_C3[1] * xy * z * SHC(10, ch) +
                   SH_C3[2] * y * ((real)4.0 * zz - xx - yy) * SHC(11, ch) +
                   SH_C3[3] * z * ((real)2.0 * zz - (real)3.0 * xx - (real)3.0 * yy) * SHC(12, ch) +
                   SH_C3[4] * x * ((real)4.0 * zz - xx - yy) * SHC(13, ch) + SH_C3[5] * z * (xx - yy) * SHC(14, ch) +
                   SH_C3[6] * x * (xx - (real)3.0 * yy) * SHC(15, ch);
        }
      }
    }
    result += (real)0.5;
    clamped[3 * idx + ch] = (result < 0);
    out[ch] = result > 0 ? result : 0;
  }
#undef SHC
}

/* CR/forward.cu:118-152 (no quaternion normalisation, :127) */
static void computeCov3D(const real scale[3], real mod, const real rot[4], real* cov3D) {
  mat3 S = mat3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
  S.m[0][0] = mod * scale[0];
  S.m[1][1] = mod * scale[1];
  S.m[2][2] = mod * scale[2];
  real r = rot[0], x = rot[1], y = rot[2], z = rot[3];
  mat3 R = mat3_cols((real)1 - (real)2 * (y * y + z * z), (real)2 * (x * y - r * z), (real)2 * (x * z + r * y),
                     (real)2 * (x * y + r * z), (real)1 - (real)2 * (x * x + z * z), (real)2 * (y * z - r * x),
                     (real)2 * (x * z - r * y), (real)2 * (y * z + r * x), (real)1 - (real)2 * (x * x + y * y));
  mat3 M = mat3_mul(S, R);
  mat3 Sigma = mat3_mul(mat3_tr(M), M);
  cov3D[0] = Sigma.m[0][0]; cov3D[1] = Sigma.m[0][1]; cov3D[2] = Sigma.m[0][2];
  cov3D[3] = Sigma.m[1][1]; cov3D[4] = Sigma.m[1][2]; cov3D[5] = Sigma.m[2][2];
}

/* shared by CR/forward.cu:74-113 and CR/backward_distwar.cu:167-200: builds t (clamped), J, W, T=W*J, Vrk, cov2D(+0.3) */
typedef struct { real t[3]; real txtz, tytz, limx, limy; mat3 J, W, T, Vrk, cov; } cov2d_ctx;
static void cov2d_common(const real mean[3], real fx, real fy, real tan_fovx, real tan_fovy, const real* cov3D,
                         const real* V, cov2d_ctx* c) {
  transformPoint4x3(mean, V, c->t);
  c->limx = (real)1.3 * tan_fovx;
  c->limy = (real)1.3 * tan_fovy;
  c->txtz = c->t[0] / c->t[2];
  c->tytz = c->t[1] / c->t[2];
  c->t[0] = rmin(c->limx, rmax(-c->limx, c->txtz)) * c->t[2];
  c->t[1] = rmin(c->limy, rmax(-c->limy, c->tytz)) * c->t[2];
  real tz = c->t[2];
  c->J = mat3_cols(fx / tz, 0, -(fx * c->t[0]) / (tz * tz), 0, fy / tz, -(fy * c->t[1]) / (tz * tz), 0, 0, 0);
  c->W = mat3_cols(V[0], V[4], V[8], V[1], V[5], V[9], V[2], V[6], V[10]);
  c->T = mat3_mul(c->W, c->J);
  c->Vrk = mat3_cols(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
  c->cov = mat3_mul(mat3_mul(mat3_tr(c->T), mat3_tr(c->Vrk)), c->T);
  c->cov.m[0][0] += (real)0.3;
  c->cov.m[1][1] += (real)0.3;
}

/* ------------------------------------------------------------------------------------------
 * Forward preprocess: CR/forward.cu:155-256 (+ in_frustum CR/auxiliary.h:139-164).
 * Per-Gaussian outputs are only written for Gaussians that survive (reference leaves the
 * slots of dropped Gaussians uninitialised; here they are left untouched, caller zero-fills).
 * Returns num_rendered = sum(tiles_touched); also fills point_offsets (inclusive scan,
 * CR/rasterizer_impl.cu:277).
 * ------------------------------------------------------------------------------------------ */
int SYM(hgs_oracle_preprocess)(int P, int D, int M, int W, int H, const real* means3D, const real* shs,
                               const real* colors_precomp, const real* opacities, const real* scales,
                               real scale_modifier, const real* rotations, const real* cov3D_precomp,
                               const real* viewmatrix, const real* projmatrix, const real* campos, real tan_fovx,
                               real tan_fovy, int* radii, real* means2D, real* depths, real* cov3Ds, real* conic_opacity,
                               real* rgb, uint8_t* clamped, uint32_t* tiles_touched, uint32_t* point_offsets) {
  const real focal_y = H / ((real)2.0 * tan_fovy); /* CR/rasterizer_impl.cu:222-223 */
  const real focal_x = W / ((real)2.0 * tan_fovx);
  const uint32_t gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
#pragma omp parallel for schedule(static)
  for (int idx = 0; idx < P; idx++) {
    radii[idx] = 0;
    tiles_touched[idx] = 0;
    real p_orig[3] = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
    real p_view[3];
    transformPoint4x3(p_orig, viewmatrix, p_view);
    if (p_view[2] <= (real)0.2) continue; /* CR/auxiliary.h:154 */
    real p_hom[4];
    transformPoint4x4(p_orig, projmatrix, p_hom);
    real p_w = (real)1.0 / (p_hom[3] + (real)0.0000001);
    real p_proj[3] = {p_hom[0] * p_w, p_hom[1] * p_w, p_hom[2] * p_w};
    const real* cov3D;
    if (cov3D_precomp) {
      cov3D = cov3D_precomp + 6 * (size_t)idx;
    } else {
      computeCov3D(scales + 3 * (size_t)idx, scale_modifier, rotations + 4 * (size_t)idx, cov3Ds + 6 * (size_t)idx);
      cov3D = cov3Ds + 6 * (size_t)idx;
    }
    cov2d_ctx c;
    cov2d_common(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, viewmatrix, &c);
    real cx = c.cov.m[0][0], cy = c.cov.m[0][1], cz = c.cov.m[1][1];
    real det = cx * cz - cy * cy;
    if (det == 0) continue;
    real det_inv = (real)1.0 / det;
    real conic[3] = {cz * det_inv, -cy * det_inv, cx * det_inv};
    real mid = (real)0.5 * (cx + cz);
    real lambda1 = mid + R_SQRT(rmax((real)0.1, mid * mid - det));
    real lambda2 = mid - R_SQRT(rmax((real)0.1, mid * mid - det));
    real my_radius = R_CEIL((real)3.0 * R_SQRT(rmax(lambda1, lambda2)));
    real pix[2] = {ndc2Pix(p_proj[0], W), ndc2Pix(p_proj[1], H)};
    uint32_t rmn[2], rmx[2];
    getRect(pix[0], pix[1], f2i(my_radius), rmn, rmx, gx, gy);
    if ((rmx[0] - rmn[0]) * (rmx[1] - rmn[1]) == 0) continue;
    if (!colors_precomp) {
      real col[3];
      computeColorFromSH(idx, D, M, means3D, campos, shs, clamped, col);
      rgb[3 * idx] = col[0]; rgb[3 * idx + 1] = col[1]; rgb[3 * idx + 2] = col[2];
    }
    depths[idx] = p_view[2];
    radii[idx] = f2i(my_radius);
    means2D[2 * idx] = pix[0];
    means2D[2 * idx + 1] = pix[1];
    conic_opacity[4 * idx] = conic[0]; conic_opacity[4 * idx + 1] = conic[1];
    conic_opacity[4 * idx + 2] = conic[2]; conic_opacity[4 * idx + 3] = opacities[idx];
    tiles_touched[idx] = (rmx[1] - rmn[1]) * (rmx[0] - rmn[0]);
  }
  uint32_t acc = 0;
  for (int i = 0; i < P; i++) { acc += tiles_touched[i]; point_offsets[i] = acc; }
  return (int)acc;
}

/* CR/rasterizer_impl.cu:35-50 */
uint32_t SYM(hgs_oracle_higher_msb)(uint32_t n) {
  uint32_t msb = sizeof(n) * 4, step = msb;
  while (step > 1) {
    step /= 2;
    if (n >> msb) msb += step; else msb -= step;
  }
  if (n >> msb) msb++;
  return msb;
}

/* stable LSD radix sort of (key,value) pairs on bits [0,end_bit) -- the semantics of
 * cub::DeviceRadixSort::SortPairs(..., 0, 32+bit) at CR/rasterizer_impl.cu:303-308 */
static void radix_sort_pairs(uint64_t* keys, uint32_t* vals, uint64_t* ktmp, uint32_t* vtmp, size_t n, int end_bit) {
  uint64_t *ki = keys, *ko = ktmp;
  uint32_t *vi = vals, *vo = vtmp;
  for (int shift = 0; shift < end_bit; shift += 8) {
    int bits = end_bit - shift < 8 ? end_bit - shift : 8;
    uint32_t mask = (1u << bits) - 1;
    size_t hist[257];
    memset(hist, 0, sizeof(hist));
    for (size_t i = 0; i < n; i++) hist[((ki[i] >> shift) & mask) + 1]++;
    for (int b = 0; b < 256; b++) hist[b + 1] += hist[b];
    for (size_t i = 0; i < n; i++) {
      size_t d = hist[(ki[i] >> shift) & mask]++;
      ko[d] = ki[i];
      vo[d] = vi[i];
    }
    uint64_t* tk = ki; ki = ko; ko = tk;
    uint32_t* tv = vi; vi = vo; vo = tv;
  }
  if (ki != keys) { memcpy(keys, ki, n * sizeof(uint64_t)); memcpy(vals, vi, n * sizeof(uint32_t)); }
}

/* ------------------------------------------------------------------------------------------
 * Binning: duplicateWithKeys (CR/rasterizer_impl.cu:70-111) + SortPairs (:300-308) +
 * memset/identifyTileRanges (:310-317, :116-138).
 * keys_sorted[R], point_list[R], ranges[2*T].  Depth bits always come from the fp32 depth.
 * ------------------------------------------------------------------------------------------ */
void SYM(hgs_oracle_bin)(int P, int W, int H, int R, const int* radii, const real* means2D, const real* depths,
                         const uint32_t* point_offsets, uint64_t* keys_sorted, uint32_t* point_list, uint32_t* ranges) {
  const uint32_t gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
  for (int idx = 0; idx < P; idx++) {
    if (radii[idx] > 0) {
      uint32_t off = (idx == 0) ? 0 : point_offsets[idx - 1];
      uint32_t rmn[2], rmx[2];
      getRect(means2D[2 * idx], means2D[2 * idx + 1], radii[idx], rmn, rmx, gx, gy);
      float df = (float)depths[idx];
      uint32_t dbits;
      memcpy(&dbits, &df, 4);
      for (uint32_t y = rmn[1]; y < rmx[1]; y++)
        for (uint32_t x = rmn[0]; x < rmx[0]; x++) {
          uint64_t key = (uint64_t)(y * gx + x);
          key <<= 32;
          key |= dbits;
          keys_sorted[off] = key;
          point_list[off] = (uint32_t)idx;
          off++;
        }
    }
  }
  if (R > 0) {
    uint64_t* kt = (uint64_t*)malloc((size_t)R * 8);
    uint32_t* vt = (uint32_t*)malloc((size_t)R * 4);
    int bit = (int)SYM(hgs_oracle_higher_msb)(gx * gy);
    radix_sort_pairs(keys_sorted, point_list, kt, vt, (size_t)R, 32 + bit);
    free(kt);
    free(vt);
  }
  memset(ranges, 0, (size_t)gx * gy * 2 * sizeof(uint32_t));
  for (int i = 0; i < R; i++) {
    uint32_t cur = (uint32_t)(keys_sorted[i] >> 32);
    if (i == 0) ranges[2 * cur] = 0;
    else {
      uint32_t prev = (uint32_t)(keys_sorted[i - 1] >> 32);
      if (cur != prev) { ranges[2 * prev + 1] = i; ranges[2 * cur] = i; }
    }
    if (i == R - 1) ranges[2 * cur + 1] = R;
  }
}

/* ------------------------------------------------------------------------------------------
 * Forward blend: renderCUDA, CR/forward.cu:261-374.  One pixel at a time; the block-level
 * staging/voting of the reference only affects scheduling, not values.
 * ------------------------------------------------------------------------------------------ */
void SYM(hgs_oracle_render)(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const real* means2D,
                            const real* features, const real* conic_opacity, const real* bg, real* final_T,
                            uint32_t* n_contrib, real* out_color) {
  const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
#pragma omp parallel for schedule(dynamic, 1)
  for (int tile = 0; tile < gx * gy; tile++) {
    int tx = tile % gx, ty = tile / gx;
    uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
    for (int ly = 0; ly < BLOCK_Y; ly++)
      for (int lx = 0; lx < BLOCK_X; lx++) {
        int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
        if (px >= W || py >= H) continue;
        size_t pix_id = (size_t)W * py + px;
        real pixf[2] = {(real)px, (real)py};
        real T = (real)1.0;
        uint32_t contributor = 0, last_contributor = 0;
        real C[NCH] = {0, 0, 0};
        for (uint32_t e = r0; e < r1; e++) {
          contributor++;
          uint32_t id = point_list[e];
          real dx = means2D[2 * id] - pixf[0], dy = means2D[2 * id + 1] - pixf[1];
          const real* co = conic_opacity + 4 * (size_t)id;
          real power = (real)-0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
          if (power > 0) continue;
          real alpha = rmin((real)0.99, co[3] * R_EXP(power));
          if (alpha < (real)1.0 / (real)255.0) continue;
          real test_T = T * (1 - alpha);
          if (test_T < (real)0.0001) break; /* done = true: nothing later touches this pixel */
          for (int ch = 0; ch < NCH; ch++) C[ch] += features[NCH * (size_t)id + ch] * alpha * T;
          T = test_T;
          last_contributor = contributor;
        }
        final_T[pix_id] = T;
        n_contrib[pix_id] = last_contributor;
        for (int ch = 0; ch < NCH; ch++) out_color[(size_t)ch * H * W + pix_id] = C[ch] + T * bg[ch];
      }
  }
}

/* ------------------------------------------------------------------------------------------
 * Backward blend: renderCUDABW_original, CR/backward_distwar.cu:855-1014 (the butterfly /
 * serialized variants :400-852 only change how the same terms are summed).
 * Accumulators are double: the oracle returns the exactly-ordered sum of the per-(pixel,entry)
 * terms; the GPU's fp32 summation order is free (reference: float atomics).
 * acc layout per Gaussian: [dmean2D.x, dmean2D.y, dconic.x, dconic.y, dconic.w, dopacity, dcolor0..2]
 * ------------------------------------------------------------------------------------------ */
void SYM(hgs_oracle_render_backward)(int P, int W, int H, const uint32_t* ranges, const uint32_t* point_list,
                                     const real* bg, const real* means2D, const real* conic_opacity, const real* colors,
                                     const real* final_Ts, const uint32_t* n_contrib, const real* dL_dpixels,
                                     double* acc /* [P][9], zeroed here */,
                                     uint8_t* fragile /* [P] or NULL: see below */,
                                     uint8_t* touched /* [P] or NULL: see below */) {
  const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
  memset(acc, 0, (size_t)P * 9 * sizeof(double));
  const real ddelx_dx = (real)(0.5 * W), ddely_dy = (real)(0.5 * H);
#pragma omp parallel for schedule(dynamic, 1)
  for (int tile = 0; tile < gx * gy; tile++) {
    int tx = tile % gx, ty = tile / gx;
    uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
    for (int ly = 0; ly < BLOCK_Y; ly++)
      for (int lx = 0; lx < BLOCK_X; lx++) {
        int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
        if (px >= W || py >= H) continue;
        size_t pix_id = (size_t)W * py + px;
        real pixf[2] = {(real)px, (real)py};
        const real T_final = final_Ts[pix_id];
        real T = T_final;
        uint32_t contributor = r1 - r0;
        const uint32_t last_contributor = n_contrib[pix_id];
        real accum_rec[NCH] = {0, 0, 0}, dL_dpixel[NCH], last_color[NCH] = {0, 0, 0};
        for (int ch = 0; ch < NCH; ch++) dL_dpixel[ch] = dL_dpixels[(size_t)ch * H * W + pix_id];
        real last_alpha = 0;
        int pixel_fragile = 0;
        for (uint32_t k = 0; k < r1 - r0; k++) {
          uint32_t id = point_list[r1 - k - 1];
          contributor--;
          if (contributor >= last_contributor) continue;
          real dx = means2D[2 * id] - pixf[0], dy = means2D[2 * id + 1] - pixf[1];
          const real* co = conic_opacity + 4 * (size_t)id;
          real power = (real)-0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
          if (power > 0) {
            if (fragile && (double)power < 1e-5) { fragile[id] = 1; pixel_fragile = 1; }
            continue;
          }
          real G = R_EXP(power);
          real alpha = rmin((real)0.99, co[3] * G);
          /* Checker aid (not part of the reference): a Gaussian is marked fragile when one of its (pixel, entry)
           * decisions sits within 1e-4 relative of the alpha >= 1/255 threshold or within 1e-5 of power > 0 -- an
           * implementation whose exp differs by an ulp may branch the other way there (the CUDA reference against
           * any CPU code has the same property).  Tests bound the number of such Gaussians and hold every other
           * one to the tolerance. */
          if (fragile && (fabs((double)alpha * 255.0 - 1.0) < 1e-4 || fabs((double)power) < 1e-5)) { fragile[id] = 1; pixel_fragile = 1; }
          if (alpha < (real)1.0 / (real)255.0) continue;
          T = T / ((real)1.0 - alpha);
          real dchannel_dcolor = alpha * T;
          real dL_dalpha = 0;
          double* a = acc + 9 * (size_t)id;
          for (int ch = 0; ch < NCH; ch++) {
            real c = colors[NCH * (size_t)id + ch];
            accum_rec[ch] = last_alpha * last_color[ch] + ((real)1.0 - last_alpha) * accum_rec[ch];
            last_color[ch] = c;
            real dL_dchannel = dL_dpixel[ch];
            dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
            double v = (double)(dchannel_dcolor * dL_dchannel);
#pragma omp atomic
            a[6 + ch] += v;
          }
          dL_dalpha *= T;
          last_alpha = alpha;
          real bg_dot_dpixel = 0;
          for (int ch = 0; ch < NCH; ch++) bg_dot_dpixel += bg[ch] * dL_dpixel[ch];
          dL_dalpha += (-T_final / ((real)1.0 - alpha)) * bg_dot_dpixel;
          real dL_dG = co[3] * dL_dalpha;
          real gdx = G * dx, gdy = G * dy;
          real dG_ddelx = -gdx * co[0] - gdy * co[1];
          real dG_ddely = -gdy * co[2] - gdx * co[1];
          double v0 = (double)(dL_dG * dG_ddelx * ddelx_dx), v1 = (double)(dL_dG * dG_ddely * ddely_dy);
          double v2 = (double)((real)-0.5 * gdx * dx * dL_dG), v3 = (double)((real)-0.5 * gdx * dy * dL_dG);
          double v4 = (double)((real)-0.5 * gdy * dy * dL_dG), v5 = (double)(G * dL_dalpha);
#pragma omp atomic
          a[0] += v0;
#pragma omp atomic
          a[1] += v1;
#pragma omp atomic
          a[2] += v2;
#pragma omp atomic
          a[3] += v3;
#pragma omp atomic
          a[4] += v4;
#pragma omp atomic
          a[5] += v5;
        }
        /* Checker aid, second part: a decision that an implementation takes the other way changes this PIXEL's transmittance
         * chain by that entry's alpha (~1/255) and the colour behind every nearer entry -- i.e. the terms of EVERY Gaussian the
         * pixel blends by up to ~0.4 % of what the pixel contributes to them, not only the fragile one's.  Those Gaussians are
         * enumerated too (`touched`): the pixel's walk is repeated and every blended entry marked. */
        if (touched && pixel_fragile) {
          uint32_t c2 = r1 - r0;
          for (uint32_t k = 0; k < r1 - r0; k++) {
            uint32_t id = point_list[r1 - k - 1];
            c2--;
            if (c2 >= last_contributor) continue;
            real dx = means2D[2 * id] - pixf[0], dy = means2D[2 * id + 1] - pixf[1];
            const real* co = conic_opacity + 4 * (size_t)id;
            real power = (real)-0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
            if (power > 0) continue;
            real alpha = rmin((real)0.99, co[3] * R_EXP(power));
            if (alpha < (real)1.0 / (real)255.0) continue;
            touched[id] = 1;
          }
        }
      }
  }
}

/* CR/auxiliary.h:107-117 */
static void dnormvdv3(const real v[3], const real dv[3], real out[3]) {
  real sum2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  real invsum32 = (real)1.0 / R_SQRT(sum2 * sum2 * sum2);
  out[0] = ((+sum2 - v[0] * v[0]) * dv[0] - v[1] * v[0] * dv[1] - v[2] * v[0] * dv[2]) * invsum32;
  out[1] = (-v[0] * v[1] * dv[0] + (sum2 - v[1] * v[1]) * dv[1] - v[2] * v[1] * dv[2]) * invsum32;
  out[2] = (-v[0] * v[2] * dv[0] - v[1] * v[2] * dv[1] + (sum2 - v[2] * v[2]) * dv[2]) * invsum32;
}

/* CR/backward_distwar.cu:21-140 */
static void computeColorFromSH_bw(int idx, int deg, int max_coeffs, const real* means, const real* campos,
                                  const real* shs, const uint8_t* clamped, const real* dL_dcolor, real* dL_dmeans,
                                  real* dL_dshs) {
  real dir_orig[3] = {means[3 * idx] - campos[0], means[3 * idx + 1] - campos[1], means[3 * idx + 2] - campos[2]};
  real len = R_SQRT(dir_orig[0] * dir_orig[0] + dir_orig[1] * dir_orig[1] + dir_orig[2] * dir_orig[2]);
  real x = dir_orig[0] / len, y = dir_orig[1] / len, z = dir_orig[2] / len;
  const real* sh = shs + (size_t)idx * max_coeffs * 3;
  real* dsh = dL_dshs + (size_t)idx * max_coeffs * 3;
  real dRGB[3];
  for (int ch = 0; ch < 3; ch++) dRGB[ch] = dL_dcolor[3 * idx + ch] * (clamped[3 * idx + ch] ? (real)0 : (real)1);
  real dRGBdx[3] = {0, 0, 0}, dRGBdy[3] = {0, 0, 0}, dRGBdz[3] = {0, 0, 0};
#define SHC(k, ch) sh[3 * (k) + (ch)]
#define DSH(k, coef) for (int ch = 0; ch < 3; ch++) dsh[3 * (k) + ch] = (coef) * dRGB[ch]
  DSH(0, SH_C0);
  if (deg > 0) {
    real d1 = -SH_C1 * y, d2 = SH_C1 * z, d3 = -SH_C1 * x;
    DSH(1, d1); DSH(2, d2); DSH(3, d3);
    for (int ch = 0; ch < 3; ch++) {
      dRGBdx[ch] = -SH_C1 * SHC(3, ch);
      dRGBdy[ch] = -SH_C1 * SHC(1, ch);
      dRGBdz[ch] = SH_C1 * SHC(2, ch);
    }
    if (deg > 1) {
      real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      real d4 = SH_C2[0] * xy, d5 = SH_C2[1] * yz, d6 = SH_C2[2] * ((real)2.0 * zz - xx - yy), d7 = SH_C2[3] * xz,
           d8 = SH_C2[4] * (xx - yy);
      DSH(4, d4); DSH(5, d5); DSH(6, d6); DSH(7, d7); DSH(8, d8);
      for (int ch = 0; ch < 3; ch++) {
        dRGBdx[ch] += SH_C2[0] * y * SHC(4, ch) + SH_C2[2] * (real)2.0 * -x * SHC(6, ch) + SH_C2[3] * z * SHC(7, ch) +
                      SH_C2[4] * (real)2.0 * x * SHC(8, ch);
        dRGBdy[ch] += SH_C2[0] * x * SHC(4, ch) + SH_C2[1] * z * SHC(5, ch) + SH_C2[2] * (real)2.0 * -y * SHC(6, ch) +
                      SH_C2[4] * (real)2.0 * -y * SHC(8, ch);
        dRGBdz[ch] += SH_C2[1] * y * SHC(5, ch) + SH_C2[2] * (real)2.0 * (real)2.0 * z * SHC(6, ch) +
                      SH_C2[3] * x * SHC(7, ch);
      }
      if (deg > 2) {
        real d9 = SH_C3[0] * y * ((real)3.0 * xx - yy), d10 = SH_C3[1] * xy * z,
             d11 = SH_C3[2] * y * ((real)4.0 * zz - xx - yy),
             d12 = SH_C3[3] * z * ((real)2.0 * zz - (real)3.0 * xx - (real)3.0 * yy),
             d13 = SH_C3[4] * x * ((real)4.0 * zz - xx - yy), d14 = SH_C3[5] * z * (xx - yy),
             d15 = SH_C3[6] * x * (xx - (real)3.0 * yy);
        DSH(9, d9); DSH(10, d10); DSH(11, d11); DSH(12, d12); DSH(13, d13); DSH(14, d14); DSH(15, d15);
        for (int ch = 0; ch < 3; ch++) {
          dRGBdx[ch] += (SH_C3[0] * SHC(9, ch) * (real)3.0 * (real)2.0 * xy + SH_C3[1] * SHC(10, ch) * yz +
                         SH_C3[2] * SHC(11, ch) * (real)-2.0 * xy + SH_C3[3] * SHC(12, ch) * (real)-3.0 * (real)2.0 * xz +
                         SH_C3[4] * SHC(13, ch) * ((real)-3.0 * xx + (real)4.0 * zz - yy) +
                         SH_C3[5] * SHC(14, ch) * (real)2.0 * xz + SH_C3[6] * SHC(15, ch) * (real)3.0 * (xx - yy));
          dRGBdy[ch] += (SH_C3[0] * SHC(9, ch) * (real)3.0 * (xx - yy) + SH_C3[1] * SHC(10, ch) * xz +
                         SH_C3[2] * SHC(11, ch) * ((real)-3.0 * yy + (real)4.0 * zz - xx) +
                         SH_C3[3] * SHC(12, ch) * (real)-3.0 * (real)2.0 * yz + SH_C3[4] * SHC(13, ch) * (real)-2.0 * xy +
                         SH_C3[5] * SHC(14, ch) * (real)-2.0 * yz + SH_C3[6] * SHC(15, ch) * (real)-3.0 * (real)2.0 * xy);
          dRGBdz[ch] += (SH_C3[1] * SHC(10, ch) * xy + SH_C3[2] * SHC(11, ch) * (real)4.0 * (real)2.0 * yz +
                         SH_C3[3] * SHC(12, ch) * (real)3.0 * ((real)2.0 * zz - xx - yy) +
                         SH_C3[4] * SHC(13, ch) * (real)4.0 * (real)2.0 * xz + SH_C3[5] * SHC(14, ch) * (xx - yy));
        }
      }
    }
  }
#undef SHC
#undef DSH
  real dL_ddir[3];
  dL_ddir[0] = dRGBdx[0] * dRGB[0] + dRGBdx[1] * dRGB[1] + dRGBdx[2] * dRGB[2];
  dL_ddir[1] = dRGBdy[0] * dRGB[0] + dRGBdy[1] * dRGB[1] + dRGBdy[2] * dRGB[2];
  dL_ddir[2] = dRGBdz[0] * dRGB[0] + dRGBdz[1] * dRGB[1] + dRGBdz[2] * dRGB[2];
  real dm[3];
  dnormvdv3(dir_orig, dL_ddir, dm);
  dL_dmeans[3 * idx] += dm[0];
  dL_dmeans[3 * idx + 1] += dm[1];
  dL_dmeans[3 * idx + 2] += dm[2];
}

/* CR/backward_distwar.cu:279-342 */
static void computeCov3D_bw(int idx, const real scale[3], real mod, const real rot[4], const real* dL_dcov3Ds,
                            real* dL_dscales, real* dL_drots) {
  real r = rot[0], x = rot[1], y = rot[2], z = rot[3];
  mat3 R = mat3_cols((real)1 - (real)2 * (y * y + z * z), (real)2 * (x * y - r * z), (real)2 * (x * z + r * y),
                     (real)2 * (x * y + r * z), (real)1 - (real)2 * (x * x + z * z), (real)2 * (y * z - r * x),
                     (real)2 * (x * z - r * y), (real)2 * (y * z + r * x), (real)1 - (real)2 * (x * x + y * y));
  mat3 S = mat3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
  real s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
  S.m[0][0] = s[0]; S.m[1][1] = s[1]; S.m[2][2] = s[2];
  mat3 M = mat3_mul(S, R);
  const real* d = dL_dcov3Ds + 6 * (size_t)idx;
  mat3 dSigma = mat3_cols(d[0], (real)0.5 * d[1], (real)0.5 * d[2], (real)0.5 * d[1], d[3], (real)0.5 * d[4],
                          (real)0.5 * d[2], (real)0.5 * d[4], d[5]);
  /* dL_dM = 2.0f * M * dL_dSigma  -> (2*M) * dSigma */
  mat3 M2;
  for (int c = 0; c < 3; c++) for (int w = 0; w < 3; w++) M2.m[c][w] = M.m[c][w] * (real)2.0;
  mat3 dM = mat3_mul(M2, dSigma);
  mat3 Rt = mat3_tr(R), dMt = mat3_tr(dM);
  for (int k = 0; k < 3; k++)
    dL_dscales[3 * idx + k] = Rt.m[k][0] * dMt.m[k][0] + Rt.m[k][1] * dMt.m[k][1] + Rt.m[k][2] * dMt.m[k][2];
  for (int k = 0; k < 3; k++) for (int w = 0; w < 3; w++) dMt.m[k][w] *= s[k];
#define D(a, b) dMt.m[a][b]
  real q0 = 2 * z * (D(0, 1) - D(1, 0)) + 2 * y * (D(2, 0) - D(0, 2)) + 2 * x * (D(1, 2) - D(2, 1));
  real q1 = 2 * y * (D(1, 0) + D(0, 1)) + 2 * z * (D(2, 0) + D(0, 2)) + 2 * r * (D(1, 2) - D(2, 1)) -
            4 * x * (D(2, 2) + D(1, 1));
  real q2 = 2 * x * (D(1, 0) + D(0, 1)) + 2 * r * (D(2, 0) - D(0, 2)) + 2 * z * (D(1, 2) + D(2, 1)) -
            4 * y * (D(2, 2) + D(0, 0));
  real q3 = 2 * r * (D(0, 1) - D(1, 0)) + 2 * x * (D(2, 0) + D(0, 2)) + 2 * y * (D(1, 2) + D(2, 1)) -
            4 * z * (D(1, 1) + D(0, 0));
#undef D
  dL_drots[4 * idx] = q0; dL_drots[4 * idx + 1] = q1; dL_drots[4 * idx + 2] = q2; dL_drots[4 * idx + 3] = q3;
}

/* ------------------------------------------------------------------------------------------
 * Backward preprocess = computeCov2DCUDA (CR/backward_distwar.cu:145-275) followed by
 * preprocessCUDA (:347-397).  All outputs must be zero-filled by the caller
 * (DGR/rasterize_points.cu:151-159); dL_dconic is [P][4] with .z unused (trap 9).
 * ------------------------------------------------------------------------------------------ */
void SYM(hgs_oracle_preprocess_backward)(int P, int D, int M, int W, int H, const real* means3D, const int* radii,
                                         const real* shs, const uint8_t* clamped, const real* scales,
                                         const real* rotations, real scale_modifier, const real* cov3Ds,
                                         const real* viewmatrix, const real* projmatrix, real tan_fovx, real tan_fovy,
                                         const real* campos, const real* dL_dmean2D /*[P][3]*/,
                                         const real* dL_dconic /*[P][4]*/, real* dL_dmeans, real* dL_dcolor,
                                         real* dL_dcov, real* dL_dsh, real* dL_dscale, real* dL_drot) {
  const real h_y = H / ((real)2.0 * tan_fovy), h_x = W / ((real)2.0 * tan_fovx);
#pragma omp parallel for schedule(static)
  for (int idx = 0; idx < P; idx++) {
    if (!(radii[idx] > 0)) continue;
    /* ---- computeCov2DCUDA ---- */
    real mean[3] = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
    real dcx = dL_dconic[4 * idx], dcy = dL_dconic[4 * idx + 1], dcz = dL_dconic[4 * idx + 3];
    cov2d_ctx c;
    cov2d_common(mean, h_x, h_y, tan_fovx, tan_fovy, cov3Ds + 6 * (size_t)idx, viewmatrix, &c);
    const real x_grad_mul = (c.txtz < -c.limx || c.txtz > c.limx) ? 0 : 1;
    const real y_grad_mul = (c.tytz < -c.limy || c.tytz > c.limy) ? 0 : 1;
    real a = c.cov.m[0][0], b = c.cov.m[0][1], cc = c.cov.m[1][1];
    real denom = a * cc - b * b;
    real dL_da = 0, dL_db = 0, dL_dc = 0;
    real denom2inv = (real)1.0 / ((denom * denom) + (real)0.0000001);
    real* dcv = dL_dcov + 6 * (size_t)idx;
#define T_(i, j) c.T.m[i][j]
#define V_(i, j) c.Vrk.m[i][j]
#define W_(i, j) c.W.m[i][j]
    if (denom2inv != 0) {
      dL_da = denom2inv * (-cc * cc * dcx + 2 * b * cc * dcy + (denom - a * cc) * dcz);
      dL_dc = denom2inv * (-a * a * dcz + 2 * a * b * dcy + (denom - a * cc) * dcx);
      dL_db = denom2inv * 2 * (b * cc * dcx - (denom + 2 * b * b) * dcy + a * b * dcz);
      dcv[0] = (T_(0, 0) * T_(0, 0) * dL_da + T_(0, 0) * T_(1, 0) * dL_db + T_(1, 0) * T_(1, 0) * dL_dc);
      dcv[3] = (T_(0, 1) * T_(0, 1) * dL_da + T_(0, 1) * T_(1, 1) * dL_db + T_(1, 1) * T_(1, 1) * dL_dc);
      dcv[5] = (T_(0, 2) * T_(0, 2) * dL_da + T_(0, 2) * T_(1, 2) * dL_db + T_(1, 2) * T_(1, 2) * dL_dc);
      dcv[1] = 2 * T_(0, 0) * T_(0, 1) * dL_da + (T_(0, 0) * T_(1, 1) + T_(0, 1) * T_(1, 0)) * dL_db +
               2 * T_(1, 0) * T_(1, 1) * dL_dc;
      dcv[2] = 2 * T_(0, 0) * T_(0, 2) * dL_da + (T_(0, 0) * T_(1, 2) + T_(0, 2) * T_(1, 0)) * dL_db +
               2 * T_(1, 0) * T_(1, 2) * dL_dc;
      dcv[4] = 2 * T_(0, 2) * T_(0, 1) * dL_da + (T_(0, 1) * T_(1, 2) + T_(0, 2) * T_(1, 1)) * dL_db +
               2 * T_(1, 1) * T_(1, 2) * dL_dc;
    } else {
      for (int i = 0; i < 6; i++) dcv[i] = 0;
    }
    real dL_dT00 = 2 * (T_(0, 0) * V_(0, 0) + T_(0, 1) * V_(0, 1) + T_(0, 2) * V_(0, 2)) * dL_da +
                   (T_(1, 0) * V_(0, 0) + T_(1, 1) * V_(0, 1) + T_(1, 2) * V_(0, 2)) * dL_db;
    real dL_dT01 = 2 * (T_(0, 0) * V_(1, 0) + T_(0, 1) * V_(1, 1) + T_(0, 2) * V_(1, 2)) * dL_da +
                   (T_(1, 0) * V_(1, 0) + T_(1, 1) * V_(1, 1) + T_(1, 2) * V_(1, 2)) * dL_db;
    real dL_dT02 = 2 * (T_(0, 0) * V_(2, 0) + T_(0, 1) * V_(2, 1) + T_(0, 2) * V_(2, 2)) * dL_da +
                   (T_(1, 0) * V_(2, 0) + T_(1, 1) * V_(2, 1) + T_(1, 2) * V_(2, 2)) * dL_db;
    real dL_dT10 = 2 * (T_(1, 0) * V_(0, 0) + T_(1, 1) * V_(0, 1) + T_(1, 2) * V_(0, 2)) * dL_dc +
                   (T_(0, 0) * V_(0, 0) + T_(0, 1) * V_(0, 1) + T_(0, 2) * V_(0, 2)) * dL_db;
    real dL_dT11 = 2 * (T_(1, 0) * V_(1, 0) + T_(1, 1) * V_(1, 1) + T_(1, 2) * V_(1, 2)) * dL_dc +
                   (T_(0, 0) * V_(1, 0) + T_(0, 1) * V_(1, 1) + T_(0, 2) * V_(1, 2)) * dL_db;
    real dL_dT12 = 2 * (T_(1, 0) * V_(2, 0) + T_(1, 1) * V_(2, 1) + T_(1, 2) * V_(2, 2)) * dL_dc +
                   (T_(0, 0) * V_(2, 0) + T_(0, 1) * V_(2, 1) + T_(0, 2) * V_(2, 2)) * dL_db;
    real dL_dJ00 = W_(0, 0) * dL_dT00 + W_(0, 1) * dL_dT01 + W_(0, 2) * dL_dT02;
    real dL_dJ02 = W_(2, 0) * dL_dT00 + W_(2, 1) * dL_dT01 + W_(2, 2) * dL_dT02;
    real dL_dJ11 = W_(1, 0) * dL_dT10 + W_(1, 1) * dL_dT11 + W_(1, 2) * dL_dT12;
    real dL_dJ12 = W_(2, 0) * dL_dT10 + W_(2, 1) * dL_dT11 + W_(2, 2) * dL_dT12;
#undef T_
#undef V_
#undef W_
    real tz = (real)1.0 / c.t[2], tz2 = tz * tz, tz3 = tz2 * tz;
    real dL_dt[3];
    dL_dt[0] = x_grad_mul * -h_x * tz2 * dL_dJ02;
    dL_dt[1] = y_grad_mul * -h_y * tz2 * dL_dJ12;
    dL_dt[2] = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * c.t[0]) * tz3 * dL_dJ02 +
               (2 * h_y * c.t[1]) * tz3 * dL_dJ12;
    real dm[3];
    transformVec4x3Transpose(dL_dt, viewmatrix, dm);
    dL_dmeans[3 * idx] = dm[0]; dL_dmeans[3 * idx + 1] = dm[1]; dL_dmeans[3 * idx + 2] = dm[2];

    /* ---- preprocessCUDA (backward) ---- */
    const real* proj = projmatrix;
    real m_hom[4];
    transformPoint4x4(mean, proj, m_hom);
    real m_w = (real)1.0 / (m_hom[3] + (real)0.0000001);
    real mul1 = (proj[0] * mean[0] + proj[4] * mean[1] + proj[8] * mean[2] + proj[12]) * m_w * m_w;
    real mul2 = (proj[1] * mean[0] + proj[5] * mean[1] + proj[9] * mean[2] + proj[13]) * m_w * m_w;
    real g0 = dL_dmean2D[3 * idx], g1 = dL_dmean2D[3 * idx + 1];
    dL_dmeans[3 * idx] += (proj[0] * m_w - proj[3] * mul1) * g0 + (proj[1] * m_w - proj[3] * mul2) * g1;
    dL_dmeans[3 * idx + 1] += (proj[4] * m_w - proj[7] * mul1) * g0 + (proj[5] * m_w - proj[7] * mul2) * g1;
    dL_dmeans[3 * idx + 2] += (proj[8] * m_w - proj[11] * mul1) * g0 + (proj[9] * m_w - proj[11] * mul2) * g1;
    if (shs) computeColorFromSH_bw(idx, D, M, means3D, campos, shs, clamped, dL_dcolor, dL_dmeans, dL_dsh);
    if (scales)
      computeCov3D_bw(idx, scales + 3 * (size_t)idx, scale_modifier, rotations + 4 * (size_t)idx, dL_dcov, dL_dscale,
                      dL_drot);
  }
}

/* CR/rasterizer_impl.cu:54-66 */
void SYM(hgs_oracle_mark_visible)(int P, const real* means3D, const real* viewmatrix, uint8_t* present) {
  for (int idx = 0; idx < P; idx++) {
    real p[3] = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]}, v[3];
    transformPoint4x3(p, viewmatrix, v);
    present[idx] = !(v[2] <= (real)0.2);
  }
}
