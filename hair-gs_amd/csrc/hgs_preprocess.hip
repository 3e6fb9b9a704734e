// hgs_preprocess.hip -- per-Gaussian kernels (one lane per Gaussian, 256-thread workgroups):
//   preprocess_fwd   cull, cov3D, EWA cov2D, conic, radius, tile rect, SH->RGB, per-tile instance counts
//                    replaces preprocessCUDA (cuda_rasterizer/forward.cu:155-256)
//   scatter          instance emission into per-tile segments (replaces duplicateWithKeys,
//                    cuda_rasterizer/rasterizer_impl.cu:70-111, and the P-wide InclusiveSum :277)
//   preprocess_bwd   per-instance gradient gather + computeCov2DCUDA + preprocessCUDA(bwd)
//                    (cuda_rasterizer/backward_distwar.cu:145-275, 347-397) in one pass
//   mark_visible     checkFrustum (cuda_rasterizer/rasterizer_impl.cu:54-66)
//
// Built with -ffp-contract=off: every fp32 expression that decides a radius, a tile rectangle or a
// depth key is evaluated operation by operation in the reference's order (glm column-major products,
// type_mat3x3.inl:486-520), so keys/rects are bit-identical to the CPU oracle.
#include "hgs_common.h"
#include "hgs_smooth.h"
#include "hgs_prologue.h"
#include "hgs_strand_fwd.h"
#include "hgs_strand_bwd.h"

namespace {

__device__ const float kSH_C0 = 0.28209479177387814f;
__device__ const float kSH_C1 = 0.4886025119029199f;
__device__ const float kSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                    -1.0925484305920792f, 0.5462742152960396f};
__device__ const float kSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                    0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                                    -0.5900435899266435f};

struct V3 { float x, y, z; };
struct M3 { float m[3][3]; };  // m[col][row], glm convention

__device__ __forceinline__ M3 m3mul(const M3& A, const M3& B) {
  M3 r;
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int w = 0; w < 3; w++) r.m[c][w] = A.m[0][w] * B.m[c][0] + A.m[1][w] * B.m[c][1] + A.m[2][w] * B.m[c][2];
  return r;
}
__device__ __forceinline__ M3 m3tr(const M3& A) {
  M3 r;
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int w = 0; w < 3; w++) r.m[c][w] = A.m[w][c];
  return r;
}
__device__ __forceinline__ V3 xform4x3(V3 p, const float* M) {
  return {M[0] * p.x + M[4] * p.y + M[8] * p.z + M[12], M[1] * p.x + M[5] * p.y + M[9] * p.z + M[13],
          M[2] * p.x + M[6] * p.y + M[10] * p.z + M[14]};
}
__device__ __forceinline__ float ndc2pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

// rotation matrix exactly as written at forward.cu:134-138 (columns), quaternion taken as given
__device__ __forceinline__ M3 quat_R(float r, float x, float y, float z) {
  M3 R;
  R.m[0][0] = 1.f - 2.f * (y * y + z * z); R.m[0][1] = 2.f * (x * y - r * z); R.m[0][2] = 2.f * (x * z + r * y);
  R.m[1][0] = 2.f * (x * y + r * z); R.m[1][1] = 1.f - 2.f * (x * x + z * z); R.m[1][2] = 2.f * (y * z - r * x);
  R.m[2][0] = 2.f * (x * z - r * y); R.m[2][1] = 2.f * (y * z + r * x); R.m[2][2] = 1.f - 2.f * (x * x + y * y);
  return R;
}

struct Cov2D {
  V3 t; float txtz, tytz, limx, limy;
  M3 W, T, Vrk;
  float a, b, c;  // cov2D (+0.3 low-pass on the diagonal)
};
// shared by forward (forward.cu:74-113) and backward (backward_distwar.cu:167-200)
__device__ __forceinline__ void cov2d(V3 mean, float fx, float fy, float tan_fovx, float tan_fovy, const float* cov3D,
                                      const float* V, Cov2D& o) {
  o.t = xform4x3(mean, V);
  o.limx = 1.3f * tan_fovx;
  o.limy = 1.3f * tan_fovy;
  o.txtz = o.t.x / o.t.z;
  o.tytz = o.t.y / o.t.z;
  o.t.x = fminf(o.limx, fmaxf(-o.limx, o.txtz)) * o.t.z;
  o.t.y = fminf(o.limy, fmaxf(-o.limy, o.tytz)) * o.t.z;
  const float tz = o.t.z;
  M3 J;
  J.m[0][0] = fx / tz; J.m[0][1] = 0.f; J.m[0][2] = -(fx * o.t.x) / (tz * tz);
  J.m[1][0] = 0.f; J.m[1][1] = fy / tz; J.m[1][2] = -(fy * o.t.y) / (tz * tz);
  J.m[2][0] = 0.f; J.m[2][1] = 0.f; J.m[2][2] = 0.f;
  o.W.m[0][0] = V[0]; o.W.m[0][1] = V[4]; o.W.m[0][2] = V[8];
  o.W.m[1][0] = V[1]; o.W.m[1][1] = V[5]; o.W.m[1][2] = V[9];
  o.W.m[2][0] = V[2]; o.W.m[2][1] = V[6]; o.W.m[2][2] = V[10];
  o.T = m3mul(o.W, J);
  o.Vrk.m[0][0] = cov3D[0]; o.Vrk.m[0][1] = cov3D[1]; o.Vrk.m[0][2] = cov3D[2];
  o.Vrk.m[1][0] = cov3D[1]; o.Vrk.m[1][1] = cov3D[3]; o.Vrk.m[1][2] = cov3D[4];
  o.Vrk.m[2][0] = cov3D[2]; o.Vrk.m[2][1] = cov3D[4]; o.Vrk.m[2][2] = cov3D[5];
  const M3 cov = m3mul(m3mul(m3tr(o.T), m3tr(o.Vrk)), o.T);
  o.a = cov.m[0][0] + 0.3f;
  o.b = cov.m[0][1];
  o.c = cov.m[1][1] + 0.3f;
}

__device__ __forceinline__ uint32_t block_sum_256(uint32_t v, uint32_t* lds4) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = v;
  __syncthreads();
  return lds4[0] + lds4[1] + lds4[2] + lds4[3];
}

// ---- block-private tile table ------------------------------------------------------------------------------------
// Consecutive Gaussians are consecutive strand segments: the 256 of a block touch only a few dozen distinct tiles, and
// a global atomic per (Gaussian, tile) serialises hundreds of read-modify-writes on the same L2 line (the two kernels
// below spent ~30 us each mostly there).  Instances are first counted in an LDS hash table keyed by tile (LDS atomics),
// then ONE global atomic per distinct tile and block publishes the count / reserves the slots.  Gaussians covering more
// than TH_MAX_AREA tiles, and inserts that find the table full, fall back to direct global atomics.
#define TH_LOG 10
#define TH_EMPTY 0xFFFFFFFFu
#define TH_MAX_AREA 16u
#define TH_PROBES 8
template <int LOG> struct TileHashT { static constexpr int SIZE = 1 << LOG; uint32_t key[SIZE]; uint32_t cnt[SIZE]; uint32_t base[SIZE]; };
using TileHash = TileHashT<TH_LOG>;
#define TH_SIZE (1 << TH_LOG)

template <int LOG> __device__ __forceinline__ void th_init(TileHashT<LOG>& h) {
  for (int i = threadIdx.x; i < (1 << LOG); i += HGS_BLOCK) { h.key[i] = TH_EMPTY; h.cnt[i] = 0u; }
}
template <int LOG> __device__ __forceinline__ int th_insert(TileHashT<LOG>& h, uint32_t t) {   // slot of tile t (inserting it), -1: table full
  uint32_t s = (t * 2654435761u) >> (32 - LOG);
#pragma unroll 1
  for (int k = 0; k < TH_PROBES; k++) {
    const uint32_t prev = atomicCAS(&h.key[s], TH_EMPTY, t);
    if (prev == TH_EMPTY || prev == t) return (int)s;
    s = (s + 1) & ((1 << LOG) - 1);
  }
  return -1;
}
template <int LOG> __device__ __forceinline__ int th_find(const TileHashT<LOG>& h, uint32_t t) {  // slot of an inserted tile, -1 if it never got in
  uint32_t s = (t * 2654435761u) >> (32 - LOG);
#pragma unroll 1
  for (int k = 0; k < TH_PROBES; k++) {
    const uint32_t cur = h.key[s];
    if (cur == t) return (int)s;
    if (cur == TH_EMPTY) return -1;
    s = (s + 1) & ((1 << LOG) - 1);
  }
  return -1;
}

// ------------------------------------------------------------------------------------------------
// The model's raw parameters, for the kernels that derive a lane's Gaussian themselves (hgs_hair_forward_preprocess /
// hgs_cloud_forward_preprocess): inputs, and the arrays the derived Gaussians are written to (the backward and the scatter
// kernel read them).
struct HgsParamSrc {
  const float* ep; const long long* pairs; const float* width; float f;        // strands
  const float* scaling_raw; const float* rotation_raw;                         // cloud (its means are a.means3D)
  const float* opacity_raw; const float* mask_raw;
  float* xyz; float* scale; float* quat; float* opacity; float* extra4;        // (xyz: strands only)
};
enum { SRC_GIVEN = 0, SRC_STRAND = 1, SRC_CLOUD = 2 };

// SRC != SRC_GIVEN: lane idx derives its Gaussian from the raw parameters (the device functions of strand_fwd_kernel /
// cloud_fwd_kernel: the same bits), stores it, and goes on with the values in registers; a.scales / rotations / opacities
// (and, for strands, a.means3D) are not read.
#ifndef HGS_PPF_DEAL
#define HGS_PPF_DEAL 1     // (0: A/B builds without the block-wide counting of large rectangles)
#endif
// development aid (tools/dev/ppf_trace.py; build with -DHGS_PPF_TRACE=1): per workgroup of the preprocess kernel the 10 ns ticks
// at which it entered, had its rectangles (and its lanes' own counting), had the block sum, had dealt its large rectangles, ended
#ifndef HGS_PPF_TRACE
#define HGS_PPF_TRACE 0
#endif
#ifndef HGS_PPF_DIRECT
#define HGS_PPF_DIRECT 2048u   // dealt instances of a block beyond which they are counted with global atomics, not through the LDS table
#endif
#if HGS_PPF_TRACE
#define PPF_TRACE_MAX 8192
__device__ unsigned long long g_ppf_trace[PPF_TRACE_MAX][8];
#define PPF_MARK(k) do { if (threadIdx.x == 0 && blockIdx.x < PPF_TRACE_MAX) g_ppf_trace[blockIdx.x][k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PPF_MARK(k) do { } while (0)
#endif
template <int SRC>
__device__ __forceinline__ void preprocess_fwd_body(const HgsFwdArgs& a, const HgsGeom& g, const HgsImage& im, int* radii,
                                                    const HgsParamSrc& st, uint32_t* red, TileHash& th) {
  constexpr bool STRAND = SRC == SRC_STRAND, DERIVED = SRC != SRC_GIVEN;
  // Round 4: every load that depends on nothing is issued HERE, in front of the dependent chain.  The kernel is a chain of
  // memory round trips at ~4 wavefronts per CU (vector pipe 0.16): the ISA of round 3 fetched the view matrix, then the
  // projection matrix, then opacity / mask, then the SH coefficients each behind a store of the lane's derived Gaussian --
  // a store the compiler must assume may alias them -- i.e. one more exposed round trip apiece.  The matrices are
  // wave-uniform and unwritten during the launch (the hair / cloud kernels read the view TABLE's row, not the slot a rider
  // is filling): loaded at entry they become scalar loads.
  float Vm[16], Pmat[16], cam[3];
#pragma unroll
  for (int k = 0; k < 16; k++) { Vm[k] = a.viewmatrix[k]; Pmat[k] = a.projmatrix[k]; }
#pragma unroll
  for (int k = 0; k < 3; k++) cam[k] = a.campos[k];
  PPF_MARK(0);
  const int idx = blockIdx.x * HGS_BLOCK + threadIdx.x;
  const bool live = idx < a.P;
  const size_t li = live ? (size_t)idx : 0;                   // (loads of the lanes past P read Gaussian 0 and are dropped)
  long long i0 = 0, i1 = 0;
  float w_raw = 0.f, o_raw = 0.f, m_raw = 0.f, s_raw[3] = {0.f, 0.f, 0.f}, op_in = 0.f, sc_in[3] = {0.f, 0.f, 0.f}, sh_dc[3] = {0.f, 0.f, 0.f};
  float4 r_raw = make_float4(1.f, 0.f, 0.f, 0.f), q_in = make_float4(1.f, 0.f, 0.f, 0.f);
  V3 p_in = {0.f, 0.f, 0.f};
  if (SRC == SRC_STRAND) {
    i0 = st.pairs[2 * li]; i1 = st.pairs[2 * li + 1];
    w_raw = st.width[li]; o_raw = st.opacity_raw[li]; m_raw = st.mask_raw[li];
  } else if (SRC == SRC_CLOUD) {
    s_raw[0] = st.scaling_raw[3 * li]; s_raw[1] = st.scaling_raw[3 * li + 1]; s_raw[2] = st.scaling_raw[3 * li + 2];
    r_raw = ((const float4*)st.rotation_raw)[li];
    o_raw = st.opacity_raw[li]; m_raw = st.mask_raw[li];
  } else {
    op_in = a.opacities[li];
    if (!a.cov3D_precomp) {
      sc_in[0] = a.scales[3 * li]; sc_in[1] = a.scales[3 * li + 1]; sc_in[2] = a.scales[3 * li + 2];
      q_in = ((const float4*)a.rotations)[li];
    }
  }
  if (!STRAND) p_in = V3{a.means3D[3 * li], a.means3D[3 * li + 1], a.means3D[3 * li + 2]};
  if (!a.colors_precomp) {
    const float* sh = a.shs + li * a.M * 3;
    sh_dc[0] = sh[0]; sh_dc[1] = sh[1]; sh_dc[2] = sh[2];
  }
  th_init(th);
  __syncthreads();
  const int gx = (a.W + HGS_TILE - 1) / HGS_TILE, gy = (a.H + HGS_TILE - 1) / HGS_TILE;
  uint32_t ntiles = 0;
  uint32_t big_area = 0, big_x0y0 = 0, big_w = 0;   // a rectangle of more than TH_MAX_AREA tiles: counted by the whole block
  if (live) {
    int my_radius_i = 0;
    HgsRect rc = {0, 0, 0, 0, 0, 0};
    HgsStrandGaussian sgn = {};
    float opacity_v = 0.f, sc0 = 0.f, sc1 = 0.f, sc2 = 0.f;
    float4 qd = make_float4(1.f, 0.f, 0.f, 0.f);
    if (SRC == SRC_STRAND) {
      sgn = hgs_strand_gaussian(st.ep[3 * i0], st.ep[3 * i0 + 1], st.ep[3 * i0 + 2], st.ep[3 * i1], st.ep[3 * i1 + 1],
                                st.ep[3 * i1 + 2], w_raw, st.f);
      opacity_v = hgs_sigmoid(o_raw);
      sc0 = sgn.s0; sc1 = sgn.sw; sc2 = sgn.sw;
      qd = make_float4(sgn.q0, sgn.q1, sgn.q2, sgn.q3);
      st.xyz[3 * (size_t)idx] = sgn.mx; st.xyz[3 * (size_t)idx + 1] = sgn.my; st.xyz[3 * (size_t)idx + 2] = sgn.mz;
      ((float4*)st.extra4)[idx] = make_float4(hgs_sigmoid(m_raw), sgn.ux, sgn.uy, sgn.uz);
    } else if (SRC == SRC_CLOUD) {
      const HgsCloudGaussian c = hgs_cloud_gaussian(s_raw[0], s_raw[1], s_raw[2], r_raw, o_raw, m_raw);
      opacity_v = c.opacity; sc0 = c.s0; sc1 = c.s1; sc2 = c.s2; qd = c.q;
      ((float4*)st.extra4)[idx] = c.extra;
    }
    if (DERIVED) {
      st.scale[3 * (size_t)idx] = sc0; st.scale[3 * (size_t)idx + 1] = sc1; st.scale[3 * (size_t)idx + 2] = sc2;
      ((float4*)st.quat)[idx] = qd;
      st.opacity[idx] = opacity_v;
    }
    do {
      const V3 p = STRAND ? V3{sgn.mx, sgn.my, sgn.mz} : p_in;
      const V3 pv = xform4x3(p, Vm);
      if (pv.z <= 0.2f) {  // auxiliary.h:154 (the `prefiltered` trap of :156-160 is not reproduced: it aborts the GPU)
        break;
      }
      const float* Pm = Pmat;
      const float hx = Pm[0] * p.x + Pm[4] * p.y + Pm[8] * p.z + Pm[12];
      const float hy = Pm[1] * p.x + Pm[5] * p.y + Pm[9] * p.z + Pm[13];
      const float hw = Pm[3] * p.x + Pm[7] * p.y + Pm[11] * p.z + Pm[15];
      const float p_w = 1.0f / (hw + 0.0000001f);
      const float projx = hx * p_w, projy = hy * p_w;
      float cov3[6];
      if (a.cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 6; k++) cov3[k] = a.cov3D_precomp[6 * (size_t)idx + k];
      } else {
        const float mod = a.scale_modifier;
        const float s0 = mod * (DERIVED ? sc0 : sc_in[0]), s1 = mod * (DERIVED ? sc1 : sc_in[1]),
                    s2 = mod * (DERIVED ? sc2 : sc_in[2]);
        const float4 q = DERIVED ? qd : q_in;
        const M3 R = quat_R(q.x, q.y, q.z, q.w);
        M3 Mm;  // M = S * R  (S diagonal: M[c][r] = s_r * R[c][r]; the zero terms of the full product add exact zeros)
#pragma unroll
        for (int c = 0; c < 3; c++) { Mm.m[c][0] = s0 * R.m[c][0]; Mm.m[c][1] = s1 * R.m[c][1]; Mm.m[c][2] = s2 * R.m[c][2]; }
        const M3 Sg = m3mul(m3tr(Mm), Mm);
        cov3[0] = Sg.m[0][0]; cov3[1] = Sg.m[0][1]; cov3[2] = Sg.m[0][2];
        cov3[3] = Sg.m[1][1]; cov3[4] = Sg.m[1][2]; cov3[5] = Sg.m[2][2];
#pragma unroll
        for (int k = 0; k < 6; k++) g.cov3D[6 * (size_t)idx + k] = cov3[k];
      }
      const float focal_y = a.H / (2.0f * a.tan_fovy), focal_x = a.W / (2.0f * a.tan_fovx);
      Cov2D c2;
      cov2d(p, focal_x, focal_y, a.tan_fovx, a.tan_fovy, cov3, Vm, c2);
      const float det = c2.a * c2.c - c2.b * c2.b;
      if (det == 0.0f) break;
      const float det_inv = 1.f / det;
      const float4 conic_o = {c2.c * det_inv, -c2.b * det_inv, c2.a * det_inv, DERIVED ? opacity_v : op_in};
      const float mid = 0.5f * (c2.a + c2.c);
      const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
      const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
      const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
      const float pixx = ndc2pix(projx, a.W), pixy = ndc2pix(projy, a.H);
      const int ri = hgs_f2i(my_radius);
      int x0 = hgs_f2i((pixx - ri) / HGS_TILE), y0 = hgs_f2i((pixy - ri) / HGS_TILE);
      int x1 = hgs_f2i((pixx + ri + HGS_TILE - 1) / HGS_TILE), y1 = hgs_f2i((pixy + ri + HGS_TILE - 1) / HGS_TILE);
      x0 = min(gx, max(0, x0)); y0 = min(gy, max(0, y0));
      x1 = min(gx, max(0, x1)); y1 = min(gy, max(0, y1));
      const uint32_t area = (uint32_t)(x1 - x0) * (uint32_t)(y1 - y0);
      if (area == 0) break;
      if (!a.colors_precomp) {
        // computeColorFromSH, forward.cu:20-71
        V3 d = {p.x - cam[0], p.y - cam[1], p.z - cam[2]};
        const float len = sqrtf(d.x * d.x + d.y * d.y + d.z * d.z);
        const float x = d.x / len, y = d.y / len, z = d.z / len;
        const float* sh = a.shs + (size_t)idx * a.M * 3;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
          float res = kSH_C0 * sh_dc[ch];
          if (a.D > 0) {
            res = res - kSH_C1 * y * sh[3 + ch] + kSH_C1 * z * sh[6 + ch] - kSH_C1 * x * sh[9 + ch];
            if (a.D > 1) {
              const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
              res = res + kSH_C2[0] * xy * sh[12 + ch] + kSH_C2[1] * yz * sh[15 + ch] +
                    kSH_C2[2] * (2.0f * zz - xx - yy) * sh[18 + ch] + kSH_C2[3] * xz * sh[21 + ch] +
                    kSH_C2[4] * (xx - yy) * sh[24 + ch];
              if (a.D > 2) {
                res = res + kSH_C3[0] * y * (3.0f * xx - yy) * sh[27 + ch] + kSH_C3[1] * xy * z * sh[30 + ch] +
                      kSH_C3[2] * y * (4.0f * zz - xx - yy) * sh[33 + ch] +
                      kSH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[36 + ch] +
                      kSH_C3[4] * x * (4.0f * zz - xx - yy) * sh[39 + ch] + kSH_C3[5] * z * (xx - yy) * sh[42 + ch] +
                      kSH_C3[6] * x * (xx - 3.0f * yy) * sh[45 + ch];
              }
            }
          }
          res += 0.5f;
          g.clamped[3 * (size_t)idx + ch] = (res < 0.f);
          g.rgb[3 * (size_t)idx + ch] = fmaxf(res, 0.0f);
        }
      }
      g.depths[idx] = pv.z;
      g.means2D[idx] = make_float2(pixx, pixy);
      g.conic_opacity[idx] = conic_o;
      my_radius_i = ri;
      if (a.tile_cull) {
        // The reference's square of 3-sigma radius (forward.cu:229-235) is kept for `radii`, but a pixel blends this
        // Gaussian only where opacity * exp(power) >= 1/255 (forward.cu:358), i.e. inside the ellipse 0.5 d^T Q d <= tau,
        // tau = ln(255 opacity), whose bounding box has half extents sqrt(2 tau cov_xx), sqrt(2 tau cov_yy) (cov = Q^-1).
        // Tiles outside that box (inflated like the quadrant masks of the sort kernel: 0.05% + 0.01 px) hold no pixel
        // that passes the test: they get no instance.  For thin strand Gaussians that is a quarter of all instances;
        // images and gradients are bit-identical with and without (tests/test_gpu_raster.py).
        const float tau = logf(255.f * conic_o.w);
        const float cdet = conic_o.x * conic_o.z - conic_o.y * conic_o.y;
        if (!(tau > 0.f)) {
          x1 = x0; y1 = y0;                 // alpha < 1/255 everywhere
        } else if (cdet > 0.f) {
          const float ex = sqrtf(2.f * tau * (conic_o.z / cdet)) * 1.0005f + 0.01f;
          const float ey = sqrtf(2.f * tau * (conic_o.x / cdet)) * 1.0005f + 0.01f;
          if (ex < 1e8f && ey < 1e8f) {     // (false for NaN / inf: no shrinking)
            // tile t holds pixels 16 t .. 16 t + 15 (centres at integers): overlap <=> c + e >= 16 t and c - e <= 16 t + 15
            const float lox = ceilf((pixx - ex - (float)(HGS_TILE - 1)) * (1.f / HGS_TILE)), hix = floorf((pixx + ex) * (1.f / HGS_TILE));
            const float loy = ceilf((pixy - ey - (float)(HGS_TILE - 1)) * (1.f / HGS_TILE)), hiy = floorf((pixy + ey) * (1.f / HGS_TILE));
            x0 = max(x0, (int)fmaxf(lox, 0.f)); x1 = min(x1, (int)fminf(hix, (float)gx) + 1);
            y0 = max(y0, (int)fmaxf(loy, 0.f)); y1 = min(y1, (int)fminf(hiy, (float)gy) + 1);
            if (x1 < x0) x1 = x0;
            if (y1 < y0) y1 = y0;
          }
        }
      }
      const uint32_t area_kept = (uint32_t)(x1 - x0) * (uint32_t)(y1 - y0);
      ntiles = area_kept;
      rc.x0 = (uint16_t)x0; rc.y0 = (uint16_t)y0; rc.x1 = (uint16_t)x1; rc.y1 = (uint16_t)y1;
      // per-tile instance counts: integer atomics, order-independent (block-private table first, see TileHash).  A lane walks
      // its own rectangle only while that is short; larger ones are dealt to the whole block below.
      if (!HGS_PPF_DEAL || area_kept <= TH_MAX_AREA) {
        for (int ty = y0; ty < y1; ty++)
          for (int tx = x0; tx < x1; tx++) {
            const uint32_t t = (uint32_t)(ty * gx + tx);
            const int sl = area_kept <= TH_MAX_AREA ? th_insert(th, t) : -1;
            if (sl >= 0) atomicAdd(&th.cnt[sl], 1u);
            else atomicAdd(&im.tile_count[HGS_TILE_SLOT(t, im.tile_mask)], 1u);
          }
      } else {
        big_area = a.row_runs ? (uint32_t)(y1 - y0) : area_kept;   // units dealt below: tile rows / tiles
        big_x0y0 = (uint32_t)x0 | ((uint32_t)y0 << 16);
        big_w = (uint32_t)(x1 - x0);
      }
    } while (0);
    radii[idx] = my_radius_i;
    g.tiles_touched[idx] = ntiles;
    g.rect[idx] = rc;
  }
  // ---- (round 6) rectangles of more than TH_MAX_AREA tiles: their instances are counted by ALL threads of the block, evenly.
  // Rounds 1-5 let the lane walk its own rectangle with one global atomic per tile: a workgroup took as long as its largest
  // Gaussian, and a Stage-I cloud at 1080p (12 tiles per Gaussian on average, hundreds for some: 2.3 M instances of 195 k
  // Gaussians) sent every one of those atomics to the same few thousand counters -- 298 us for a kernel that takes 12 at
  // north_star.  Now (only in blocks that hold such a rectangle: one barrier elsewhere) every lane leaves (origin, width, area) in
  // LDS, a block scan gives the prefix, thread t takes the instances [t, t + 1) x ceil(total / 256) of the concatenated
  // rectangles -- one binary search, then a walk -- through the same LDS tile table: one global atomic per distinct tile and
  // block.  Counts are integers: the same numbers whoever adds them.
  // (the block-wide "any" rides on the barrier of the block sum: blocks without such a rectangle -- every block of a fresh strand
  // model -- pay nothing)
  __shared__ uint32_t d_any[4];
#if HGS_PPF_TRACE
  __syncthreads();
  PPF_MARK(1);
#endif
  {
    const unsigned long long anyb = __ballot(big_area != 0u);
    if ((threadIdx.x & 63) == 0) d_any[threadIdx.x >> 6] = anyb != 0ull ? 1u : 0u;
  }
  const uint32_t bs = block_sum_256(ntiles, red);   // (its barriers also order the table updates above)
  if (threadIdx.x == 0) g.block_sums[blockIdx.x] = bs;
  PPF_MARK(2);
  if (HGS_PPF_DEAL && (d_any[0] | d_any[1] | d_any[2] | d_any[3]) != 0u) {
    __shared__ uint32_t d_org[HGS_BLOCK], d_w[HGS_BLOCK], d_off[HGS_BLOCK + 1], d_ws[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = hgs_wave_incl_scan(big_area, lane);
    if (lane == 63) d_ws[wave] = inc;
    d_org[threadIdx.x] = big_x0y0;
    d_w[threadIdx.x] = big_w;
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (int w = 0; w < 4; w++) { if (w < wave) base += d_ws[w]; total += d_ws[w]; }
    d_off[threadIdx.x] = base + inc - big_area;
    if (threadIdx.x == HGS_BLOCK - 1) d_off[HGS_BLOCK] = total;
    __syncthreads();
    const uint32_t per = (total + HGS_BLOCK - 1) / HGS_BLOCK;
    const uint32_t k0 = min(total, threadIdx.x * per), k1 = min(total, k0 + per);
    if (k0 < k1) {
      int lo = 0, hi = HGS_BLOCK;                       // largest j with d_off[j] <= k0
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (d_off[mid] <= k0) lo = mid; else hi = mid; }
      int j = lo;
      uint32_t l = k0 - d_off[j], nj = d_off[j + 1] - d_off[j];
      uint32_t org = d_org[j], w = d_w[j];
      if (a.row_runs) {
        // (HGS_COUNT_ROW_RUNS) the unit is a tile ROW of a rectangle: +1 where its run of tiles starts, -1 behind its end, in the
        // row-major array im.tile_delta -- a run that ends at the frame's right edge closes on the next row's first entry, which is
        // where the running sum of tile_delta_kernel (ONE sum over all tiles, not one per row) has to drop.  Two atomics for a
        // row of any width against one per tile: a Stage-I cloud's large Gaussians are 10-30 tiles wide (tools/dev/ppf_trace.py:
        // the heaviest block of such a frame deals 14 000 tiles, ~1400 rows).  Rows of one or two tiles go to the counters directly.
        for (uint32_t k = k0; k < k1; k++) {
          while (l >= nj) { j++; l = 0; nj = d_off[j + 1] - d_off[j]; if (nj) { org = d_org[j]; w = d_w[j]; } }
          const uint32_t t = ((org >> 16) + l) * (uint32_t)gx + (org & 0xFFFFu);
          if (w <= 2u) {
            atomicAdd(&im.tile_count[HGS_TILE_SLOT(t, im.tile_mask)], 1u);
            if (w == 2u) atomicAdd(&im.tile_count[HGS_TILE_SLOT(t + 1u, im.tile_mask)], 1u);
          } else {
            atomicAdd(&im.tile_delta[t], 1);
            atomicAdd(&im.tile_delta[t + w], -1);
          }
          l++;
        }
      } else {
        int tx = (int)(org & 0xFFFFu) + (int)(l % max(w, 1u)), ty = (int)(org >> 16) + (int)(l / max(w, 1u));
        for (uint32_t k = k0; k < k1; k++) {
          while (l >= nj) {                               // next Gaussian with a dealt rectangle
            j++; l = 0; nj = d_off[j + 1] - d_off[j];
            if (nj) { org = d_org[j]; w = d_w[j]; tx = (int)(org & 0xFFFFu); ty = (int)(org >> 16); }
          }
          const uint32_t t = (uint32_t)(ty * gx + tx);
          // (a block with more dealt instances than the table has room for distinct tiles goes to the counters directly: its
          // instances are runs of consecutive tiles, nearly all distinct, and a full table costs every insert its eight probes --
          // tools/dev/ppf_trace.py: 55 us for the 14 000 instances of a Stage-I frame's heaviest block, ~1 us per instance and thread)
          const int sl = total > HGS_PPF_DIRECT ? -1 : th_insert(th, t);
          if (sl >= 0) atomicAdd(&th.cnt[sl], 1u);
          else atomicAdd(&im.tile_count[HGS_TILE_SLOT(t, im.tile_mask)], 1u);
          l++;
          if (++tx == (int)((org & 0xFFFFu) + w)) { tx = (int)(org & 0xFFFFu); ty++; }
        }
      }
    }
    __syncthreads();                                    // (the table is flushed below)
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && a.fused_scan_ptr) {   // (status words were cleared by the launch before this one)
    im.status[HGS_ST_SCANPTR_LO] = (uint32_t)a.fused_scan_ptr | (a.row_runs ? 1u : 0u);   // (a 4-byte aligned pointer: bit 0 tells the scan workgroups of the scatter kernel about the row-run marks)
    im.status[HGS_ST_SCANPTR_HI] = (uint32_t)(a.fused_scan_ptr >> 32);
  }
  PPF_MARK(3);
  for (int i = threadIdx.x; i < TH_SIZE; i += HGS_BLOCK)
    if (th.key[i] != TH_EMPTY) atomicAdd(&im.tile_count[HGS_TILE_SLOT(th.key[i], im.tile_mask)], th.cnt[i]);
#if HGS_PPF_TRACE
  __syncthreads();
  PPF_MARK(4);
  if (threadIdx.x == 0 && blockIdx.x < PPF_TRACE_MAX) g_ppf_trace[blockIdx.x][5] = bs;
#endif
}

__global__ __launch_bounds__(HGS_BLOCK) void preprocess_fwd_kernel(HgsFwdArgs a, HgsGeom g, HgsImage im, int* radii) {
  __shared__ uint32_t red[4];
  __shared__ TileHash th;
  preprocess_fwd_body<SRC_GIVEN>(a, g, im, radii, HgsParamSrc{}, red, th);
}

// The iteration's FIRST launch for a strand model (hgs_hair_forward_preprocess): strand parameters -> Gaussians -> preprocess in
// one kernel, with the riders the parameter kernel used to carry (include/hgs.h HgsStrandFusion: smoothness partial sums;
// HgsPrologue: view select + clearing of the image buffer's counters -- `pro` stays the LAST argument, where the graph functions
// find it).  Two things differ from the two-launch form because the riders now run BESIDE the preprocess workgroups:
//   * the view matrices are read from the view TABLE's row (pro.table[pro.view]), not from the slot the rider fills;
//   * the rider's zero range must not hold the tile counters these workgroups increment (the caller passes the range behind
//     them: they are left at zero by the scatter kernel's scan workgroups, which clear what they have read), and the two status
//     words workgroup 0 writes are stepped over.
__global__ __launch_bounds__(HGS_BLOCK) void hair_preprocess_fwd_kernel(HgsFwdArgs a, HgsGeom g, HgsImage im, int* radii,
                                                                        HgsParamSrc st, HgsStrandFusion fu, HgsPrologue pro) {
  __shared__ uint32_t red[4];
  __shared__ TileHash th;
  const unsigned npro = pro.table ? hgs_prologue_blocks(pro.zero_bytes / 4) : 0u;
  if (blockIdx.x >= gridDim.x - npro) {
    hgs_prologue_block(pro, blockIdx.x - (gridDim.x - npro), npro, im.status + HGS_ST_SCANPTR_LO);
    return;
  }
  const int nb_seg = (a.P + HGS_BLOCK - 1) / HGS_BLOCK;
  if ((int)blockIdx.x >= nb_seg) {   // smoothness partial sums over the same endpoints
    hgs_smooth_fwd_block((int)blockIdx.x - nb_seg, fu.n_smooth, st.ep, fu.smooth_pairs, fu.cos_threshold, fu.eps, fu.smooth_partials,
                         (float*)red, (float4*)fu.smooth_pair_grads);
    return;
  }
  if (pro.table) {
    const HgsViewTargets* row = pro.table + pro.view;
    a.viewmatrix = row->viewmatrix; a.projmatrix = row->projmatrix; a.campos = row->campos;
  }
  preprocess_fwd_body<SRC_STRAND>(a, g, im, radii, st, red, th);
}

// The same for the Stage-I cloud (hgs_cloud_forward_preprocess: cloud_fwd_kernel + preprocess_fwd_kernel; no smoothness term)
__global__ __launch_bounds__(HGS_BLOCK) void cloud_preprocess_fwd_kernel(HgsFwdArgs a, HgsGeom g, HgsImage im, int* radii,
                                                                         HgsParamSrc st, HgsPrologue pro) {
  __shared__ uint32_t red[4];
  __shared__ TileHash th;
  const unsigned npro = pro.table ? hgs_prologue_blocks(pro.zero_bytes / 4) : 0u;
  if (blockIdx.x >= gridDim.x - npro) {
    hgs_prologue_block(pro, blockIdx.x - (gridDim.x - npro), npro, im.status + HGS_ST_SCANPTR_LO);
    return;
  }
  if (pro.table) {
    const HgsViewTargets* row = pro.table + pro.view;
    a.viewmatrix = row->viewmatrix; a.projmatrix = row->projmatrix; a.campos = row->campos;
  }
  preprocess_fwd_body<SRC_CLOUD>(a, g, im, radii, st, red, th);
}

// The scatter kernel's LDS: a 512-slot tile table and one record per Gaussian of the block (what an instance needs of its
// Gaussian), so that instances can be dealt to the threads evenly (see the kernel).
#ifndef SC_TH_LOG
#define SC_TH_LOG 9
#endif
using ScTable = TileHashT<SC_TH_LOG>;
struct ScRec { uint32_t x0y0, w, depth; float x, y, hx, hy, nx, ny, rn; int mode; };

// development aid (tools/dev/scatter_trace.py; build with -DHGS_SCATTER_TRACE=1): per workgroup of the scatter kernel the 10 ns
// ticks at which it entered, had its loads and block prefix, had counted its tiles, had reserved its slots, saw the scan, ended
#ifndef HGS_SCATTER_TRACE
#define HGS_SCATTER_TRACE 0
#endif
#if HGS_SCATTER_TRACE
#define SC_TRACE_MAX 8192
__device__ unsigned long long g_sc_trace[SC_TRACE_MAX][8];
#define SC_MARK(k) do { if (threadIdx.x == 0 && blockIdx.x < SC_TRACE_MAX) g_sc_trace[blockIdx.x][k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SC_MARK(k) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------
// scatter: one lane per Gaussian.  Exclusive instance offset = block prefix (scan kernel) + in-block scan;
// every touched tile gets key = depth_bits<<32 | gaussian_id appended to the tile's segment (order inside the
// segment is irrelevant: the per-tile sort key is unique).
//
// Fused scan (capacity mode, status[2..3] != 0): no scan launch ran.  That one-workgroup kernel sat alone on the GPU for
// 9-11 us between two grid-wide kernels; here EVERY workgroup scans the T tile counts itself (32 loads per thread at
// 1080p, all L2 hits, offsets kept in LDS) and sums the block sums before it; workgroup 0 also publishes `ranges`, the
// instance count and its sticky maximum.  Same integers either way.
// (round 6) PARTS.  A workgroup places at most HGS_SC_PART instances.  The first HGS_SC_PART instances of a block of 256
// Gaussians are its own workgroup's; a block with more -- the near, large Gaussians of a Stage-I cloud or of a merged strand
// model: 2750 instances in the median block, 19000 in the largest (tools/dev/instance_stats.py), and the launch lasted as long as
// that one (tools/dev/scatter_trace.py: 90 % of the workgroups done after 45 us of 254) -- is finished by HELPER workgroups at
// the end of the grid, one per further part.  Every workgroup derives the same partition from the per-block instance counts
// the preprocess kernel left (block_sums): helper h finds its (block, part) by one pass over them.  A helper loads the block's
// Gaussians like its owner, counts / reserves / places its share through its own tile table; the per-Gaussian duties (offsets,
// record templates) stay with the owner.  Order inside a tile's segment is irrelevant (the per-tile sort key is unique).
#ifndef HGS_SC_PART
#define HGS_SC_PART 4096u
#endif
__device__ __forceinline__ uint32_t hgs_sc_parts(uint32_t block_sum) { return block_sum <= HGS_SC_PART ? 1u : (block_sum + HGS_SC_PART - 1u) / HGS_SC_PART; }

__global__ __launch_bounds__(HGS_BLOCK) void scatter_kernel(int P, int gx, int T, uint32_t Rcap, const float* __restrict__ feat,
                                                            const float* __restrict__ extra, HgsGeom g, HgsImage im,
                                                            HgsBinning b, int scan_wg) {
  __shared__ uint32_t wsum[4];
  // one LDS block, carved: the tile table (512 slots: a block's instances fall into a few dozen to ~150 distinct tiles), the
  // per-Gaussian records and the instance prefix of the balanced loops below; a scan workgroup keeps its share's offsets in it
  __shared__ __align__(16) unsigned char sc_lds[sizeof(ScTable) + HGS_BLOCK * sizeof(ScRec) + (HGS_BLOCK + 4) * sizeof(uint32_t)];
  ScTable& th = *(ScTable*)sc_lds;
  ScRec* srec = (ScRec*)(sc_lds + sizeof(ScTable));
  uint32_t* ioff = (uint32_t*)(sc_lds + sizeof(ScTable) + HGS_BLOCK * sizeof(ScRec));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long parked = ((unsigned long long)im.status[HGS_ST_SCANPTR_HI] << 32) | im.status[HGS_ST_SCANPTR_LO];
  const unsigned long long report = parked & ~1ull;
  const bool fused = report != 0ull, row_runs = (parked & 1ull) != 0ull;
  int bid = (int)blockIdx.x - scan_wg;             // Gaussian block of this workgroup; < 0: a scan workgroup
  const int nblk_g = (P + HGS_BLOCK - 1) / HGS_BLOCK;
  uint32_t my_part = 0u;                           // which HGS_SC_PART instances of the block this workgroup places
  SC_MARK(0);
  if (bid >= nblk_g) {
    // ---- a helper: which (block, part)?  Thread t sums the further parts of a contiguous share of the blocks; a block scan of
    // the shares finds the thread whose share holds part number h, which walks its share once more.
    __shared__ uint32_t h_found[2];
    const uint32_t h = (uint32_t)(bid - nblk_g);
    const int per_t = (nblk_g + HGS_BLOCK - 1) / HGS_BLOCK;
    const int j0 = (int)threadIdx.x * per_t, j1 = min(nblk_g, j0 + per_t);
    // (block_sums holds the blocks' raw instance counts when the scan is fused into this kernel, their exclusive prefix after a
    // scan_kernel launch: blocking mode)
    const uint32_t R_all = fused ? 0u : im.status[HGS_ST_R];
    auto block_count = [&](int j) -> uint32_t {
      if (fused) return g.block_sums[j];
      return (j + 1 < nblk_g ? g.block_sums[j + 1] : R_all) - g.block_sums[j];
    };
    uint32_t mine = 0u;
    for (int j = j0; j < j1; j++) mine += hgs_sc_parts(block_count(j)) - 1u;
    const uint32_t inc = hgs_wave_incl_scan(mine, lane);
    if (lane == 63) wsum[wave] = inc;
    if (threadIdx.x == 0) h_found[0] = 0xFFFFFFFFu;
    __syncthreads();
    uint32_t base = inc - mine;
    for (int w = 0; w < wave; w++) base += wsum[w];
    if (h >= base && h < base + mine) {
      uint32_t acc = base;
      for (int j = j0; j < j1; j++) {
        const uint32_t ex = hgs_sc_parts(block_count(j)) - 1u;
        if (h < acc + ex) { h_found[0] = (uint32_t)j; h_found[1] = h - acc + 1u; break; }
        acc += ex;
      }
    }
    __syncthreads();
    if (h_found[0] == 0xFFFFFFFFu) return;         // more helpers than further parts: nothing to do
    bid = (int)h_found[0];
    my_part = h_found[1];
    __syncthreads();                               // (wsum is reused below)
  }
  if (bid < 0) {
    // ---- fused scan: `scan_wg` extra workgroups (dispatched first) scan the tile counts and publish `ranges`, the chunk
    // work items of long lists, the instance count and its sticky maximum, while the others load, count and reserve; they
    // need the offsets only when they place their keys.  (Round 1 had every workgroup scan all T counts itself: 8 of a
    // workgroup's 18 us.  Round 2: one scan workgroup, 8 us until its flag -- the others were through with their own 6.6 us
    // of loads, counting and reservation by then and waited.  Round 3: the tiles are shared between HGS_SCAN_WGS
    // workgroups; each gathers, scans and publishes only its share (16 KB of agent-scope stores instead of 64); the totals
    // of the shares in front travel through three status words.)
    if (!fused) return;
    // Round 5: the tiles' segments are ALLOCATED, not scanned.  Rounds 2-4 laid the segments out in tile order: an exclusive
    // scan of all T counts, shared by four workgroups that each needed the totals of the shares in front of theirs (three status
    // words, a chain of agent-scope round trips) and kept their offsets in LDS to publish them coalesced -- done ~8.5 us into
    // the launch, 2 us after the other workgroups had counted and reserved (tools/dev/scatter_trace.py).  Nothing needs the
    // tile order: a segment only has to be contiguous and its own (the per-tile sort and the blend go through `ranges`).  So:
    // one tile per thread, an inclusive scan inside the wavefront, ONE returning atomic per wavefront on the pass's
    // allocation cursor (status word HGS_ST_ALLOC) for the wavefront's 64 tiles, one coalesced store of the ranges -- three
    // dependent round trips, no workgroup waits for another.  The segments of a frame then sit in the order the wavefronts'
    // atomics arrived: the binning buffer's layout differs run to run, every tile's list and everything computed from it does
    // not (tests/test_gpu_raster.py::test_capacity_mode_binning_equals_blocking_mode compares tile by tile; the blocking mode,
    // scan_kernel, keeps the reference's layout).
    static_assert(HGS_FUSED_SCAN_MAX_T <= HGS_SCAN_WGS * HGS_BLOCK, "one tile per thread of the scan workgroups");
    const int t = (int)blockIdx.x * HGS_BLOCK + (int)threadIdx.x;
    const uint32_t slot = HGS_TILE_SLOT(t, im.tile_mask);
    uint32_t cnt = t < T ? im.tile_count[slot] : 0u;
    // the counter is dead from here on: leave it at zero for the next pass over this image buffer (whose first kernel may
    // count into it beside the workgroups that clear the rest of the buffer's counters: hair_preprocess_fwd_kernel)
    if (cnt) im.tile_count[slot] = 0u;
    if (row_runs) {
      // (round 6, HGS_COUNT_ROW_RUNS) the preprocess launch counted its large rectangles as +1 / -1 marks per tile row: a tile's
      // count is its counter plus the running sum of the marks up to it, in row-major tile order.  Every scan workgroup sums the
      // marks in front of its tiles itself (at most 8160 coalesced words out of the L2) -- no workgroup waits for another.
      __shared__ int rr_s[8];
      int front = 0;
      for (int i = (int)threadIdx.x; i < (int)blockIdx.x * HGS_BLOCK; i += HGS_BLOCK) front += im.tile_delta[i];
      const int d = t < T ? im.tile_delta[t] : 0;
      const int d_inc = (int)hgs_wave_incl_scan((uint32_t)d, lane), f_inc = (int)hgs_wave_incl_scan((uint32_t)front, lane);
      if (lane == 63) { rr_s[wave] = d_inc; rr_s[4 + wave] = f_inc; }
      __syncthreads();
      int run = d_inc + rr_s[4] + rr_s[5] + rr_s[6] + rr_s[7];
      for (int w = 0; w < wave; w++) run += rr_s[w];
      if (t < T) cnt += (uint32_t)run;
    }
    const uint32_t inc = hgs_wave_incl_scan(cnt, lane);
    const uint32_t wave_total = (uint32_t)__shfl((int)inc, 63, 64);
    uint32_t base = 0u;
    if (lane == 63 && wave_total)
      base = __hip_atomic_fetch_add(&im.status[HGS_ST_ALLOC], wave_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    base = (uint32_t)__shfl((int)base, 63, 64);
    const uint32_t o = base + inc - cnt;
    // Agent-scope stores: the other workgroups of this launch read them, from other XCDs too.
    if (t < T) hgs_st_agent((unsigned long long*)&im.ranges[t], cnt ? ((unsigned long long)(o + cnt) << 32) | o : 0ull);
    hgs_drain_stores();
    __syncthreads();
    __shared__ uint32_t last_s;
    if (threadIdx.x == 0) {
      const uint32_t done = __hip_atomic_fetch_add(&im.status[HGS_ST_SCAN_DONE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (done == (uint32_t)scan_wg - 1u) {   // the last one: every wavefront's allocation has returned
        const uint32_t end_all = hgs_ld_agent(&im.status[HGS_ST_ALLOC]);
        im.status[HGS_ST_R] = end_all;
        atomicMax((unsigned int*)report, end_all);   // sticky maximum for graph replays (hgs.h)
      }
      last_s = done == (uint32_t)scan_wg - 1u ? 1u : 0u;
    }
    if (row_runs) {   // the last scan workgroup (all of them have read their marks) leaves the marks at zero for the next pass
      __syncthreads();
      if (last_s)
        for (int i = (int)threadIdx.x; i <= T; i += HGS_BLOCK) im.tile_delta[i] = 0;
    }
    if (t < T) hgs_emit_sort_items((uint32_t)t, cnt, (uint32_t)T, im);   // long lists: one sort workgroup per chunk (read by the NEXT kernel)
    SC_MARK(5);
    return;
  }
  th_init(th);
  const int idx = bid * HGS_BLOCK + threadIdx.x;
  uint32_t part = 0;
  if (fused)
    for (int j = threadIdx.x; j < bid; j += HGS_BLOCK) part += g.block_sums[j];   // raw sums (no scan kernel ran)
  const uint32_t n = idx < P ? g.tiles_touched[idx] : 0;
  HgsRect rc = {0, 0, 0, 0, 0, 0};
  // this lane's Gaussian: loaded now, with everything else the prologue needs, used after the scans (for a culled
  // Gaussian these slots hold whatever the previous pass left: never used, n == 0)
  float2 xy = make_float2(0.f, 0.f);
  float4 co = make_float4(0.f, 0.f, 0.f, 0.f), ex = make_float4(0.f, 0.f, 0.f, 0.f);
  float depth = 0.f, f0 = 0.f, f1 = 0.f, f2 = 0.f;
  if (idx < P) {
    rc = g.rect[idx];
    xy = g.means2D[idx];
    co = g.conic_opacity[idx];
    depth = g.depths[idx];
    f0 = feat[3 * (size_t)idx]; f1 = feat[3 * (size_t)idx + 1]; f2 = feat[3 * (size_t)idx + 2];
    if (extra) ex = ((const float4*)extra)[idx];
  }
  uint32_t blk_base = 0;
  if (fused) {
    blk_base = block_sum_256(part, wsum);
    __syncthreads();                                 // wsum is reused
  }
  const uint32_t incl = hgs_wave_incl_scan(n, lane);
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  uint32_t base = fused ? blk_base : g.block_sums[bid];  // exclusive block prefix
  for (int w = 0; w < wave; w++) base += wsum[w];
  if (idx < P && my_part == 0u) {
    const uint32_t off_incl = base + incl;
    g.point_offsets[idx] = off_incl;
    if (n != 0) {
      rc.off = off_incl - n;
      g.rect[idx] = rc;
    }
  }
  if (n == 0) rc = HgsRect{0, 0, 0, 0, 0, 0};
  // Round 4: the instances of the block's Gaussians are dealt to the threads EVENLY.  Rounds 1-3 let every lane walk its own
  // rectangle, so a workgroup took as long as its largest Gaussian: on the state 1000 iterations of training leave (2.9
  // instances per Gaussian on average, up to 16 per lane) the key placement took 9.5 us per workgroup on average and 20 at
  // most for 3.9 at initialisation (tools/dev/scatter_trace.py), and the launch 52 us for 15.  Every Gaussian leaves a
  // record in LDS; thread t takes the instances [t, t + 1) * ceil(total / 256) of the block's concatenated rectangles
  // (one binary search over the prefix, then a walk) in both passes.  Order inside a tile's segment is irrelevant: the
  // per-tile sort key is unique.  Large Gaussians are dealt out like the others (a 36-tile Gaussian walked by its own lane,
  // one returning global atomic per tile, was what the last workgroup of a launch was still doing 19 us after the median
  // one had finished); an instance whose tile finds the table full takes the direct global atomic.
  const uint32_t ns = n;
  const uint32_t incs = hgs_wave_incl_scan(ns, lane);
  __syncthreads();                                   // wsum is reused
  if (lane == 63) wsum[wave] = incs;
  const HgsQuadCull qc = hgs_quad_cull(co);
  {
    ScRec r;
    r.x0y0 = (uint32_t)rc.x0 | ((uint32_t)rc.y0 << 16); r.w = (uint32_t)(rc.x1 - rc.x0); r.depth = __float_as_uint(depth);
    r.x = xy.x; r.y = xy.y; r.hx = qc.hx; r.hy = qc.hy; r.nx = qc.nx; r.ny = qc.ny; r.rn = qc.rn; r.mode = qc.mode;
    srec[threadIdx.x] = r;
  }
  __syncthreads();
  uint32_t sbase = 0, stotal = 0;
  for (int w = 0; w < 4; w++) { if (w < wave) sbase += wsum[w]; stotal += wsum[w]; }
  ioff[threadIdx.x] = sbase + incs - ns;
  if (threadIdx.x == HGS_BLOCK - 1) ioff[HGS_BLOCK] = stotal;
  __syncthreads();
  SC_MARK(1);
  // this workgroup's part of the block's instances, [p_lo, p_hi), and this thread's share of it, [k0, k1), which starts inside
  // Gaussian j0 at its l0-th tile
  // (a launch without helper workgroups -- a pass that cannot hold many instances per Gaussian -- has one part per block)
  const bool split = (int)gridDim.x > scan_wg + nblk_g && hgs_sc_parts(stotal) > 1u;
  const uint32_t p_lo = split ? min(stotal, my_part * HGS_SC_PART) : 0u;
  const uint32_t p_hi = split ? min(stotal, p_lo + HGS_SC_PART) : stotal;
  const uint32_t per = (p_hi - p_lo + HGS_BLOCK - 1) / HGS_BLOCK;
  const uint32_t k0 = min(p_hi, p_lo + threadIdx.x * per), k1 = min(p_hi, k0 + per);
  int j0 = 0;
  if (k0 < k1) {
    int lo = 0, hi = HGS_BLOCK;                       // largest j with ioff[j] <= k0 (its Gaussian has an instance there)
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (ioff[mid] <= k0) lo = mid; else hi = mid; }
    j0 = lo;
  }
  const uint32_t l0 = k0 < k1 ? k0 - ioff[j0] : 0u;
  auto for_my_instances = [&](auto&& fn) {
    if (k0 >= k1) return;
    int j = j0;
    uint32_t l = l0, nj = ioff[j + 1] - ioff[j];
    ScRec r = srec[j];
    int tx = (int)(r.x0y0 & 0xFFFFu) + (int)(l % r.w), ty = (int)(r.x0y0 >> 16) + (int)(l / r.w);
    for (uint32_t k = k0; k < k1; k++) {
      while (l >= nj) {                               // next Gaussian with an instance
        j++; l = 0; nj = ioff[j + 1] - ioff[j];
        if (nj) { r = srec[j]; tx = (int)(r.x0y0 & 0xFFFFu); ty = (int)(r.x0y0 >> 16); }
      }
      fn(j, r, tx, ty);
      l++;
      if (++tx == (int)((r.x0y0 & 0xFFFFu) + r.w)) { tx = (int)(r.x0y0 & 0xFFFFu); ty++; }
    }
  };
  // pass 1: count this block's instances per tile in LDS
  for_my_instances([&](int, const ScRec&, int tx, int ty) {
    const int sl = th_insert(th, (uint32_t)(ty * gx + tx));
    if (sl >= 0) atomicAdd(&th.cnt[sl], 1u);
  });
  __syncthreads();
  SC_MARK(2);
  // one global atomic per distinct tile reserves the block's slots in that tile's segment
  for (int i = threadIdx.x; i < ScTable::SIZE; i += HGS_BLOCK)
    if (th.key[i] != TH_EMPTY) { th.base[i] = atomicAdd(&im.tile_cursor[HGS_TILE_SLOT(th.key[i], im.tile_mask)], th.cnt[i]); th.cnt[i] = 0u; }
#if HGS_SCATTER_TRACE
  __syncthreads();
  SC_MARK(3);
#endif
  if (fused) {
    // the offsets of the scan workgroup are needed from here on (it was dispatched first and has had this workgroup's
    // loads, counting and reservation to finish; bounded wait: HGS_ST_TIMEOUT / HGS_WAIT_TIMED_OUT instead of a hung GPU)
    if (threadIdx.x == 0) {
      int spin = 0;
      while (hgs_ld_agent(&im.status[HGS_ST_SCAN_DONE]) < (uint32_t)scan_wg && ++spin < (1 << 21)) __builtin_amdgcn_s_sleep(4);
      if (spin >= (1 << 21)) { im.status[HGS_ST_TIMEOUT] = 1u; atomicMax((unsigned int*)report, 0xFFFFFFFFu); }
    }
  }
  __syncthreads();
  // (round 6) a tile's segment start is read ONCE per distinct tile and block -- into the table entry that already holds the
  // block's reservation inside the segment -- instead of once per instance: the agent-scope load (it bypasses the XCD's L2: the
  // scan workgroups may sit on another XCD) was the dependent round trip of every key placed; on many-tile states (a Stage-I
  // cloud at 1080p: 12 instances per Gaussian, 3000 per workgroup) the placement loop was a chain of them
  // (only where a block places more than a few keys per thread: a pass over the table and a barrier otherwise cost more than
  // the two or three loads per thread they replace -- north_star: 14.2 -> 15.4 us with it everywhere)
  const bool cached_start = p_hi - p_lo > 4u * HGS_BLOCK;
  if (cached_start) {
    for (int i = threadIdx.x; i < ScTable::SIZE; i += HGS_BLOCK)
      if (th.key[i] != TH_EMPTY) th.base[i] += fused ? hgs_ld_agent(&im.ranges[th.key[i]].x) : im.ranges[th.key[i]].x;
    __syncthreads();
  }
  SC_MARK(4);
  if (n != 0 && my_part == 0u) {
    // this Gaussian's instance-record template (HgsGeom::grec), read back once per instance by the sort kernel
    float4* rec = g.grec + 4 * (size_t)idx;
    rec[0] = make_float4(xy.x, xy.y, co.x, co.y);
    rec[1] = make_float4(co.z, co.w, f0, f1);
    rec[2] = make_float4(f2, ex.x, ex.y, ex.z);
    rec[3] = make_float4(ex.w, __uint_as_float(rc.off), __uint_as_float((uint32_t)rc.x0 | ((uint32_t)rc.y0 << 16)),
                         __uint_as_float((uint32_t)(rc.x1 - rc.x0)));
  }
  // pass 2: place the keys
  const uint32_t idx0 = (uint32_t)bid * HGS_BLOCK;
  auto place = [&](uint64_t key0, const HgsQuadCull& q, float2 c, int tx, int ty) {
    const uint32_t t = (uint32_t)(ty * gx + tx);
    const uint64_t key = key0 | hgs_quadrant_mask(q, c, tx, ty);
    const int sl = th_find(th, t);
    const uint32_t start = (sl >= 0 && cached_start) ? 0u : (fused ? hgs_ld_agent(&im.ranges[t].x) : im.ranges[t].x);
    const uint32_t pos = start + (sl >= 0 ? th.base[sl] + atomicAdd(&th.cnt[sl], 1u)
                                          : atomicAdd(&im.tile_cursor[HGS_TILE_SLOT(t, im.tile_mask)], 1u));
    if (pos < Rcap) b.keys[pos] = key;
    else im.status[HGS_ST_OVERFLOW] = 1;  // overflow: caller under-sized the binning buffer
  };
  for_my_instances([&](int j, const ScRec& r, int tx, int ty) {
    HgsQuadCull q;
    q.tau = 0.f; q.hx = r.hx; q.hy = r.hy; q.nx = r.nx; q.ny = r.ny; q.rn = r.rn; q.mode = r.mode;
    place(((uint64_t)r.depth << 32) | ((idx0 + (uint32_t)j) << HGS_QMASK_SHIFT), q, make_float2(r.x, r.y), tx, ty);
  });
#if HGS_SCATTER_TRACE
  __syncthreads();
  SC_MARK(5);
#endif
}

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ V3 dnormvdv(V3 v, V3 dv) {  // auxiliary.h:107-117
  const float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
  const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
  V3 r;
  r.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
  r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
  r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
  return r;
}


// ---- (round 6) a wavefront's instance rows summed by the WHOLE wavefront --------------------------------------------------
// The rows of consecutive Gaussians are consecutive in the scratch (a Gaussian's slot range starts at the exclusive prefix of
// tiles_touched over the Gaussians in front of it: rc.off), so the rows of a wavefront's 64 Gaussians form ONE contiguous run
// of R_w rows with a segment per Gaussian.  Rounds 1-5 gave every lane its own segment: a wavefront took as long as its longest
// one.  That is nothing on fresh strands (1.7 rows per Gaussian) and most of the step on the states the three-stage workflow
// lives in: a Stage-I cloud at 1080p has 12.8 rows per Gaussian but 67 in the mean wavefront's longest lane (p99: 548, max 4240),
// the merged Stage-III start model 5.8 / 38 (tools/dev/instance_stats.py) -- preprocess_bwd_kernel 725 / 226 us where the
// north_star step's takes 14.  Here the run is streamed: 16 rows per load (lane = quarter * 16 + row: four lanes share a row of
// 64 bytes, 1 KB contiguous per instruction, four loads in flight), a segmented inclusive scan over the 16 rows inside each
// 16-lane DPP row (row_shr 1, 2, 4, 8; the segment heads of a chunk travel through 16 words of LDS), and the last row of every
// segment in the chunk adds the segment's partial sum to the owner's accumulator in LDS.  The association of a Gaussian's sum
// then depends on where its rows fall in the 16-row chunks (fixed for a given pass: bitwise reproducible; different from the
// in-lane loop's, which waves whose longest segment is short keep).
// (a wavefront streams its run when its longest segment exceeds this many rows.  Same box, 8 / 16 / 32 / 64 / never: C2 -- five
// evenly spread instances per Gaussian -- 23.4 / 15.8 / 15.7 / 15.8 / 15.8 us, i.e. streaming a wavefront whose lanes are busy
// alike costs 50 % more than their loops; stage3_merged without row_reduce_kernel 85 / 93 / 84 / 87 / 220; stage1_1080p 366 /
// 380 / 311 / 370 / 727)
#ifndef HGS_PPB_LIGHT_MAX
#define HGS_PPB_LIGHT_MAX 32
#endif
__device__ __forceinline__ float hgs_dpp_shr_f(float v, int d) {   // lane i <- lane i - d inside its 16-lane row, 0 where there is none
  const int x = __float_as_int(v);
  int r;
  switch (d) {
    case 1: r = __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false); break;
    case 2: r = __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false); break;
    case 4: r = __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false); break;
    default: r = __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false); break;
  }
  return __int_as_float(r);
}
__device__ __forceinline__ int hgs_dpp_shr_i(int x, int d) {
  switch (d) {
    case 1: return __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);
    case 2: return __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);
    case 4: return __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);
    default: return __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);
  }
}
// All 64 lanes of the wavefront must be active.  `base`: the run's first row (wave-uniform), rq = float4 per row (3 or 4),
// nr = this lane's segment length (0: none).  acc: [64][4] float4 of LDS owned by this wavefront, head: 16 words of it.
// Returns the lane's own 16 sums in out[0..3].
__device__ __forceinline__ void hgs_stream_rows(const float4* __restrict__ base, int rq, uint32_t nr, int lane, float4 (*acc)[4],
                                                uint32_t* head, float4 out[4]) {
  const uint32_t inc = hgs_wave_incl_scan(nr, lane);
  const uint32_t pre = inc - nr;
  const uint32_t Rw = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
  const int g = lane & 15, q = lane >> 4;
#pragma unroll
  for (int k = 0; k < 4; k++) acc[lane][k] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (lane < 16) head[lane] = 0xFFFFFFFFu;
  const uint32_t nchunks = (Rw + 15u) >> 4;
  auto ld = [&](uint32_t c) -> float4 {
    const uint32_t r = c * 16u + (uint32_t)g;
    return (r < Rw && q < rq) ? base[(size_t)r * rq + q] : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  int carry = 0;                                                  // owner + 1 of the row in front of the chunk (wave-uniform)
  float4 b0 = ld(0), b1 = ld(1), b2 = ld(2), b3 = ld(3);
  for (uint32_t c = 0; c < nchunks; c++) {
    float4 v = b0;
    b0 = b1; b1 = b2; b2 = b3; b3 = ld(c + 4u);
    const uint32_t r0 = c * 16u;
    if (nr != 0u && pre - r0 < 16u) head[pre - r0] = (c << 8) | (uint32_t)lane;   // (pre >= r0 and pre < r0 + 16)
    const uint32_t h = head[g];                                   // (LDS operations of a wavefront execute in order)
    const bool is_head = (h >> 8) == c;
    int own = is_head ? (int)(h & 63u) + 1 : 0;
    int f = is_head ? 1 : 0;
    const int f0 = f;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
      const float ux = hgs_dpp_shr_f(v.x, d), uy = hgs_dpp_shr_f(v.y, d), uz = hgs_dpp_shr_f(v.z, d), uw = hgs_dpp_shr_f(v.w, d);
      const int fu = hgs_dpp_shr_i(f, d), ou = hgs_dpp_shr_i(own, d);
      if (!f) { v.x += ux; v.y += uy; v.z += uz; v.w += uw; }
      f |= fu;
      own = max(own, ou);
    }
    if (own == 0) own = carry;
    // the last row of a segment inside this chunk: the next row starts another, or the chunk / the run ends
    const int next_head = __builtin_amdgcn_update_dpp(1, f0, 0x101, 0xF, 0xF, false);   // row_shl:1 (lane 15 of a row: 1)
    const uint32_t r = r0 + (uint32_t)g;
    const bool tail = r < Rw && (next_head != 0 || r + 1u == Rw);
    if (tail && q < rq && own > 0) {
      float4 a = acc[own - 1][q];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      acc[own - 1][q] = a;
    }
    carry = __builtin_amdgcn_readlane(own, 15);
  }
#pragma unroll
  for (int k = 0; k < 4; k++) out[k] = acc[lane][k];
}


// ---- (round 6) row_reduce_kernel: the per-Gaussian sums of the instance rows as a launch of its own, balanced by ROWS ------
// hgs_stream_rows balances a wavefront's 64 Gaussians, not the wavefronts: on the states the three-stage workflow lives in the
// rows per wavefront spread from 670 (median) to 6400, the rows per workgroup from 2750 to 19000 (tools/dev/instance_stats.py: a
// Stage-I cloud at 1080p) and the launch lasts as long as its heaviest workgroup (365 us where the bytes take 40).  This kernel
// cuts the scratch into runs of HGS_RR_RPW rows, one per wavefront, whatever Gaussians they belong to.  A row names its Gaussian
// in its sixteenth float (blend_bwd_kernel<7>), so a run needs no offsets: 16 rows per load as in hgs_stream_rows, a segment
// head wherever the id changes, the segmented scan inside the 16-lane DPP rows, the open segment's sum carried from chunk to
// chunk in registers.  A segment that ends inside the run is stored once: to gsum[id] when it also began inside it, to the run's
// `first` partial when it came in from the run in front; the segment still open at the end of the run goes to its `last`
// partial (a segment that spans a whole run is its `first`).  preprocess_bwd_kernel then reads ONE row per Gaussian -- or, for
// the few Gaussians whose rows cross a run boundary, last(w0) + first(w0 + 1) + ... + first(w1), in that order: every sum is a
// fixed sequence of additions for a given pass (bitwise reproducible).
__global__ __launch_bounds__(HGS_BLOCK) void row_reduce_kernel(const float4* __restrict__ rows, const uint32_t* __restrict__ status,
                                                               uint32_t Rcap, float4* __restrict__ gsum, float4* __restrict__ partial,
                                                               uint32_t P) {
  const int lane = threadIdx.x & 63, g = lane & 15, q = lane >> 4;
  const uint32_t w = blockIdx.x * (HGS_BLOCK / 64) + (threadIdx.x >> 6);
  if (status[HGS_ST_OVERFLOW] != 0u) return;                        // a void pass: nothing was written, nothing is read
  const uint32_t R = min(status[HGS_ST_R], Rcap);
  const uint32_t r_lo = w * (uint32_t)HGS_RR_RPW;
  if (r_lo >= R) return;
  const uint32_t r_hi = min(R, r_lo + (uint32_t)HGS_RR_RPW);
  const uint32_t* ids = (const uint32_t*)rows;                       // id of row r: word 16 r + 15
  const uint32_t NONE = 0xFFFFFFFFu;
  const uint32_t id_before = r_lo > 0u ? ids[(size_t)(r_lo - 1u) * 16 + 15] : NONE;
  const uint32_t id_after = r_hi < R ? ids[(size_t)r_hi * 16 + 15] : NONE;
  const uint32_t nchunks = (r_hi - r_lo + 15u) >> 4;
  auto ld = [&](uint32_t c, uint32_t& id) -> float4 {
    const uint32_t r = r_lo + c * 16u + (uint32_t)g;
    if (r < r_hi) { id = ids[(size_t)r * 16 + 15]; return rows[(size_t)r * 4 + q]; }
    id = NONE;
    return make_float4(0.f, 0.f, 0.f, 0.f);
  };
  uint32_t i0, i1, i2, i3;
  float4 b0 = ld(0, i0), b1 = ld(1, i1), b2 = ld(2, i2), b3 = ld(3, i3);
  // the segment that is open in front of the current chunk (wave-uniform): its id, whether it is the one that came in from the
  // run in front, whether any of its rows lie in this run yet (then `carry` holds their sum: lane g == 0 of every quarter)
  uint32_t open_id = id_before;
  bool open_from_before = true, have_carry = false;
  float4 carry = make_float4(0.f, 0.f, 0.f, 0.f);
  auto store_segment = [&](uint32_t id, bool from_before, bool to_after, const float4& v) {
    if (from_before) partial[((size_t)w * 2 + 0) * 4 + q] = v;
    else if (to_after) partial[((size_t)w * 2 + 1) * 4 + q] = v;
    else if (id < P) gsum[(size_t)id * 4 + q] = v;
  };
  for (uint32_t c = 0; c < nchunks; c++) {
    float4 v = b0;
    const uint32_t id = i0;
    b0 = b1; i0 = i1; b1 = b2; i1 = i2; b2 = b3; i2 = i3; b3 = ld(c + 4u, i3);
    const uint32_t r = r_lo + c * 16u + (uint32_t)g;
    const bool valid = r < r_hi;
    const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)open_id, (int)id, 0x111, 0xF, 0xF, false);   // row_shr:1 (g == 0: open_id)
    const bool is_head = valid && id != prev;
    if (g == 0) {
      // the open segment goes on through row 0 (the carry joins it), or it ended with the chunk in front (stored now)
      if (!is_head) { v.x += carry.x; v.y += carry.y; v.z += carry.z; v.w += carry.w; }
      else if (have_carry) store_segment(open_id, open_from_before, false, carry);
    }
    int f = is_head ? 1 : 0;
    const int f0 = f;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
      const float ux = hgs_dpp_shr_f(v.x, d), uy = hgs_dpp_shr_f(v.y, d), uz = hgs_dpp_shr_f(v.z, d), uw = hgs_dpp_shr_f(v.w, d);
      const int fu = hgs_dpp_shr_i(f, d);
      if (!f) { v.x += ux; v.y += uy; v.z += uz; v.w += uw; }
      f |= fu;
    }
    // rows that end their segment inside the chunk: the next row starts another one, or the run ends here.  (Row 15 of a chunk
    // that is not the run's last never does: its segment stays open and is settled by the next chunk's row 0.)
    const int next_head = __builtin_amdgcn_update_dpp(0, f0, 0x101, 0xF, 0xF, false);   // row_shl:1 (g == 15: 0)
    const bool last_row = valid && r + 1u == r_hi;
    const bool from_carry = f == 0;                                    // no head at or in front of this row in the chunk
    if (valid && (next_head != 0 || last_row))
      store_segment(id, from_carry && open_from_before, last_row && id == id_after, v);
    // the segment left open behind the chunk is row 15's
    open_from_before = open_from_before && __builtin_amdgcn_readlane(f, 15) == 0;
    open_id = (uint32_t)__builtin_amdgcn_readlane((int)id, 15);
    have_carry = true;
    carry.x = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v.x), 0x121, 0xF, 0xF, false));   // row_ror:1: lane 15 -> lane 0
    carry.y = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v.y), 0x121, 0xF, 0xF, false));
    carry.z = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v.z), 0x121, 0xF, 0xF, false));
    carry.w = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v.w), 0x121, 0xF, 0xF, false));
  }
}

// DC_ONLY: the strand / Stage-I default -- SH degree 0 with one stored coefficient (or precomputed colours): the
// view-dependent SH code is compiled out (136 -> fewer registers for a kernel that lives on its occupancy).
// MODE (round 5, the backward mirror of hair_preprocess_fwd_kernel / cloud_preprocess_fwd_kernel): the lane that has just
// finished Gaussian k's gradients holds everything the parameters' backward needs of it in registers.
//   MODE 0  the rasterizer's own outputs (nine gradient tensors, hgs_backward / hgs_backward_multi);
//   MODE 1  strand model (hgs_backward_multi_params, HGS_PARAMS_HAIR): the segment's geometry backward is applied HERE, once --
//           two endpoint contributions (h - gD, h + gD; hgs_strand_bwd.h), d_width, d_opacity_raw, d_mask_raw, dL_dsh and the
//           densification statistics leave the lane instead of 124 bytes of per-Gaussian gradients that strand_bwd_kernel's
//           segment lane and both of its endpoint lanes would read again; what remains of that kernel is the endpoint gather
//           (hgs_hair_endpoint_gather: two 16-byte loads per endpoint);
//   MODE 2  Stage-I cloud (HGS_PARAMS_CLOUD): raw scaling / rotation / opacity / mask gradients and the statistics from the same
//           lane -- cloud_bwd_kernel's launch is gone; the loss head's deferred tail rides in one spare workgroup here.
// The parameter arithmetic is the shared device code of hgs_strand_bwd.h, evaluated without contraction in both translation
// units: the fused backward's results are the two-launch form's bit for bit (tests/test_gpu_train.py).
template <bool DC_ONLY, int MODE>
// (five waves per SIMD: 87 VGPRs; at six the compiler spills six registers and the kernel takes 11.7 instead of 8.9 us)
#ifndef HGS_PPB_WAVES
#define HGS_PPB_WAVES 5
#endif
// (MODE 1 / 2 with view-dependent SH: 116 VGPRs, four waves per SIMD; the DC-only forms fit five: 88 / 86)
__global__ __launch_bounds__(HGS_BLOCK) __attribute__((amdgpu_waves_per_eu((MODE == 0 || DC_ONLY) ? HGS_PPB_WAVES : HGS_PPB_WAVES - 1))) void preprocess_bwd_kernel(HgsBwdArgs a, HgsGeom g, HgsBinning b,
                                                                   const float* __restrict__ inst_grad, uint32_t Rcap,
                                                                   const uint32_t* __restrict__ status, HgsParamBackward pb) {
  __shared__ float4 s_acc[HGS_BLOCK / 64][64][4];     // hgs_stream_rows: per wavefront, one accumulator row per lane
  __shared__ uint32_t s_head[HGS_BLOCK / 64][16];
  if ((int)(blockIdx.x * HGS_BLOCK) >= a.P) {
    // (the loss head's deferred tail: the spare workgroup behind the launch's own, all of whose lanes are past P)
    if (MODE != 0 && pb.head_tail.out && blockIdx.x == gridDim.x - 1) hgs_head_tail_block(pb.head_tail);
    return;
  }
  // (lanes past P of the last workgroup stay: the row summation below is a wavefront-wide operation; they read Gaussian P - 1
  // and leave before anything is stored)
  const bool live = (int)(blockIdx.x * HGS_BLOCK + threadIdx.x) < a.P;
  const int idx = live ? (int)(blockIdx.x * HGS_BLOCK + threadIdx.x) : a.P - 1;
  const int D = DC_ONLY ? 0 : a.D;
  // A forward that overflowed its binning capacity (status[1]) dropped instances: their rows of the scratch were never
  // written.  Such a pass is void; its backward returns EXACTLY ZERO for every gradient (deterministic, finite) and the
  // caller repeats the step with a larger capacity (include/hgs.h).
  const bool void_pass = status[HGS_ST_OVERFLOW] != 0u;
  const int M = DC_ONLY ? 1 : a.M;
  float dmx = 0.f, dmy = 0.f, dcx = 0.f, dcy = 0.f, dcw = 0.f, dop = 0.f, dcol[3] = {0.f, 0.f, 0.f};
  float dex[4] = {0.f, 0.f, 0.f, 0.f}, dmx_rgb = 0.f, dmy_rgb = 0.f;
  float dmean[3] = {0.f, 0.f, 0.f}, dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float dscale[3] = {0.f, 0.f, 0.f}, drot[4] = {0.f, 0.f, 0.f, 0.f};
  float dsh_dc[3] = {0.f, 0.f, 0.f};               // gradient of the SH DC coefficient (MODE 1 / 2: the in-lane Adam update reads it)
  // this Gaussian's own data: every load issued here, before the row loop below (whose data-dependent trip count the
  // compiler will not move loads across): one memory round trip for all of it instead of one per dependent stage
  const bool vis = live && a.radii[idx] > 0 && !void_pass;
  const HgsRect rc_pre = g.rect[idx];
  const uint32_t n_pre = g.tiles_touched[idx];
  const float4 co_pre = g.conic_opacity[idx];
  const V3 mean_pre = {a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]};
  const float* cov3D_pre = (a.cov3D_precomp ? a.cov3D_precomp : g.cov3D) + 6 * (size_t)idx;
  float cov3_pre[6];
#pragma unroll
  for (int k = 0; k < 6; k++) cov3_pre[k] = cov3D_pre[k];
  float4 q_pre = make_float4(1.f, 0.f, 0.f, 0.f);
  float s_pre[3] = {1.f, 1.f, 1.f};
  if (a.scales) {
    q_pre = ((const float4*)a.rotations)[idx];
    s_pre[0] = a.scales[3 * idx]; s_pre[1] = a.scales[3 * idx + 1]; s_pre[2] = a.scales[3 * idx + 2];
  }
  // (round 4) likewise the wave-uniform matrices and the clamp flags: read where they are used -- behind the row loop, behind
  // the zero-gradient stores of the lanes that are not visible -- they were three more exposed round trips (vector loads: the
  // compiler cannot prove them unwritten there); at entry they are scalar loads
  // (read through the constant address space: wave-uniform and unwritten during the launch, so always scalar loads into
  // SGPRs -- with the loss head's tail among the kernel's code the compiler no longer proves that by itself and the 35 values
  // take VGPRs: 114 instead of 88)
  typedef const __attribute__((address_space(4))) float* ConstF;
  const ConstF vm_c = (ConstF)(uintptr_t)a.viewmatrix, pj_c = (ConstF)(uintptr_t)a.projmatrix, cam_c = (ConstF)(uintptr_t)a.campos;
  float Vm_pre[16], pj_pre[16], cam_pre[3];
#pragma unroll
  for (int k = 0; k < 16; k++) { Vm_pre[k] = vm_c[k]; pj_pre[k] = pj_c[k]; }
#pragma unroll
  for (int k = 0; k < 3; k++) cam_pre[k] = cam_c[k];
  bool clamped_pre[3] = {false, false, false};
  if (a.shs) {
#pragma unroll
    for (int ch = 0; ch < 3; ch++) clamped_pre[ch] = g.clamped[3 * (size_t)idx + ch] != 0;
  }
  // (MODE 1 / 2: the parameter side's own inputs, none of which depends on the rows below)
  HgsSegGeom seg_pre = {0.f, 0.f, 0.f};
  float4 raw_rot_pre = make_float4(1.f, 0.f, 0.f, 0.f);
  float mask_pre = 0.f;
  const int radius_pre = a.radii[idx];
  if (MODE == 1) seg_pre = hgs_segment_geom(idx, pb.endpoints, pb.endpoint_pairs);
  if (MODE == 2) raw_rot_pre = ((const float4*)pb.rotation_raw)[idx];
  if (MODE != 0) mask_pre = pb.extra4[4 * (size_t)idx];
  {
    // ---- deterministic gather of this Gaussian's per-instance partial sums (instance order = tile rect order)
    const HgsRect rc = rc_pre;
    const uint32_t n = n_pre;
    const int row_floats = a.n_extra ? 16 : HGS_INST_GRAD_FLOATS;
    // (under-sized binning buffer: the forward already flagged the overflow; rows beyond the capacity do not exist)
    const uint32_t nr = (!vis || rc.off >= Rcap) ? 0u : min(n, Rcap - rc.off);
    const float4* rows = (const float4*)(inst_grad + (size_t)rc.off * row_floats);   // rows in Gaussian-major order
    const int rq = row_floats / 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long heavy = __ballot(nr > (uint32_t)HGS_PPB_LIGHT_MAX);
    const unsigned long long some = __ballot(nr != 0u);
    if (a.row_sums) {
      // row_reduce_kernel ran in front of this launch: one row per Gaussian -- or the partial sums of the runs its rows cross
      if (nr != 0u) {
        const uint32_t w0 = rc.off / (uint32_t)HGS_RR_RPW, w1 = (rc.off + nr - 1u) / (uint32_t)HGS_RR_RPW;
        float4 o4[4];
        if (w0 == w1) {
          const float4* r = (const float4*)a.row_sums + 4 * (size_t)idx;
          o4[0] = r[0]; o4[1] = r[1]; o4[2] = r[2]; o4[3] = r[3];
        } else {
          const float4* pr = (const float4*)a.row_partials;
          const float4* r = pr + ((size_t)w0 * 2 + 1) * 4;                   // what run w0 left open
          o4[0] = r[0]; o4[1] = r[1]; o4[2] = r[2]; o4[3] = r[3];
          for (uint32_t w = w0 + 1u; w <= w1; w++) {                         // what came into the runs behind it, in order
            const float4* t = pr + ((size_t)w * 2 + 0) * 4;
#pragma unroll
            for (int k = 0; k < 4; k++) { const float4 x = t[k]; o4[k].x += x.x; o4[k].y += x.y; o4[k].z += x.z; o4[k].w += x.w; }
          }
        }
        dmx = o4[0].x; dmy = o4[0].y; dcx = o4[0].z; dcy = o4[0].w;
        dcw = o4[1].x; dop = o4[1].y; dcol[0] = o4[1].z; dcol[1] = o4[1].w; dcol[2] = o4[2].x;
        dex[0] = o4[2].y; dex[1] = o4[2].z; dex[2] = o4[2].w; dex[3] = o4[3].x;
        dmx_rgb = o4[3].y; dmy_rgb = o4[3].z;
      }
    } else if (heavy != 0ull) {
      // a wavefront with a long segment: its whole run of rows is streamed (hgs_stream_rows)
      const int first = (int)__builtin_ctzll(some);
      const unsigned long long bp = (unsigned long long)(uintptr_t)rows;
      const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)bp, first);
      const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(bp >> 32), first);
      const float4* base = (const float4*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
      float4 o4[4];
      hgs_stream_rows(base, rq, nr, lane, s_acc[wave], s_head[wave], o4);
      dmx = o4[0].x; dmy = o4[0].y; dcx = o4[0].z; dcy = o4[0].w;
      dcw = o4[1].x; dop = o4[1].y; dcol[0] = o4[1].z; dcol[1] = o4[1].w; dcol[2] = o4[2].x;
      if (a.n_extra) {
        dex[0] = o4[2].y; dex[1] = o4[2].z; dex[2] = o4[2].w; dex[3] = o4[3].x;
        dmx_rgb = o4[3].y; dmy_rgb = o4[3].z;
      }
    } else {
    // HGS_PPB_ROWS rows in flight per trip, added in instance order (the sums are the same sums; a lane with 30 instances used to
    // pay 30 dependent trips through the cache hierarchy, and its workgroup with it)
#ifndef HGS_PPB_ROWS
#define HGS_PPB_ROWS 2
#endif
    for (uint32_t k = 0; k < nr; k += HGS_PPB_ROWS) {
      float4 q[HGS_PPB_ROWS][4];
#pragma unroll
      for (int u = 0; u < HGS_PPB_ROWS; u++) {
        const uint32_t ku = min(k + (uint32_t)u, nr - 1u);            // (clamped: loaded again, not added)
        const float4* r = rows + (size_t)ku * rq;
        q[u][0] = r[0]; q[u][1] = r[1]; q[u][2] = r[2];
        q[u][3] = a.n_extra ? r[3] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < HGS_PPB_ROWS; u++) {
        if (k + (uint32_t)u < nr) {
          const float4 r0 = q[u][0], r1 = q[u][1], r2 = q[u][2], r3 = q[u][3];
          dmx += r0.x; dmy += r0.y; dcx += r0.z; dcy += r0.w;
          dcw += r1.x; dop += r1.y; dcol[0] += r1.z; dcol[1] += r1.w; dcol[2] += r2.x;
          if (a.n_extra) {  // row = [.., dcolor 0..6, rgb-only dmean2D.xy]
            dex[0] += r2.y; dex[1] += r2.z; dex[2] += r2.w; dex[3] += r3.x;
            dmx_rgb += r3.y; dmy_rgb += r3.z;
          }
        }
      }
    }
    }
  }
  if (vis) {
    // The rows hold sums of moments of u = G dL/dalpha (blend_bwd_kernel): dmx = S(u dx), dmy = S(u dy), dcx = S(u dx dx),
    // dcy = S(u dx dy), dcw = S(u dy dy), dop = S(u).  backward_distwar.cu:1002-1011 in terms of them (dL_dG = opacity
    // dL_dalpha, dG_ddelx = -G (a dx + b dy), dG_ddely = -G (c dy + b dx), ddel_dx = 0.5 W, ddel_dy = 0.5 H):
    {
      const float4 co = co_pre;
      const float sx = 0.5f * a.W * co.w, sy = 0.5f * a.H * co.w;
      const float mx = dmx, my = dmy, mrx = dmx_rgb, mry = dmy_rgb;
      dmx = sx * (-co.x * mx - co.y * my);
      dmy = sy * (-co.z * my - co.y * mx);
      dmx_rgb = sx * (-co.x * mrx - co.y * mry);
      dmy_rgb = sy * (-co.z * mry - co.y * mrx);
      const float h = -0.5f * co.w;
      dcx *= h; dcy *= h; dcw *= h;
    }
    // ---- computeCov2DCUDA, backward_distwar.cu:145-275
    const V3 mean = mean_pre;
    float cov3[6];
#pragma unroll
    for (int k = 0; k < 6; k++) cov3[k] = cov3_pre[k];
    const float h_y = a.H / (2.0f * a.tan_fovy), h_x = a.W / (2.0f * a.tan_fovx);
    Cov2D c;
    cov2d(mean, h_x, h_y, a.tan_fovx, a.tan_fovy, cov3, Vm_pre, c);
    const float x_grad_mul = (c.txtz < -c.limx || c.txtz > c.limx) ? 0.f : 1.f;
    const float y_grad_mul = (c.tytz < -c.limy || c.tytz > c.limy) ? 0.f : 1.f;
    // cov2D = [[A, B], [B, Cc]] = (t0 V t0, t0 V t1; ., t1 V t1) with V = Vrk (symmetric) and t0, t1 the two vectors
    // T_(0, .), T_(1, .) of the projection T = W J.  Everything below is that bilinear form differentiated:
    //   dL/dcov2D from dL/dconic (conic = cov2D^-1; the reference's regularised 1 / (det^2 + 1e-7), :206-214)
    //   dL/dV    = da t0 t0^T + db/2 (t0 t1^T + t1 t0^T) + dc t1 t1^T      (off-diagonal entries count twice, :216-232)
    //   dL/dt0   = 2 da V t0 + db V t1,   dL/dt1 = 2 dc V t1 + db V t0     (:237-249; V t0 and V t1 evaluated once each)
    //   dL/dJ    = rows of W against those two vectors                      (:251-255)
    const float A = c.a, B = c.b, Cc = c.c;
    const float denom = A * Cc - B * B;
    const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    const float t0[3] = {c.T.m[0][0], c.T.m[0][1], c.T.m[0][2]}, t1[3] = {c.T.m[1][0], c.T.m[1][1], c.T.m[1][2]};
    float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
    if (denom2inv != 0) {
      dL_da = denom2inv * (-Cc * Cc * dcx + 2 * B * Cc * dcy + (denom - A * Cc) * dcw);
      dL_dc = denom2inv * (-A * A * dcw + 2 * A * B * dcy + (denom - A * Cc) * dcx);
      dL_db = denom2inv * 2 * (B * Cc * dcx - (denom + 2 * B * B) * dcy + A * B * dcw);
      constexpr int TRI[6][2] = {{0, 0}, {0, 1}, {0, 2}, {1, 1}, {1, 2}, {2, 2}};   // storage order of the symmetric 3x3
#pragma unroll
      for (int k = 0; k < 6; k++) {
        const int i = TRI[k][0], j = TRI[k][1];
        const float sym = i == j ? 1.f : 2.f;
        dcov[k] = sym * t0[i] * t0[j] * dL_da + (i == j ? t0[i] * t1[i] : t0[i] * t1[j] + t0[j] * t1[i]) * dL_db +
                  sym * t1[i] * t1[j] * dL_dc;
      }
    }
    float Vt0[3], Vt1[3], g0[3], g1[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
      Vt0[j] = t0[0] * c.Vrk.m[j][0] + t0[1] * c.Vrk.m[j][1] + t0[2] * c.Vrk.m[j][2];
      Vt1[j] = t1[0] * c.Vrk.m[j][0] + t1[1] * c.Vrk.m[j][1] + t1[2] * c.Vrk.m[j][2];
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
      g0[j] = 2 * Vt0[j] * dL_da + Vt1[j] * dL_db;
      g1[j] = 2 * Vt1[j] * dL_dc + Vt0[j] * dL_db;
    }
    auto w_row = [&](int r, const float (&g)[3]) { return c.W.m[r][0] * g[0] + c.W.m[r][1] * g[1] + c.W.m[r][2] * g[2]; };
    const float dL_dJ00 = w_row(0, g0), dL_dJ02 = w_row(2, g0), dL_dJ11 = w_row(1, g1), dL_dJ12 = w_row(2, g1);
    const float tz = 1.f / c.t.z, tz2 = tz * tz, tz3 = tz2 * tz;
    const float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
    const float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
    const float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * c.t.x) * tz3 * dL_dJ02 +
                         (2 * h_y * c.t.y) * tz3 * dL_dJ12;
    const float* Vm = Vm_pre;
    dmean[0] = Vm[0] * dL_dtx + Vm[1] * dL_dty + Vm[2] * dL_dtz;  // transformVec4x3Transpose, auxiliary.h:89-97
    dmean[1] = Vm[4] * dL_dtx + Vm[5] * dL_dty + Vm[6] * dL_dtz;
    dmean[2] = Vm[8] * dL_dtx + Vm[9] * dL_dty + Vm[10] * dL_dtz;

    // ---- preprocessCUDA (backward), backward_distwar.cu:347-397
    const float* pj = pj_pre;
    const float m_hw = pj[3] * mean.x + pj[7] * mean.y + pj[11] * mean.z + pj[15];
    const float m_w = 1.0f / (m_hw + 0.0000001f);
    const float mul1 = (pj[0] * mean.x + pj[4] * mean.y + pj[8] * mean.z + pj[12]) * m_w * m_w;
    const float mul2 = (pj[1] * mean.x + pj[5] * mean.y + pj[9] * mean.z + pj[13]) * m_w * m_w;
    dmean[0] += (pj[0] * m_w - pj[3] * mul1) * dmx + (pj[1] * m_w - pj[3] * mul2) * dmy;
    dmean[1] += (pj[4] * m_w - pj[7] * mul1) * dmx + (pj[5] * m_w - pj[7] * mul2) * dmy;
    dmean[2] += (pj[8] * m_w - pj[11] * mul1) * dmx + (pj[9] * m_w - pj[11] * mul2) * dmy;

    if (a.shs) {
      // computeColorFromSH (backward), backward_distwar.cu:21-140
      const V3 dir_orig = {mean.x - cam_pre[0], mean.y - cam_pre[1], mean.z - cam_pre[2]};
      const float len = sqrtf(dir_orig.x * dir_orig.x + dir_orig.y * dir_orig.y + dir_orig.z * dir_orig.z);
      const float x = dir_orig.x / len, y = dir_orig.y / len, z = dir_orig.z / len;
      const float* sh = a.shs + (size_t)idx * M * 3;
      float* dsh = a.dL_dsh + (size_t)idx * M * 3;
      float dRGB[3];
#pragma unroll
      for (int ch = 0; ch < 3; ch++) dRGB[ch] = dcol[ch] * (clamped_pre[ch] ? 0.f : 1.f);
      float ddx[3] = {0, 0, 0}, ddy[3] = {0, 0, 0}, ddz[3] = {0, 0, 0};
#define SHC(k, ch) sh[3 * (k) + (ch)]
#define DSH(k, coef) { const float cf_ = (coef); dsh[3 * (k)] = cf_ * dRGB[0]; dsh[3 * (k) + 1] = cf_ * dRGB[1]; dsh[3 * (k) + 2] = cf_ * dRGB[2]; }
      DSH(0, kSH_C0);
#pragma unroll
      for (int ch = 0; ch < 3; ch++) dsh_dc[ch] = kSH_C0 * dRGB[ch];
      if (D > 0) {
        DSH(1, -kSH_C1 * y); DSH(2, kSH_C1 * z); DSH(3, -kSH_C1 * x);
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
          ddx[ch] = -kSH_C1 * SHC(3, ch);
          ddy[ch] = -kSH_C1 * SHC(1, ch);
          ddz[ch] = kSH_C1 * SHC(2, ch);
        }
        if (D > 1) {
          const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
          DSH(4, kSH_C2[0] * xy); DSH(5, kSH_C2[1] * yz); DSH(6, kSH_C2[2] * (2.f * zz - xx - yy));
          DSH(7, kSH_C2[3] * xz); DSH(8, kSH_C2[4] * (xx - yy));
#pragma unroll
          for (int ch = 0; ch < 3; ch++) {
            ddx[ch] += kSH_C2[0] * y * SHC(4, ch) + kSH_C2[2] * 2.f * -x * SHC(6, ch) + kSH_C2[3] * z * SHC(7, ch) +
                       kSH_C2[4] * 2.f * x * SHC(8, ch);
            ddy[ch] += kSH_C2[0] * x * SHC(4, ch) + kSH_C2[1] * z * SHC(5, ch) + kSH_C2[2] * 2.f * -y * SHC(6, ch) +
                       kSH_C2[4] * 2.f * -y * SHC(8, ch);
            ddz[ch] += kSH_C2[1] * y * SHC(5, ch) + kSH_C2[2] * 2.f * 2.f * z * SHC(6, ch) + kSH_C2[3] * x * SHC(7, ch);
          }
          if (D > 2) {
            DSH(9, kSH_C3[0] * y * (3.f * xx - yy)); DSH(10, kSH_C3[1] * xy * z);
            DSH(11, kSH_C3[2] * y * (4.f * zz - xx - yy)); DSH(12, kSH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy));
            DSH(13, kSH_C3[4] * x * (4.f * zz - xx - yy)); DSH(14, kSH_C3[5] * z * (xx - yy));
            DSH(15, kSH_C3[6] * x * (xx - 3.f * yy));
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
              ddx[ch] += (kSH_C3[0] * SHC(9, ch) * 3.f * 2.f * xy + kSH_C3[1] * SHC(10, ch) * yz +
                          kSH_C3[2] * SHC(11, ch) * -2.f * xy + kSH_C3[3] * SHC(12, ch) * -3.f * 2.f * xz +
                          kSH_C3[4] * SHC(13, ch) * (-3.f * xx + 4.f * zz - yy) + kSH_C3[5] * SHC(14, ch) * 2.f * xz +
                          kSH_C3[6] * SHC(15, ch) * 3.f * (xx - yy));
              ddy[ch] += (kSH_C3[0] * SHC(9, ch) * 3.f * (xx - yy) + kSH_C3[1] * SHC(10, ch) * xz +
                          kSH_C3[2] * SHC(11, ch) * (-3.f * yy + 4.f * zz - xx) + kSH_C3[3] * SHC(12, ch) * -3.f * 2.f * yz +
                          kSH_C3[4] * SHC(13, ch) * -2.f * xy + kSH_C3[5] * SHC(14, ch) * -2.f * yz +
                          kSH_C3[6] * SHC(15, ch) * -3.f * 2.f * xy);
              ddz[ch] += (kSH_C3[1] * SHC(10, ch) * xy + kSH_C3[2] * SHC(11, ch) * 4.f * 2.f * yz +
                          kSH_C3[3] * SHC(12, ch) * 3.f * (2.f * zz - xx - yy) + kSH_C3[4] * SHC(13, ch) * 4.f * 2.f * xz +
                          kSH_C3[5] * SHC(14, ch) * (xx - yy));
            }
          }
        }
      }
      // coefficients above the active degree receive zero gradient
      for (int k = (D + 1) * (D + 1); k < M; k++) { dsh[3 * k] = 0.f; dsh[3 * k + 1] = 0.f; dsh[3 * k + 2] = 0.f; }
#undef SHC
#undef DSH
      const V3 dL_ddir = {ddx[0] * dRGB[0] + ddx[1] * dRGB[1] + ddx[2] * dRGB[2],
                          ddy[0] * dRGB[0] + ddy[1] * dRGB[1] + ddy[2] * dRGB[2],
                          ddz[0] * dRGB[0] + ddz[1] * dRGB[1] + ddz[2] * dRGB[2]};
      const V3 dm = dnormvdv(dir_orig, dL_ddir);
      dmean[0] += dm.x; dmean[1] += dm.y; dmean[2] += dm.z;
    }
    if (a.scales) {
      // computeCov3D (backward), backward_distwar.cu:279-342
      const float4 q = q_pre;
      const float r = q.x, x = q.y, y = q.z, z = q.w;
      const M3 R = quat_R(r, x, y, z);
      const float s[3] = {a.scale_modifier * s_pre[0], a.scale_modifier * s_pre[1], a.scale_modifier * s_pre[2]};
      M3 M2;  // 2 * (S * R)
#pragma unroll
      for (int cc = 0; cc < 3; cc++)
#pragma unroll
        for (int w = 0; w < 3; w++) M2.m[cc][w] = (s[w] * R.m[cc][w]) * 2.0f;
      M3 dSig;
      dSig.m[0][0] = dcov[0]; dSig.m[0][1] = 0.5f * dcov[1]; dSig.m[0][2] = 0.5f * dcov[2];
      dSig.m[1][0] = 0.5f * dcov[1]; dSig.m[1][1] = dcov[3]; dSig.m[1][2] = 0.5f * dcov[4];
      dSig.m[2][0] = 0.5f * dcov[2]; dSig.m[2][1] = 0.5f * dcov[4]; dSig.m[2][2] = dcov[5];
      const M3 dM = m3mul(M2, dSig);
      const M3 Rt = m3tr(R);
      M3 dMt = m3tr(dM);
#pragma unroll
      for (int k = 0; k < 3; k++)
        dscale[k] = Rt.m[k][0] * dMt.m[k][0] + Rt.m[k][1] * dMt.m[k][1] + Rt.m[k][2] * dMt.m[k][2];
#pragma unroll
      for (int k = 0; k < 3; k++)
#pragma unroll
        for (int w = 0; w < 3; w++) dMt.m[k][w] *= s[k];
#define D_(i, j) dMt.m[i][j]
      drot[0] = 2 * z * (D_(0, 1) - D_(1, 0)) + 2 * y * (D_(2, 0) - D_(0, 2)) + 2 * x * (D_(1, 2) - D_(2, 1));
      drot[1] = 2 * y * (D_(1, 0) + D_(0, 1)) + 2 * z * (D_(2, 0) + D_(0, 2)) + 2 * r * (D_(1, 2) - D_(2, 1)) -
                4 * x * (D_(2, 2) + D_(1, 1));
      drot[2] = 2 * x * (D_(1, 0) + D_(0, 1)) + 2 * r * (D_(2, 0) - D_(0, 2)) + 2 * z * (D_(1, 2) + D_(2, 1)) -
                4 * y * (D_(2, 2) + D_(0, 0));
      drot[3] = 2 * r * (D_(0, 1) - D_(1, 0)) + 2 * x * (D_(2, 0) + D_(0, 2)) + 2 * y * (D_(1, 2) + D_(2, 1)) -
                4 * z * (D_(1, 1) + D_(0, 0));
#undef D_
    }
  } else if (a.shs) {
    float* dsh = a.dL_dsh + (size_t)idx * M * 3;
    for (int k = 0; k < 3 * M; k++) dsh[k] = 0.f;
  }
  if (!live) return;
  // every output is written (zeros for culled Gaussians): no separate zero-fill pass
  // returned screen-space gradient: in the single-pass mode the RGB channels' share only (what the reference's
  // densification statistics read from the RGB pass); dL_dmeans3D above used the total
  const float gm2x = a.n_extra ? dmx_rgb : dmx, gm2y = a.n_extra ? dmy_rgb : dmy;
  if (MODE == 0) {
    a.dL_dmeans2D[3 * idx] = gm2x;
    a.dL_dmeans2D[3 * idx + 1] = gm2y;
    a.dL_dmeans2D[3 * idx + 2] = 0.f;
    if (a.n_extra) ((float4*)a.dL_dextra)[idx] = make_float4(dex[0], dex[1], dex[2], dex[3]);
    ((float4*)a.dL_dconic)[idx] = make_float4(dcx, dcy, 0.f, dcw);
    a.dL_dopacity[idx] = dop;
    a.dL_dcolors[3 * idx] = dcol[0]; a.dL_dcolors[3 * idx + 1] = dcol[1]; a.dL_dcolors[3 * idx + 2] = dcol[2];
    a.dL_dmeans3D[3 * idx] = dmean[0]; a.dL_dmeans3D[3 * idx + 1] = dmean[1]; a.dL_dmeans3D[3 * idx + 2] = dmean[2];
#pragma unroll
    for (int k = 0; k < 6; k++) a.dL_dcov3D[6 * (size_t)idx + k] = dcov[k];
    a.dL_dscales[3 * idx] = dscale[0]; a.dL_dscales[3 * idx + 1] = dscale[1]; a.dL_dscales[3 * idx + 2] = dscale[2];
    ((float4*)a.dL_drotations)[idx] = make_float4(drot[0], drot[1], drot[2], drot[3]);
    return;
  }
  // ---- MODE 1 / 2: the parameters' backward of this Gaussian, from registers (see the kernel's header)
  if (a.dL_dmeans2D) { a.dL_dmeans2D[3 * idx] = gm2x; a.dL_dmeans2D[3 * idx + 1] = gm2y; a.dL_dmeans2D[3 * idx + 2] = 0.f; }
  if (pb.max_radii2D) hgs_densify_stats_lane(idx, radius_pre, gm2x, gm2y, pb.max_radii2D, pb.grad_accum, pb.denom);
  // (the forward's opacity, conic_opacity.w, exists for visible Gaussians only; an invisible one has dop = 0)
  const float o_act = vis ? co_pre.w : 0.f;
  if (MODE == 1) {
    HgsSegGradVals gv;
    gv.gx[0] = dmean[0]; gv.gx[1] = dmean[1]; gv.gx[2] = dmean[2];
    gv.gs0 = dscale[0];
    gv.gq = make_float4(drot[0], drot[1], drot[2], drot[3]);
    gv.gd[0] = gv.gd[1] = gv.gd[2] = 0.f;
    gv.ge = make_float4(dex[0], dex[1], dex[2], dex[3]);
    gv.has_quat = true; gv.has_extra = true; gv.has_scale = true;
    float h[3], gD[3];
    hgs_segment_endpoint_grads_v(seg_pre, pb.dist_to_scale_factor, gv, h, gD);
    float4* sc = (float4*)pb.seg_contrib + 2 * (size_t)idx;
    sc[0] = make_float4(h[0] - gD[0], h[1] - gD[1], h[2] - gD[2], 0.f);
    sc[1] = make_float4(h[0] + gD[0], h[1] + gD[1], h[2] + gD[2], 0.f);
    const float g_w = (dscale[1] + dscale[2]) * s_pre[1];            // scale.y = exp(width): the forward's own value
    const float g_o = dop * o_act * (1.f - o_act);                   // sigmoid'
    const float g_m = dex[0] * mask_pre * (1.f - mask_pre);
    pb.d_width[idx] = g_w;
    pb.d_opacity_raw[idx] = g_o;
    pb.d_mask_raw[idx] = g_m;
    // Adam in the lane (include/hgs.h HgsAdamSlot): every gradient of this Gaussian's own parameters is final here, and nothing
    // else in the launch reads the raw parameters (the lane itself works on the forward's activations)
    hgs_adam_lane<1>(pb.adam.slot[0], (size_t)idx, &g_w, pb.adam.beta1, pb.adam.beta2, pb.adam.eps);
    hgs_adam_lane<1>(pb.adam.slot[1], (size_t)idx, &g_o, pb.adam.beta1, pb.adam.beta2, pb.adam.eps);
    hgs_adam_lane<1>(pb.adam.slot[2], (size_t)idx, &g_m, pb.adam.beta1, pb.adam.beta2, pb.adam.eps);
    hgs_adam_lane<3>(pb.adam.slot[3], (size_t)idx, dsh_dc, pb.adam.beta1, pb.adam.beta2, pb.adam.eps);
  } else {
    const HgsCloudParamGrads cg = hgs_cloud_param_grads(s_pre[0], s_pre[1], s_pre[2], raw_rot_pre, o_act, mask_pre, dscale,
                                                        make_float4(drot[0], drot[1], drot[2], drot[3]), dop,
                                                        make_float4(dex[0], dex[1], dex[2], dex[3]));
    pb.d_means3D[3 * idx] = dmean[0]; pb.d_means3D[3 * idx + 1] = dmean[1]; pb.d_means3D[3 * idx + 2] = dmean[2];
    pb.d_scaling_raw[3 * idx] = cg.d_s[0]; pb.d_scaling_raw[3 * idx + 1] = cg.d_s[1]; pb.d_scaling_raw[3 * idx + 2] = cg.d_s[2];
    ((float4*)pb.d_rotation_raw)[idx] = cg.d_r;
    pb.d_opacity_raw[idx] = cg.d_o;
    pb.d_mask_raw[idx] = cg.d_m;
    const float g_r[4] = {cg.d_r.x, cg.d_r.y, cg.d_r.z, cg.d_r.w};
    hgs_adam_lane<3>(pb.adam.slot[0], (size_t)idx, dmean, pb.adam.beta1, pb.adam.beta2, pb.adam.eps);
    hgs_adam_lane<3>(pb.adam.slot[1], (size_t)idx, cg.d_s, pb.adam.beta1, pb.adam.beta2, pb.adam.eps);
    hgs_adam_lane<4>(pb.adam.slot[2], (size_t)idx, g_r, pb.adam.beta1, pb.adam.beta2, pb.adam.eps);
    hgs_adam_lane<1>(pb.adam.slot[3], (size_t)idx, &cg.d_o, pb.adam.beta1, pb.adam.beta2, pb.adam.eps);
    hgs_adam_lane<1>(pb.adam.slot[4], (size_t)idx, &cg.d_m, pb.adam.beta1, pb.adam.beta2, pb.adam.eps);
    hgs_adam_lane<3>(pb.adam.slot[5], (size_t)idx, dsh_dc, pb.adam.beta1, pb.adam.beta2, pb.adam.eps);
  }
}

__global__ __launch_bounds__(HGS_BLOCK) void mark_visible_kernel(int P, const float* means3D, const float* V, uint8_t* present) {
  const int idx = blockIdx.x * HGS_BLOCK + threadIdx.x;
  if (idx >= P) return;
  const V3 p = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
  present[idx] = !(xform4x3(p, V).z <= 0.2f);
}

}  // namespace

int hgs_launch_preprocess_fwd(hipStream_t s, const HgsFwdArgs& a, const HgsGeom& g, const HgsImage& im, int* radii) {
  const int nblk = (a.P + HGS_BLOCK - 1) / HGS_BLOCK;
  {
    HgsProfScope _prof(s, HGS_K_PREPROCESS_FWD);
    hipLaunchKernelGGL(preprocess_fwd_kernel, dim3(nblk), dim3(HGS_BLOCK), 0, s, a, g, im, radii);
  }
  HGS_CHECK_LAUNCH();
  return 0;
}
bool hgs_preprocess_prologue_kernel(const void* func, int* n_params) {
  if (func == (const void*)hair_preprocess_fwd_kernel) { *n_params = 7; return true; }
  if (func == (const void*)cloud_preprocess_fwd_kernel) { *n_params = 6; return true; }
  return false;
}
int hgs_launch_hair_preprocess_fwd(hipStream_t s, const HgsFwdArgs& a, const HgsGeom& g, const HgsImage& im, int* radii,
                                   const float* endpoints, const long long* pairs, const float* width, float f,
                                   const float* opacity_raw, const float* mask_raw, float* xyz, float* scale, float* quat,
                                   float* opacity, float* extra4, const HgsStrandFusion& fusion) {
  HgsStrandFusion fu = fusion;
  const HgsPrologue pro = fu.prologue;
  fu.prologue = HgsPrologue{};      // (handed over as the kernel's last argument)
  const HgsParamSrc st = {endpoints, pairs, width, f, nullptr, nullptr, opacity_raw, mask_raw, xyz, scale, quat, opacity, extra4};
  const unsigned nblk = (unsigned)((a.P + HGS_BLOCK - 1) / HGS_BLOCK) + (unsigned)((fu.n_smooth + 255) / 256) +
                        (pro.table ? hgs_prologue_blocks(pro.zero_bytes / 4) : 0u);
  {
    HgsProfScope _prof(s, HGS_K_PREPROCESS_FWD);
    hipLaunchKernelGGL(hair_preprocess_fwd_kernel, dim3(nblk), dim3(HGS_BLOCK), 0, s, a, g, im, radii, st, fu, pro);
  }
  HGS_CHECK_LAUNCH();
  return 0;
}
int hgs_launch_cloud_preprocess_fwd(hipStream_t s, const HgsFwdArgs& a, const HgsGeom& g, const HgsImage& im, int* radii,
                                    const float* scaling_raw, const float* rotation_raw, const float* opacity_raw,
                                    const float* mask_raw, float* scale, float* quat, float* opacity, float* extra4,
                                    const HgsPrologue& pro) {
  const HgsParamSrc st = {nullptr, nullptr, nullptr, 0.f, scaling_raw, rotation_raw, opacity_raw, mask_raw, nullptr, scale, quat,
                          opacity, extra4};
  const unsigned nblk = (unsigned)((a.P + HGS_BLOCK - 1) / HGS_BLOCK) + (pro.table ? hgs_prologue_blocks(pro.zero_bytes / 4) : 0u);
  {
    HgsProfScope _prof(s, HGS_K_PREPROCESS_FWD);
    hipLaunchKernelGGL(cloud_preprocess_fwd_kernel, dim3(nblk), dim3(HGS_BLOCK), 0, s, a, g, im, radii, st, pro);
  }
  HGS_CHECK_LAUNCH();
  return 0;
}
int hgs_launch_scatter(hipStream_t s, int P, int W, int H, int Rcap, const float* features, const float* extra, int n_extra,
                       const HgsGeom& g, const HgsImage& im, const HgsBinning& b) {
  const int nblk = (P + HGS_BLOCK - 1) / HGS_BLOCK;
  const int gx = (W + HGS_TILE - 1) / HGS_TILE, T = gx * ((H + HGS_TILE - 1) / HGS_TILE);
  // Whether the fused-scan mode is on is a device-side fact (status words written by the preprocess kernel), so the scan
  // workgroup and its LDS table are provided whenever the mode is POSSIBLE (hgs_forward_preprocess: T and P within the
  // limits).  The table is dynamic LDS, i.e. 33 KB at 1080p for EVERY workgroup of the launch (three resident workgroups
  // per CU instead of thirteen): for the big launches that costs about what the scan kernel did (1 M Gaussians: scatter
  // 59 + scan 20 us against 81 us).  The scan workgroup leaves at once when a scan kernel ran (blocking mode).
  const size_t lds = 0;   // (no dynamic LDS: see the scan workgroups)
  const bool can_fuse = T <= HGS_FUSED_SCAN_MAX_T && P <= HGS_FUSED_SCAN_MAX_P;
  {
    HgsProfScope _prof(s, HGS_K_SCATTER);
    const int scan_wg = can_fuse ? (T + HGS_BLOCK - 1) / HGS_BLOCK : 0;     // one tile per thread (<= HGS_SCAN_WGS workgroups)
    // helper workgroups for the further parts of blocks with more than HGS_SC_PART instances (see the kernel): at most one per
    // HGS_SC_PART instances of the capacity; none where the pass cannot hold such a block's worth per block on average
    const long long cap = Rcap > 0 ? Rcap : 0;
    const int helpers = cap >= 8ll * P ? (int)((cap + HGS_SC_PART - 1) / HGS_SC_PART) : 0;
    hipLaunchKernelGGL(scatter_kernel, dim3(nblk + scan_wg + helpers), dim3(HGS_BLOCK), lds, s, P, gx, T, (uint32_t)Rcap, features, n_extra ? extra : nullptr, g, im, b, scan_wg);
  }
  HGS_CHECK_LAUNCH();
  return 0;
}
int hgs_launch_row_reduce(hipStream_t s, int P, int Rcap, const float* inst_grad, const uint32_t* status, float* row_sums,
                          float* row_partials) {
  const unsigned runs = ((unsigned)Rcap + HGS_RR_RPW - 1) / HGS_RR_RPW;
  const unsigned nblk = (runs + HGS_BLOCK / 64 - 1) / (HGS_BLOCK / 64);
  {
    HgsProfScope _prof(s, HGS_K_MISC);
    hipLaunchKernelGGL(row_reduce_kernel, dim3(nblk), dim3(HGS_BLOCK), 0, s, (const float4*)inst_grad, status, (uint32_t)Rcap,
                       (float4*)row_sums, (float4*)row_partials, (uint32_t)P);
  }
  HGS_CHECK_LAUNCH();
  return 0;
}
int hgs_launch_preprocess_bwd(hipStream_t s, const HgsBwdArgs& a, const HgsGeom& g, const HgsBinning& b,
                              const float* inst_grad, int Rcap, const uint32_t* status, const HgsParamBackward* pb) {
  const int mode = pb ? pb->kind : 0;
  const HgsParamBackward p = pb ? *pb : HgsParamBackward{};
  // (the loss head's deferred tail, MODE 1 / 2: one spare workgroup behind the launch's own)
  const int nblk = (a.P + HGS_BLOCK - 1) / HGS_BLOCK + ((mode != 0 && p.head_tail.out) ? 1 : 0);
  const bool dc = !a.shs || (a.D == 0 && a.M == 1);
  {
    HgsProfScope _prof(s, HGS_K_PREPROCESS_BWD);
#define HGS_PPB_LAUNCH(DC, MODE) hipLaunchKernelGGL((preprocess_bwd_kernel<DC, MODE>), dim3(nblk), dim3(HGS_BLOCK), 0, s, a, g, b, inst_grad, (uint32_t)Rcap, status, p)
    if (mode == 0) { if (dc) HGS_PPB_LAUNCH(true, 0); else HGS_PPB_LAUNCH(false, 0); }
    else if (mode == HGS_PARAMS_HAIR) { if (dc) HGS_PPB_LAUNCH(true, 1); else HGS_PPB_LAUNCH(false, 1); }
    else { if (dc) HGS_PPB_LAUNCH(true, 2); else HGS_PPB_LAUNCH(false, 2); }
#undef HGS_PPB_LAUNCH
  }
  HGS_CHECK_LAUNCH();
  return 0;
}
int hgs_launch_mark_visible(hipStream_t s, int P, const float* means3D, const float* viewmatrix, uint8_t* present) {
  const int nblk = (P + HGS_BLOCK - 1) / HGS_BLOCK;
  hipLaunchKernelGGL(mark_visible_kernel, dim3(nblk), dim3(HGS_BLOCK), 0, s, P, means3D, viewmatrix, present);
  HGS_CHECK_LAUNCH();
  return 0;
}

#if HGS_PPF_TRACE
extern "C" int hgs_debug_ppf_trace(unsigned long long* host_out, int n_wg) {
  if (n_wg > PPF_TRACE_MAX) n_wg = PPF_TRACE_MAX;
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_ppf_trace), (size_t)n_wg * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif
#if HGS_SCATTER_TRACE
extern "C" int hgs_debug_scatter_trace(unsigned long long* host_out, int n_wg) {
  if (n_wg > SC_TRACE_MAX) n_wg = SC_TRACE_MAX;
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_sc_trace), (size_t)n_wg * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif

