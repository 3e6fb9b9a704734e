// glm_probe.cpp -- test infrastructure: calls the reference's VENDORED glm
// (/root/reference/submodules/diff-gaussian-rasterization/third_party/glm, included where it lies by
// oracle/build_ref.py; nothing of it is copied) so that the operand order of the oracle's mat3_mul / mat3_tr
// (raster_oracle.c, claim: glm/detail/type_mat3x3.inl operator* and transpose) can be pinned bit for bit.
// The three entry points evaluate the glm expressions the rasterizer uses:
//   glm_probe_mul        A * B                          (CR/forward.cu:101 "T = W * J", :140 "M = S * R")
//   glm_probe_transpose  glm::transpose(A)
//   glm_probe_gram       glm::transpose(M) * M          (CR/forward.cu:143, Sigma of computeCov3D)
//   glm_probe_sandwich   transpose(T) * transpose(V) * T (CR/forward.cu:108, cov of computeCov2D)
// Matrices cross the C boundary as 9 floats in glm's own memory order (column-major: m[col][row]).
// Built with -O0 -ffp-contract=off: every product and sum is rounded on its own, like the oracle and the
// strict HIP translation units.
#include <glm/glm.hpp>

static glm::mat3 load(const float* p) { return glm::mat3(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8]); }
static void store(const glm::mat3& m, float* p) {
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) p[3 * c + r] = m[c][r];
}

extern "C" {
void glm_probe_mul(const float* a, const float* b, float* out) { store(load(a) * load(b), out); }
void glm_probe_transpose(const float* a, float* out) { store(glm::transpose(load(a)), out); }
void glm_probe_gram(const float* m, float* out) { glm::mat3 M = load(m); store(glm::transpose(M) * M, out); }
void glm_probe_sandwich(const float* t, const float* v, float* out) {
  glm::mat3 T = load(t), V = load(v);
  store(glm::transpose(T) * glm::transpose(V) * T, out);
}
}
