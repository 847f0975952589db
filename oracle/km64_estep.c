/* TEST INFRASTRUCTURE (oracle/) -- E-step of the KM64 canonical k-means arithmetic in plain C.
 *
 * sklearn 1.7.2's Lloyd E-step (cluster/_k_means_lloyd.pyx:_update_chunk_dense) evaluates, per (sample i,
 * centre j):  pairwise[i][j] = ||c_j||^2 + (-2) * <x_i, c_j>  with the dot product coming out of OpenBLAS
 * dgemm (an FMA chain over k = 0,1,2 starting from the rounded product x0*c0) and ||c_j||^2 out of
 * numpy's einsum (SSE2 two-lane accumulation: (c0^2 + c2^2) + c1^2), then takes the FIRST arg-min.  On
 * integer colour lattices exact ties in real arithmetic are common, so these roundings decide labels; this
 * file restates them with libm's correctly rounded fma().  Build: gcc -O2 -ffp-contract=off -shared -fPIC.
 */
#include <math.h>
#include <stdint.h>

/* X: n x 3, C: k x 3 (row major doubles); labels_out: n int32; mind_out (may be NULL): n doubles */
void km64_estep(const double* X, int64_t n, const double* C, int64_t k, int32_t* labels_out, double* mind_out) {
  for (int64_t i = 0; i < n; ++i) {
    const double x0 = X[3 * i], x1 = X[3 * i + 1], x2 = X[3 * i + 2];
    double best = 0.0;
    int32_t bj = 0;
    for (int64_t j = 0; j < k; ++j) {
      const double c0 = C[3 * j], c1 = C[3 * j + 1], c2 = C[3 * j + 2];
      const double csq = (c0 * c0 + c2 * c2) + c1 * c1;
      const double dot = fma(x2, c2, fma(x1, c1, x0 * c0));
      const double d = csq + (-2.0 * dot);
      if (j == 0 || d < best) { best = d; bj = (int32_t)j; }
    }
    labels_out[i] = bj;
    if (mind_out) mind_out[i] = best;
  }
}

/* ||c||^2 in numpy-einsum order, for callers that need the same value */
void km64_csq(const double* C, int64_t k, double* out) {
  for (int64_t j = 0; j < k; ++j) {
    const double c0 = C[3 * j], c1 = C[3 * j + 1], c2 = C[3 * j + 2];
    out[j] = (c0 * c0 + c2 * c2) + c1 * c1;
  }
}
