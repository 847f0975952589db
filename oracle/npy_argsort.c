/* TEST INFRASTRUCTURE (oracle/) -- numpy's SCALAR argsort for float64, restated.
 *
 * Reference call site: encoder/compression/clustering.py:211-218 -> sklearn/cluster/_kmeans.py::_mini_batch_step:
 *     indices_dont_reassign = np.argsort(weight_sums)[int(0.5 * X.shape[0]):]
 * np.argsort's default kind ('quicksort') is an UNSTABLE sort, and weight_sums is massively tied (most centres of a large
 * MiniBatchKMeans hold 0, 1, 2 ... samples), so WHICH of the tied centres land in the first half-batch depends on the sort's
 * inner workings.  numpy 2.2.6 (the version in the build container; unpinned by the reference, requirements.txt:2) has three:
 * x86-simd-sort's AVX-512 and AVX2 argsort kernels and, when neither is usable
 *     NPY_DISABLE_CPU_FEATURES="AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR AVX2 FMA3"
 * (or on a CPU without AVX2, or any non-x86 host), the plain C++ template this file restates:
 *     numpy/_core/src/npysort/quicksort.cpp  aquicksort_<npy::double_tag>   introsort on the index array: median of 3,
 *         Hoare partition with the pivot parked at pr - 1, the larger part pushed on a stack, insertion sort once a part
 *         has pr - pl <= 15 (measured on numpy 2.2.6: a part with pr - pl = 16 is still partitioned), depth limit
 *         2 * floor(log2(n)) and then
 *     numpy/_core/src/npysort/heapsort.cpp   aheapsort_<npy::double_tag>.
 * That setting is the HOST SETTING OF RECORD of this build for the one unportable step of the path:
 * the scalar kernel is deterministic C, runs on every host numpy supports, and the restatement is checked against
 * numpy itself under that setting (tests/golden/make_golden_npysort.py -> g15_npysort.npz: 0 mismatches) and, through the
 * whole MiniBatchKMeans fit, against scikit-learn's UNTOUCHED fit_predict (g11 "scalar" records).
 * Comparison: npy::double_tag::less(a, b) = a < b || (b != b && a == a) (NaNs last); the weights are never NaN.
 */
#include <stdint.h>

#define NPY_SMALL_QUICKSORT 15
#define NPY_QS_STACK 128

static inline int dless(double a, double b) { return a < b || (b != b && a == a); }

static int msb64(uint64_t n) {
  int d = 0;
  while (n >>= 1) ++d;
  return d;
}

static void aheapsort_f64(const double* v, int64_t* tosort, int64_t n) {
  int64_t* a = tosort - 1; /* 1-based */
  int64_t i, j, l, tmp;
  for (l = n >> 1; l > 0; --l) {
    tmp = a[l];
    for (i = l, j = l << 1; j <= n;) {
      if (j < n && dless(v[a[j]], v[a[j + 1]])) j += 1;
      if (dless(v[tmp], v[a[j]])) { a[i] = a[j]; i = j; j += j; }
      else break;
    }
    a[i] = tmp;
  }
  for (; n > 1;) {
    tmp = a[n];
    a[n] = a[1];
    n -= 1;
    for (i = 1, j = 2; j <= n;) {
      if (j < n && dless(v[a[j]], v[a[j + 1]])) j++;
      if (dless(v[tmp], v[a[j]])) { a[i] = a[j]; i = j; j += j; }
      else break;
    }
    a[i] = tmp;
  }
}

#define ISWAP(a, b) do { const int64_t t_ = (a); (a) = (b); (b) = t_; } while (0)

/* tosort must hold 0 .. n-1 on entry (np.argsort initialises it so); sorted in place.  depth0 < 0: numpy's own depth limit
 * 2 * floor(log2(num)); tests of the HIP emulation pass small values to reach the heapsort branch with ordinary inputs. */
void npy_aquicksort_f64_depth(const double* v, int64_t* tosort, int64_t num, int depth0) {
  if (num <= 1) return;
  int64_t* pl = tosort;
  int64_t* pr = tosort + num - 1;
  int64_t* stack[NPY_QS_STACK];
  int64_t** sptr = stack;
  int depth[NPY_QS_STACK];
  int* psdepth = depth;
  int cdepth = depth0 >= 0 ? depth0 : msb64((uint64_t)num) * 2;
  int64_t *pm, *pi, *pj, *pk, vi;
  double vp;
  for (;;) {
    if (cdepth < 0) {
      aheapsort_f64(v, pl, pr - pl + 1);
      goto stack_pop;
    }
    while ((pr - pl) > NPY_SMALL_QUICKSORT) {
      pm = pl + ((pr - pl) >> 1);
      if (dless(v[*pm], v[*pl])) ISWAP(*pm, *pl);
      if (dless(v[*pr], v[*pm])) ISWAP(*pr, *pm);
      if (dless(v[*pm], v[*pl])) ISWAP(*pm, *pl);
      vp = v[*pm];
      pi = pl;
      pj = pr - 1;
      ISWAP(*pm, *pj);
      for (;;) {
        do { ++pi; } while (dless(v[*pi], vp));
        do { --pj; } while (dless(vp, v[*pj]));
        if (pi >= pj) break;
        ISWAP(*pi, *pj);
      }
      pk = pr - 1;
      ISWAP(*pi, *pk);
      if (pi - pl < pr - pi) { *sptr++ = pi + 1; *sptr++ = pr; pr = pi - 1; }
      else { *sptr++ = pl; *sptr++ = pi - 1; pl = pi + 1; }
      *psdepth++ = --cdepth;
    }
    for (pi = pl + 1; pi <= pr; ++pi) {
      vi = *pi;
      vp = v[vi];
      pj = pi;
      pk = pi - 1;
      while (pj > pl && dless(vp, v[*pk])) *pj-- = *pk--;
      *pj = vi;
    }
  stack_pop:
    if (sptr == stack) break;
    pr = *(--sptr);
    pl = *(--sptr);
    cdepth = *(--psdepth);
  }
}

void npy_aquicksort_f64(const double* v, int64_t* tosort, int64_t num) { npy_aquicksort_f64_depth(v, tosort, num, -1); }

/* np.argsort(v) under the scalar setting: fills order[0..n) */
void npy_argsort_f64(const double* v, int64_t n, int64_t* order) {
  for (int64_t i = 0; i < n; ++i) order[i] = i;
  npy_aquicksort_f64(v, order, n);
}

void npy_argsort_f64_depth(const double* v, int64_t n, int64_t* order, int depth0) {
  for (int64_t i = 0; i < n; ++i) order[i] = i;
  npy_aquicksort_f64_depth(v, order, n, depth0);
}
