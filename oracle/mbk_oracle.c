/* TEST INFRASTRUCTURE (oracle/) -- native restatement of the reference's MiniBatchKMeans branch.
 *
 * Reference call site: encoder/compression/clustering.py:207-218
 *     MiniBatchKMeans(n_clusters=k, batch_size=1000, random_state=42, n_init='auto').fit_predict(colours as float64)
 * The algorithm lives in scikit-learn (1.7.2 in the build container, unpinned by the reference's requirements.txt:6)
 * and numpy's legacy RandomState; this file restates both, operation for operation, on integer colours, exactly as
 * oracle/rhccq_oracle.py::minibatch_kmeans_labels does in numpy (the two are checked against each other and against
 * sklearn itself: tests/test_oracle_golden.py).  It exists because the numpy version needs minutes for k >= 10^4 and
 * because bench.py's cpu_baseline leg wants a CPU figure that is not interpreter bound (OpenMP over the host cores).
 *
 *   numpy/random/_mt19937.c, mtrand.pyx (legacy RandomState): init_genrand seeding, 32-bit outputs; randint = masked
 *     rejection, one word per attempt; random_sample = ((a >> 5) * 2^26 + (b >> 6)) / 2^53; choice(n, p) = searchsorted
 *     (cumsum(p) / cumsum(p)[-1], u, 'right'); choice(n, replace=False, size) = permutation(n)[:size], permutation =
 *     Fisher-Yates from the top with random_interval (masked rejection);
 *   sklearn/cluster/_kmeans.py: MiniBatchKMeans.fit, _init_centroids, _kmeans_plusplus, _mini_batch_step,
 *     _mini_batch_convergence; _k_means_minibatch.pyx: update_center_dense; _k_means_lloyd.pyx: _update_chunk_dense (E-step
 *     expression, see km64_estep.c); _k_means_common.pyx: _inertia_dense / _euclidean_dense_dense.
 *   np.argsort(weight_sums) in _mini_batch_step is unstable over tied counts: `argsort_kind` selects the tie order --
 *     1 = numpy's scalar aquicksort (npy_argsort.c: what numpy runs under the host setting of record, the default of the
 *     build), 0 = the stable order (weight, index) (rounds 1-3; kept for the G11 "stable" records).
 *
 * Build: gcc -O3 -mavx2 -mfma -ffp-contract=off -fopenmp -shared -fPIC (together with km64_estep.c; the compiler never
 * fuses on its own, the only fused operations are the explicit fma() calls of the E-step).  n_threads only changes how the
 * order-independent loops are shared out; every floating-point sum that sklearn evaluates in a fixed order keeps it.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- MT19937 as numpy's legacy RandomState(seed) consumes it ------------------------------------------------ */
typedef struct {
  uint32_t key[624];
  int pos;
  int64_t consumed; /* raw 32-bit words handed out so far (what the GPU path calls the cursor) */
} mt_t;

static void mt_seed(mt_t* s, uint32_t seed) {
  for (int i = 0; i < 624; ++i) {
    s->key[i] = seed;
    seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)i + 1u;
  }
  s->pos = 624;
  s->consumed = 0;
}

static void mt_gen(mt_t* s) {
  uint32_t* k = s->key;
  int i;
  for (i = 0; i < 624 - 397; ++i) {
    const uint32_t y = (k[i] & 0x80000000u) | (k[i + 1] & 0x7fffffffu);
    k[i] = k[i + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  }
  for (; i < 623; ++i) {
    const uint32_t y = (k[i] & 0x80000000u) | (k[i + 1] & 0x7fffffffu);
    k[i] = k[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  }
  {
    const uint32_t y = (k[623] & 0x80000000u) | (k[0] & 0x7fffffffu);
    k[623] = k[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  }
  s->pos = 0;
}

static uint32_t mt_u32(mt_t* s) {
  if (s->pos == 624) mt_gen(s);
  uint32_t y = s->key[s->pos++];
  s->consumed++;
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

static double mt_double(mt_t* s) {
  const uint32_t a = mt_u32(s) >> 5, b = mt_u32(s) >> 6;
  return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
}

static uint32_t mask_for(uint32_t v) { /* smallest 2^b - 1 >= v */
  v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16;
  return v;
}

/* randint(0, n, count): legacy masked rejection (_bounded_integers.pyx); n <= 2^32 - 1 */
static void mt_randint(mt_t* s, int64_t n, int64_t count, int64_t* out) {
  const uint32_t rng = (uint32_t)(n - 1);
  if (rng == 0) { for (int64_t i = 0; i < count; ++i) out[i] = 0; return; }
  const uint32_t mask = mask_for(rng);
  for (int64_t i = 0; i < count; ++i) {
    uint32_t v;
    while ((v = (mt_u32(s) & mask)) > rng) {}
    out[i] = (int64_t)v;
  }
}

/* choice(n, p = ones / n): cdf = cumsum(p); cdf /= cdf[-1]; searchsorted(cdf, random_sample(), 'right') */
static int64_t mt_choice_uniform(mt_t* s, int64_t n) {
  const double p = 1.0 / (double)n, u = mt_double(s);
  double* cdf = (double*)malloc(sizeof(double) * (size_t)n);
  double acc = 0.0;
  for (int64_t i = 0; i < n; ++i) { acc = acc + p; cdf[i] = acc; }
  const double last = cdf[n - 1];
  int64_t lo = 0, hi = n;
  while (lo < hi) { /* first index with cdf[i] / last > u */
    const int64_t mid = (lo + hi) >> 1;
    if (cdf[mid] / last <= u) lo = mid + 1; else hi = mid;
  }
  free(cdf);
  return lo < n ? lo : n - 1;
}

/* permutation(n)[:take] */
static void mt_permutation(mt_t* s, int n, int take, int32_t* out) {
  int32_t* x = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  for (int i = 0; i < n; ++i) x[i] = i;
  for (int i = n - 1; i >= 1; --i) {
    const uint32_t mask = mask_for((uint32_t)i);
    uint32_t j;
    while ((j = (mt_u32(s) & mask)) > (uint32_t)i) {}
    const int32_t t = x[j]; x[j] = x[i]; x[i] = t;
  }
  memcpy(out, x, sizeof(int32_t) * (size_t)take);
  free(x);
}

/* ---- k-means++ (_kmeans_plusplus) in exact integers ------------------------------------------------------------ */
static inline int64_t d2(const uint8_t* a, const uint8_t* b) {
  const int64_t x = (int64_t)a[0] - b[0], y = (int64_t)a[1] - b[1], z = (int64_t)a[2] - b[2];
  return x * x + y * y + z * z;
}

/* S: m x 3 sample colours (draw order); picks_out[k]: sample positions */
static void kmeanspp_int(const uint8_t* S, int64_t m, int k, mt_t* rs, int32_t* picks_out) {
  const int T = 2 + (int)log((double)k);
  int64_t* closest = (int64_t*)malloc(sizeof(int64_t) * (size_t)m);
  int64_t* cum = (int64_t*)malloc(sizeof(int64_t) * (size_t)m);
  double* rv = (double*)malloc(sizeof(double) * (size_t)T);
  int64_t* cand = (int64_t*)malloc(sizeof(int64_t) * (size_t)T);
  int64_t* pots = (int64_t*)malloc(sizeof(int64_t) * (size_t)T);
  const int64_t first = mt_choice_uniform(rs, m);
  picks_out[0] = (int32_t)first;
  int64_t pot = 0;
  for (int64_t i = 0; i < m; ++i) { closest[i] = d2(S + 3 * i, S + 3 * first); pot += closest[i]; }
  for (int c = 1; c < k; ++c) {
    for (int t = 0; t < T; ++t) rv[t] = mt_double(rs) * (double)pot;      /* uniform(size=T) * current_pot */
    int64_t acc = 0;
    for (int64_t i = 0; i < m; ++i) { acc += closest[i]; cum[i] = acc; }  /* stable_cumsum: exact below 2^53 */
    for (int t = 0; t < T; ++t) {                                         /* searchsorted(cum, r, 'left'), clipped */
      int64_t lo = 0, hi = m;
      while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((double)cum[mid] < rv[t]) lo = mid + 1; else hi = mid;
      }
      cand[t] = lo < m ? lo : m - 1;
    }
#pragma omp parallel for schedule(static) if (m >= 16384)
    for (int t = 0; t < T; ++t) {
      const uint8_t* cc = S + 3 * cand[t];
      int64_t p = 0;
      for (int64_t i = 0; i < m; ++i) {
        const int64_t d = d2(S + 3 * i, cc);
        p += d < closest[i] ? d : closest[i];
      }
      pots[t] = p;
    }
    int best = 0;
    for (int t = 1; t < T; ++t) if (pots[t] < pots[best]) best = t;        /* np.argmin: first minimum */
    pot = pots[best];
    const uint8_t* cc = S + 3 * cand[best];
#pragma omp parallel for schedule(static) if (m >= 65536)
    for (int64_t i = 0; i < m; ++i) {
      const int64_t d = d2(S + 3 * i, cc);
      if (d < closest[i]) closest[i] = d;
    }
    picks_out[c] = (int32_t)cand[best];
  }
  free(closest); free(cum); free(rv); free(cand); free(pots);
}

/* ---- E-step (km64_estep.c's expression), shared out over threads: every sample is independent -------------------
 * centres in structure-of-arrays form so that the distances of 16 consecutive centres are evaluated as vectors (each
 * lane is the same IEEE fma chain); the scan over them keeps the FIRST minimum */
#define EB 16
static void estep(const double* X, int64_t n, const double* C, const double* csq, int64_t k, int32_t* lab) {
  const int64_t kp = (k + EB - 1) / EB * EB;
  double* c0 = (double*)malloc(sizeof(double) * 4 * (size_t)kp);
  double *c1 = c0 + kp, *c2 = c1 + kp, *cs = c2 + kp;
  for (int64_t j = 0; j < kp; ++j) {
    c0[j] = j < k ? C[3 * j] : 0.0; c1[j] = j < k ? C[3 * j + 1] : 0.0; c2[j] = j < k ? C[3 * j + 2] : 0.0;
    cs[j] = j < k ? csq[j] : INFINITY;
  }
#pragma omp parallel for schedule(static) if (n * kp >= 2000000)
  for (int64_t i = 0; i < n; ++i) {
    const double x0 = X[3 * i], x1 = X[3 * i + 1], x2 = X[3 * i + 2];
    double best = INFINITY;
    int32_t bj = 0;
    for (int64_t j0 = 0; j0 < kp; j0 += EB) {
      double d[EB];
      for (int q = 0; q < EB; ++q) d[q] = cs[j0 + q] + (-2.0 * fma(x2, c2[j0 + q], fma(x1, c1[j0 + q], x0 * c0[j0 + q])));
      for (int q = 0; q < EB; ++q)
        if (d[q] < best) { best = d[q]; bj = (int32_t)(j0 + q); }
    }
    lab[i] = bj;
  }
  free(c0);
}

typedef struct { double w; int32_t j; } wj_t;
static int cmp_wj(const void* a, const void* b) {
  const wj_t* x = (const wj_t*)a; const wj_t* y = (const wj_t*)b;
  if (x->w < y->w) return -1;
  if (x->w > y->w) return 1;
  return x->j < y->j ? -1 : (x->j > y->j ? 1 : 0);                         /* argsort_kind 0: stable (weight, index) */
}

/* MiniBatchKMeans(k, batch_size=1000, random_state=seed, n_init='auto').fit_predict on n integer colours (rgb: n x 3).
 * Outputs: centres_out k x 3, labels_out n (may be NULL: skip the final E-step), picks_out k (k-means++ picks, positions
 * in the init sample), init_idx_out (may be NULL) init_size rows, info_out[8] = {n_steps, init_size, words_consumed,
 * n_reassigned, 0...}.  max_steps < 0: sklearn's own limit.  Returns 0. */
void npy_argsort_f64(const double* v, int64_t n, int64_t* order);

int mbk_fit(const uint8_t* rgb, int64_t n, int32_t k, uint32_t seed, int64_t max_steps, int32_t n_threads, int32_t argsort_kind,
            double* centres_out, int32_t* labels_out, int32_t* picks_out, int64_t* init_idx_out, int64_t* info_out) {
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
  if (n <= 0 || k <= 0 || k > n) return -1;
  mt_t rs;
  mt_seed(&rs, seed);
  const int bs = n < 1000 ? (int)n : 1000;
  int64_t init_size = 3 * (int64_t)bs;
  if (init_size < k) init_size = 3 * (int64_t)k;
  if (init_size > n) init_size = n;
  int64_t* idx = (int64_t*)malloc(sizeof(int64_t) * (size_t)init_size);
  mt_randint(&rs, n, init_size, idx);                                      /* validation_indices: stream position only */
  if (init_size < n) mt_randint(&rs, n, init_size, idx);                   /* init_indices */
  else for (int64_t i = 0; i < n; ++i) idx[i] = i;
  uint8_t* S = (uint8_t*)malloc(3 * (size_t)init_size);
  for (int64_t i = 0; i < init_size; ++i) memcpy(S + 3 * i, rgb + 3 * idx[i], 3);
  kmeanspp_int(S, init_size, k, &rs, picks_out);
  double* C = (double*)malloc(sizeof(double) * 3 * (size_t)k);
  double* Cn = (double*)malloc(sizeof(double) * 3 * (size_t)k);
  double* csq = (double*)malloc(sizeof(double) * (size_t)k);
  double* W = (double*)calloc((size_t)k, sizeof(double));
  for (int j = 0; j < k; ++j)
    for (int f = 0; f < 3; ++f) C[3 * j + f] = (double)S[3 * picks_out[j] + f];
  if (init_idx_out) memcpy(init_idx_out, idx, sizeof(int64_t) * (size_t)init_size);
  free(S);
  int64_t* bidx = (int64_t*)malloc(sizeof(int64_t) * (size_t)bs);
  double* Xb = (double*)malloc(sizeof(double) * 3 * (size_t)bs);
  int32_t* lab = (int32_t*)malloc(sizeof(int32_t) * (size_t)bs);
  int32_t* newc = (int32_t*)malloc(sizeof(int32_t) * (size_t)bs);
  uint8_t* to_re = (uint8_t*)malloc((size_t)k);
  wj_t* order = (wj_t*)malloc(sizeof(wj_t) * (size_t)k);
  int64_t* norder = (int64_t*)malloc(sizeof(int64_t) * (size_t)k);
  int64_t n_steps = (100 * n) / bs;
  if (max_steps >= 0 && max_steps < n_steps) n_steps = max_steps;
  double ewa = 0.0, ewa_min = 0.0;
  int have_ewa = 0, have_min = 0, no_impr = 0;
  int64_t since = 0, steps_done = 0, n_reassigned = 0;
  for (int64_t s = 0; s < n_steps; ++s) {
    steps_done = s + 1;
    mt_randint(&rs, n, bs, bidx);
    for (int b = 0; b < bs; ++b)
      for (int f = 0; f < 3; ++f) Xb[3 * b + f] = (double)rgb[3 * bidx[b] + f];
    since += bs;
    int any_zero = 0;
    for (int j = 0; j < k && !any_zero; ++j) any_zero = W[j] == 0.0;
    const int do_reassign = any_zero || since >= 10 * (int64_t)k;          /* _random_reassign, before the update */
    if (do_reassign) since = 0;
    for (int j = 0; j < k; ++j) csq[j] = (C[3 * j] * C[3 * j] + C[3 * j + 2] * C[3 * j + 2]) + C[3 * j + 1] * C[3 * j + 1];
    estep(Xb, bs, C, csq, k, lab);
    double inertia = 0.0;                                                  /* _inertia_dense, one thread: batch order */
    for (int b = 0; b < bs; ++b) {
      const double* c = C + 3 * lab[b];
      double r = 0.0;
      for (int f = 0; f < 3; ++f) { const double d = Xb[3 * b + f] - c[f]; r += d * d; }
      inertia += r * 1.0;
    }
    /* update_center_dense: members of every centre in batch order */
    memcpy(Cn, C, sizeof(double) * 3 * (size_t)k);
    memset(to_re, 0, (size_t)k);                                           /* reused as "already started" flag here */
    for (int b = 0; b < bs; ++b) {
      const int j = lab[b];
      if (!to_re[j]) { to_re[j] = 1; for (int f = 0; f < 3; ++f) Cn[3 * j + f] = C[3 * j + f] * W[j]; }
      for (int f = 0; f < 3; ++f) Cn[3 * j + f] += Xb[3 * b + f] * 1.0;
    }
    /* (adding x_b to centre j while walking b upwards IS the per-centre batch order: the centres do not interact) */
    {
      /* counts per centre, then the rescale; a centre's wsum is its member count (weights 1.0 added one by one) */
      int32_t* cnt = (int32_t*)calloc((size_t)k, sizeof(int32_t));
      for (int b = 0; b < bs; ++b) cnt[lab[b]]++;
      for (int j = 0; j < k; ++j) {
        if (!cnt[j]) continue;
        W[j] += (double)cnt[j];
        const double alpha = 1.0 / W[j];
        for (int f = 0; f < 3; ++f) Cn[3 * j + f] *= alpha;
      }
      free(cnt);
    }
    if (do_reassign) {
      double wmax = W[0];
      for (int j = 1; j < k; ++j) if (W[j] > wmax) wmax = W[j];
      const double thr = 0.01 * wmax;
      int64_t cnt = 0;
      for (int j = 0; j < k; ++j) { to_re[j] = W[j] < thr; cnt += to_re[j]; }
      if ((double)cnt > 0.5 * (double)bs) {
        if (argsort_kind == 1) {
          npy_argsort_f64(W, k, norder);
          for (int64_t r = (int64_t)(0.5 * (double)bs); r < k; ++r) to_re[norder[r]] = 0;
        } else {
          for (int j = 0; j < k; ++j) { order[j].w = W[j]; order[j].j = j; }
          qsort(order, (size_t)k, sizeof(wj_t), cmp_wj);
          for (int64_t r = (int64_t)(0.5 * (double)bs); r < k; ++r) to_re[order[r].j] = 0;
        }
      }
      int nre = 0;
      for (int j = 0; j < k; ++j) nre += to_re[j];
      if (nre) {
        mt_permutation(&rs, bs, nre, newc);                                /* choice(bs, replace=False, size=nre) */
        int r = 0;
        for (int j = 0; j < k; ++j)
          if (to_re[j]) { for (int f = 0; f < 3; ++f) Cn[3 * j + f] = Xb[3 * newc[r] + f]; ++r; }
        n_reassigned += nre;
      }
      double wmin = INFINITY;
      for (int j = 0; j < k; ++j) if (!to_re[j] && W[j] < wmin) wmin = W[j];
      for (int j = 0; j < k; ++j) if (to_re[j]) W[j] = wmin;
    }
    { double* t = C; C = Cn; Cn = t; }
    /* _mini_batch_convergence */
    const double binert = inertia / (double)bs;
    if (s + 1 == 1) continue;
    if (!have_ewa) { ewa = binert; have_ewa = 1; }
    else {
      double a = (double)bs * 2.0 / ((double)n + 1.0);
      if (a > 1.0) a = 1.0;
      ewa = ewa * (1.0 - a) + binert * a;
    }
    if (!have_min || ewa < ewa_min) { no_impr = 0; ewa_min = ewa; have_min = 1; }
    else no_impr++;
    if (no_impr >= 10) break;
  }
  memcpy(centres_out, C, sizeof(double) * 3 * (size_t)k);
  if (labels_out) {
    double* X = (double*)malloc(sizeof(double) * 3 * (size_t)n);
    for (int64_t i = 0; i < 3 * n; ++i) X[i] = (double)rgb[i];
    for (int j = 0; j < k; ++j) csq[j] = (C[3 * j] * C[3 * j] + C[3 * j + 2] * C[3 * j + 2]) + C[3 * j + 1] * C[3 * j + 1];
    estep(X, n, C, csq, k, labels_out);
    free(X);
  }
  if (info_out) {
    info_out[0] = steps_done; info_out[1] = init_size; info_out[2] = rs.consumed; info_out[3] = n_reassigned;
    info_out[4] = info_out[5] = info_out[6] = info_out[7] = 0;
  }
  free(idx); free(C); free(Cn); free(csq); free(W); free(bidx); free(Xb); free(lab); free(newc); free(to_re); free(order); free(norder);
  return 0;
}

/* the k-means++ picks alone (tests of the init chain at sizes where numpy needs minutes) */
int mbk_init_picks(const uint8_t* rgb, int64_t n, int32_t k, uint32_t seed, int32_t n_threads, int32_t* picks_out, int64_t* init_idx_out) {
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
  if (n <= 0 || k <= 0 || k > n) return -1;
  mt_t rs;
  mt_seed(&rs, seed);
  const int bs = n < 1000 ? (int)n : 1000;
  int64_t init_size = 3 * (int64_t)bs;
  if (init_size < k) init_size = 3 * (int64_t)k;
  if (init_size > n) init_size = n;
  int64_t* idx = (int64_t*)malloc(sizeof(int64_t) * (size_t)init_size);
  mt_randint(&rs, n, init_size, idx);
  if (init_size < n) mt_randint(&rs, n, init_size, idx);
  else for (int64_t i = 0; i < n; ++i) idx[i] = i;
  uint8_t* S = (uint8_t*)malloc(3 * (size_t)init_size);
  for (int64_t i = 0; i < init_size; ++i) memcpy(S + 3 * i, rgb + 3 * idx[i], 3);
  kmeanspp_int(S, init_size, k, &rs, picks_out);
  if (init_idx_out) memcpy(init_idx_out, idx, sizeof(int64_t) * (size_t)init_size);
  free(S); free(idx);
  return 0;
}
