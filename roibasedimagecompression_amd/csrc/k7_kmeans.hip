// K7: KMeans(n_clusters = k, random_state = 42, n_init = 1, lloyd) used by the reference to split
// oversize colour clusters (encoder/compression/clustering.py:720-775, sklearn 1.7.2 underneath).
//
// Canonical arithmetic "KM64" (see oracle/rhccq_oracle.py): greedy k-means++ on EXACT integer
// squared distances (cumulative sums / potentials are order independent), MT19937 uniforms supplied
// by the host exactly as numpy's RandomState(42) emits them; Lloyd in float64 on mean-centred
// coordinates, E-step distance csq + (-2 * fma-chain dot) exactly as sklearn's dgemm / einsum evaluate it
// (rhccq_common.h; the file is built with -ffp-contract=off, the only fused operations are explicit), centres from exact integer member sums, sklearn's convergence rules.
//
// MI355X design: one 1024-thread workgroup per split problem (problems of a frame are batched in one
// launch, grid = #problems); points, labels, centres and integer accumulators live in LDS
// (<= 10240 points: 40 + 40 + 32 + 16 KB); the k-means++ cumulative-sum search is a block scan +
// per-thread chunk search; candidate potentials are 64-bit integer block reductions.  float64 VALU
// bound (K = 3: not an MFMA shape).
#include "rhccq_common.h"

namespace rhccq {

#include "kpp_flat.h"

constexpr int kKmThreads = 1024;
constexpr int kKmWaves = kKmThreads / 64;
constexpr int kKmCentLds = 1024;   // centres kept in LDS up to this k
constexpr int kTMax = 16;
constexpr int kKmPts = 4;          // points a thread holds in registers per pass of the Lloyd E-step
constexpr int kKmFlatS = 10;       // k-means++ with every point in a register up to this many points per thread (LDS-resident inputs)
static_assert(sizeof(FlatShared) <= sizeof(unsigned) * kKmCentLds * 4, "the register k-means++ borrows the member-sum table");

struct KmShared {
  unsigned long long red64[kKmWaves * kTMax];
  unsigned long long scan_red[kKmWaves + 1];
  double dred[kKmWaves];
  int ired[kKmWaves];
  int cand[kTMax];
  unsigned long long pots[kTMax];
  int flag;
  int best;
  double bestd;
  int besti;
};

__device__ __forceinline__ unsigned long long block_exscan64(unsigned long long v, KmShared& sh, unsigned long long* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned long long inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    unsigned long long t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  __syncthreads();
  if (lane == 63) sh.scan_red[w] = inc;
  __syncthreads();
  unsigned long long base = 0, tot = 0;
  for (int i = 0; i < kKmWaves; ++i) {
    if (i < w) base += sh.scan_red[i];
    tot += sh.scan_red[i];
  }
  *total = tot;
  return base + inc - v;
}

// KM64 distance: csq + (-2 * fma(x2, c2, fma(x1, c1, x0*c0)))   (rhccq_common.h)
__device__ __forceinline__ double km64_dist(double x0, double x1, double x2, const double* c) {
  return c[3] + (-2.0 * km64_dot(x0, x1, x2, c[0], c[1], c[2]));
}

__global__ __launch_bounds__(kKmThreads) void kmeans_kernel(const uint32_t* __restrict__ keys, const int32_t* __restrict__ desc,
                                                             const int64_t* __restrict__ koff, const double* __restrict__ rand,
                                                             double* __restrict__ work, int32_t* __restrict__ labels_out,
                                                             int32_t* __restrict__ info, uint32_t* __restrict__ gpts, int gstride,
                                                             int max_iter) {
  __shared__ uint32_t s_keys[RHCCQ_KM_LDS_MAX];
  __shared__ uint32_t s_aux[RHCCQ_KM_LDS_MAX];
  __shared__ double s_cent[kKmCentLds * 4];
  __shared__ __align__(16) unsigned s_sum[kKmCentLds * 4];
  __shared__ KmShared sh;

  const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int off = desc[p * 6 + 0], n = desc[p * 6 + 1], k = desc[p * 6 + 2], rand_off = desc[p * 6 + 3];
  const int first = desc[p * 6 + 4], T = desc[p * 6 + 5];
  int32_t* lab_out = labels_out + off;
  if (n <= 0 || k <= 0) return;

  uint32_t* P;      // packed colours
  uint32_t* aux;    // closest (k-means++) then labels (Lloyd); bit 31 = "taken" during relocation
  if (n <= RHCCQ_KM_LDS_MAX) { P = s_keys; aux = s_aux; }
  else { P = gpts + (size_t)p * 2 * gstride; aux = P + gstride; }
  double* wk = work + 8 * koff[p];
  double* C;        // [k][4] = x, y, z, csq
  unsigned* S;      // [k][4] = sum r, g, b, count
  if (k <= kKmCentLds) { C = s_cent; S = s_sum; }
  else { C = wk; S = reinterpret_cast<unsigned*>(wk + 4 * (size_t)k); }
  int* chosen = reinterpret_cast<int*>(wk + 6 * (size_t)k);   // [k] init indices (global, small)

  // ---- load, mean, tolerance ----------------------------------------------------------------
  unsigned long long sr = 0, sg = 0, sb = 0, qr = 0, qg = 0, qb = 0;
  for (int i = tid; i < n; i += kKmThreads) {
    const uint32_t kk = keys[off + i];
    P[i] = kk;
    const unsigned r = key_r(kk), g = key_g(kk), b = key_b(kk);
    sr += r; sg += g; sb += b;
    qr += r * r; qg += g * g; qb += b * b;
  }
  sr = block_sum<unsigned long long>(sr, sh.red64);
  sg = block_sum<unsigned long long>(sg, sh.red64);
  sb = block_sum<unsigned long long>(sb, sh.red64);
  qr = block_sum<unsigned long long>(qr, sh.red64);
  qg = block_sum<unsigned long long>(qg, sh.red64);
  qb = block_sum<unsigned long long>(qb, sh.red64);
  const double dn = (double)n;
  const double m0 = (double)sr / dn, m1 = (double)sg / dn, m2 = (double)sb / dn;
  const double v0 = (double)((long long)((unsigned long long)n * qr - sr * sr)) / (dn * dn);
  const double v1 = (double)((long long)((unsigned long long)n * qg - sg * sg)) / (dn * dn);
  const double v2 = (double)((long long)((unsigned long long)n * qb - sb * sb)) / (dn * dn);
  const double tol = ((v0 + v1) + v2) / 3.0 * 1e-4;

  // ---- k-means++ in exact integers ----------------------------------------------------------
  // (up to 10 points per thread: every point in a register, two barriers per pick -- kpp_flat.h; the loop below, a block scan, a
  // candidate search and a block reduction per pick, ~7 us each, serves larger inputs)
  if (n <= 16 * 64 * 4) {
    __syncthreads();      // (small inputs -- the one large split of a 4K frame's level 3 has 3 900 colours: four working waves, see kpp_flat.h)
    kpp_flat<16, 4>([&](int i) { return P[i]; }, n, k, T, first, rand + rand_off, chosen, *reinterpret_cast<FlatShared*>(s_sum));
  } else if (n <= kKmFlatS * kKmThreads) {
    __syncthreads();
    kpp_flat<kKmFlatS, kKmWaves>([&](int i) { return P[i]; }, n, k, T, first, rand + rand_off, chosen, *reinterpret_cast<FlatShared*>(s_sum));
  } else {
    const int per = (n + kKmThreads - 1) / kKmThreads;
    const int lo = min(tid * per, n), hi = min(lo + per, n);
    unsigned long long pot;
    {
      const uint32_t kf = P[first];
      unsigned long long s = 0;
      for (int i = tid; i < n; i += kKmThreads) {
        const unsigned d = (unsigned)dist2_keys(P[i], kf);
        aux[i] = d;
        s += d;
      }
      pot = block_sum<unsigned long long>(s, sh.red64);
      if (tid == 0) chosen[0] = first;
    }
    for (int c = 1; c < k; ++c) {
      const double* u = rand + rand_off + (size_t)(c - 1) * T;
      const double dpot = (double)pot;
      // cumulative-sum search (np.searchsorted(cumsum(closest), u*pot, 'left'), clipped to n-1)
      unsigned long long loc = 0;
      for (int i = lo; i < hi; ++i) loc += aux[i];
      unsigned long long tot;
      const unsigned long long base = block_exscan64(loc, sh, &tot);
      if (tid < T) sh.cand[tid] = (u[tid] * dpot <= 0.0) ? 0 : n - 1;
      __syncthreads();
      if (loc > 0) {
        for (int t = 0; t < T; ++t) {
          const double r = u[t] * dpot;
          if ((double)base < r && r <= (double)(base + loc)) {
            unsigned long long cum = base;
            int i = lo;
            for (; i < hi; ++i) {
              cum += aux[i];
              if ((double)cum >= r) break;
            }
            sh.cand[t] = i < hi ? i : hi - 1;
          }
        }
      }
      __syncthreads();
      // potentials of the T candidates
      uint32_t ck[kTMax];
      unsigned long long acc[kTMax];
  #pragma unroll
      for (int t = 0; t < kTMax; ++t) { acc[t] = 0; ck[t] = t < T ? P[sh.cand[t]] : 0u; }
      for (int i = tid; i < n; i += kKmThreads) {
        const uint32_t ki = P[i];
        const unsigned cl = aux[i];
  #pragma unroll
        for (int t = 0; t < kTMax; ++t)
          if (t < T) acc[t] += min(cl, (unsigned)dist2_keys(ki, ck[t]));
      }
  #pragma unroll
      for (int t = 0; t < kTMax; ++t) {
        if (t < T) {
          const unsigned long long w = wave_sum(acc[t]);
          if (lane == 0) sh.red64[t * kKmWaves + wave] = w;
        }
      }
      __syncthreads();
      if (tid < T) {
        unsigned long long s = 0;
        for (int w = 0; w < kKmWaves; ++w) s += sh.red64[tid * kKmWaves + w];
        sh.pots[tid] = s;
      }
      __syncthreads();
      int best = 0;
      unsigned long long bp = sh.pots[0];
      for (int t = 1; t < T; ++t)
        if (sh.pots[t] < bp) { bp = sh.pots[t]; best = t; }
      pot = bp;
      const int bi = sh.cand[best];
      const uint32_t kb = P[bi];
      for (int i = tid; i < n; i += kKmThreads) aux[i] = min(aux[i], (unsigned)dist2_keys(P[i], kb));
      if (tid == 0) chosen[c] = bi;
      __syncthreads();
    }
  }
  __threadfence_block();
  __syncthreads();

  // ---- Lloyd --------------------------------------------------------------------------------
  for (int j = tid; j < k; j += kKmThreads) {
    const uint32_t kk = P[chosen[j]];
    const double c0 = (double)key_r(kk) - m0, c1 = (double)key_g(kk) - m1, c2 = (double)key_b(kk) - m2;
    C[j * 4 + 0] = c0; C[j * 4 + 1] = c1; C[j * 4 + 2] = c2;
    C[j * 4 + 3] = km64_csq(c0, c1, c2);
  }
  for (int i = tid; i < n; i += kKmThreads) aux[i] = 0x7fffffffu;   // labels_old = -1
  __syncthreads();
  int n_iter = 0, strict = 0, relocated = 0;
  for (int it = 0; it < max_iter; ++it) {
    n_iter = it + 1;
    for (int j = tid; j < 4 * k; j += kKmThreads) S[j] = 0;
    if (tid == 0) sh.flag = 0;
    __syncthreads();
    int changed = 0;
    // E-step, centres in the OUTER loop: a thread keeps up to kKmPts of its points in registers and reads every centre
    // once for all of them (the first version walked the centres once per point: with all threads reading the same
    // centre the LDS broadcast reads, two b128 per pair, were the whole iteration).  Same per-pair arithmetic, centres
    // still visited in ascending order: the first arg-min is unchanged.
    for (int i0 = tid; i0 < n; i0 += kKmThreads * kKmPts) {
      double x0[kKmPts], x1[kKmPts], x2[kKmPts], bd[kKmPts];
      int bj[kKmPts];
      uint32_t kk[kKmPts];
#pragma unroll
      for (int q = 0; q < kKmPts; ++q) {
        const int i = i0 + q * kKmThreads;
        kk[q] = i < n ? P[i] : 0u;
        x0[q] = (double)key_r(kk[q]) - m0; x1[q] = (double)key_g(kk[q]) - m1; x2[q] = (double)key_b(kk[q]) - m2;
        bd[q] = INFINITY;
        bj[q] = 0;
      }
      for (int j = 0; j < k; ++j) {
        const double c0 = C[4 * j], c1 = C[4 * j + 1], c2 = C[4 * j + 2], cs = C[4 * j + 3];
#pragma unroll
        for (int q = 0; q < kKmPts; ++q) {
          const double d = cs + (-2.0 * km64_dot(x0[q], x1[q], x2[q], c0, c1, c2));
          if (d < bd[q]) { bd[q] = d; bj[q] = j; }
        }
      }
#pragma unroll
      for (int q = 0; q < kKmPts; ++q) {
        const int i = i0 + q * kKmThreads;
        if (i >= n) continue;
        if ((uint32_t)bj[q] != aux[i]) changed = 1;
        aux[i] = (uint32_t)bj[q];
        atomicAdd(&S[bj[q] * 4 + 0], key_r(kk[q]));
        atomicAdd(&S[bj[q] * 4 + 1], key_g(kk[q]));
        atomicAdd(&S[bj[q] * 4 + 2], key_b(kk[q]));
        atomicAdd(&S[bj[q] * 4 + 3], 1u);
      }
    }
    if (changed) sh.flag = 1;                           // benign race: all writers store 1
    __syncthreads();
    const int any_changed = sh.flag;
    // empty clusters (sklearn _relocate_empty_clusters_dense)
    int n_empty = 0;
    for (int j = tid; j < k; j += kKmThreads) n_empty += S[j * 4 + 3] == 0u;
    n_empty = block_sum<int>(n_empty, sh.ired);
    if (n_empty > 0) {
      // only the clusters that were empty BEFORE any relocation are refilled (oracle: `empty` is
      // computed once); mark them so that clusters emptied by a relocation are not picked up
      for (int j = tid; j < k; j += kKmThreads)
        if (S[j * 4 + 3] == 0u) S[j * 4 + 2] = 0xffffffffu;
      __syncthreads();
      for (int e = 0, done = 0; e < k && done < n_empty; ++e) {
        if (!(S[e * 4 + 3] == 0u && S[e * 4 + 2] == 0xffffffffu)) { continue; }
        // farthest not-yet-taken point from its own (old) centre: (distance desc, index asc)
        double bd = -1.0;
        int bi = 0x7fffffff;
        for (int i = tid; i < n; i += kKmThreads) {
          const uint32_t l = aux[i];
          if (l & 0x80000000u) continue;
          const uint32_t kk = P[i];
          const double* cc = C + 4 * l;
          const double d0 = ((double)key_r(kk) - m0) - cc[0], d1 = ((double)key_g(kk) - m1) - cc[1], d2 = ((double)key_b(kk) - m2) - cc[2];
          const double d = (d0 * d0 + d1 * d1) + d2 * d2;
          if (d > bd || (d == bd && i < bi)) { bd = d; bi = i; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const double od = __shfl_down(bd, o, 64);
          const int oi = __shfl_down(bi, o, 64);
          if (od > bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
        }
        __syncthreads();
        if (lane == 0) { sh.dred[wave] = bd; sh.ired[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
          double gd = sh.dred[0];
          int gi = sh.ired[0];
          for (int w = 1; w < kKmWaves; ++w)
            if (sh.dred[w] > gd || (sh.dred[w] == gd && sh.ired[w] < gi)) { gd = sh.dred[w]; gi = sh.ired[w]; }
          sh.bestd = gd; sh.besti = gi;
        }
        __syncthreads();
        if (done == 0 && !(sh.bestd > 0.0)) break;       // dist.max() == 0: nothing to relocate
        if (tid == 0) {
          const int f = sh.besti;
          const uint32_t old = aux[f] & 0x7fffffffu;
          const uint32_t kk = P[f];
          S[old * 4 + 0] -= key_r(kk); S[old * 4 + 1] -= key_g(kk); S[old * 4 + 2] -= key_b(kk); S[old * 4 + 3] -= 1u;
          S[e * 4 + 0] = key_r(kk); S[e * 4 + 1] = key_g(kk); S[e * 4 + 2] = key_b(kk); S[e * 4 + 3] = 1u;
          aux[f] |= 0x80000000u;
        }
        ++done; ++relocated;
        __syncthreads();
      }
      for (int i = tid; i < n; i += kKmThreads) aux[i] &= 0x7fffffffu;
      for (int j = tid; j < k; j += kKmThreads)
        if (S[j * 4 + 3] == 0u) S[j * 4 + 2] = 0u;
      __syncthreads();
    }
    // new centres, per-centre squared shift kept in the csq slot of a scratch copy
    // heaviest cluster (first arg-max of the counts) for still-empty clusters
    int hv = 0;
    if (n_empty > 0) {
      unsigned bc = 0; int bj = 0x7fffffff;
      for (int j = tid; j < k; j += kKmThreads) {
        const unsigned cnt = S[j * 4 + 3];
        if (cnt > bc || (cnt == bc && j < bj)) { bc = cnt; bj = j; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const unsigned oc = __shfl_down(bc, o, 64);
        const int oj = __shfl_down(bj, o, 64);
        if (oc > bc || (oc == bc && oj < bj)) { bc = oc; bj = oj; }
      }
      __syncthreads();
      if (lane == 0) { sh.scan_red[wave] = bc; sh.ired[wave] = bj; }
      __syncthreads();
      if (tid == 0) {
        unsigned long long gc = sh.scan_red[0]; int gj = sh.ired[0];
        for (int w = 1; w < kKmWaves; ++w)
          if (sh.scan_red[w] > gc || (sh.scan_red[w] == gc && sh.ired[w] < gj)) { gc = sh.scan_red[w]; gj = sh.ired[w]; }
        sh.best = gj;
      }
      __syncthreads();
      hv = sh.best;
    }
    // shift_j stored temporarily in wk[6k + ...]? keep it simple: two passes over k
    double* shift = wk + 7 * (size_t)k;                 // [k] doubles of scratch (global)
    double hx = 0, hy = 0, hz = 0;
    if (n_empty > 0) {
      const unsigned cnt = S[hv * 4 + 3];
      hx = (double)S[hv * 4 + 0] / (double)cnt - m0;
      hy = (double)S[hv * 4 + 1] / (double)cnt - m1;
      hz = (double)S[hv * 4 + 2] / (double)cnt - m2;
    }
    __syncthreads();
    for (int j = tid; j < k; j += kKmThreads) {
      const unsigned cnt = S[j * 4 + 3];
      double c0, c1, c2;
      if (cnt > 0) {
        c0 = (double)S[j * 4 + 0] / (double)cnt - m0;
        c1 = (double)S[j * 4 + 1] / (double)cnt - m1;
        c2 = (double)S[j * 4 + 2] / (double)cnt - m2;
      } else { c0 = hx; c1 = hy; c2 = hz; }
      const double d0 = c0 - C[j * 4 + 0], d1 = c1 - C[j * 4 + 1], d2 = c2 - C[j * 4 + 2];
      shift[j] = (d0 * d0 + d1 * d1) + d2 * d2;
      C[j * 4 + 0] = c0; C[j * 4 + 1] = c1; C[j * 4 + 2] = c2;
      C[j * 4 + 3] = km64_csq(c0, c1, c2);
    }
    __threadfence_block();
    __syncthreads();
    if (!any_changed) { strict = 1; break; }
    if (wave == 0) {
      // sum of the shifts in index order, as the oracle adds them: the wave loads 64 at a time (one coalesced
      // access instead of k dependent ones by a single thread) and folds them through readlane, left to right
      double tot = 0.0;
      for (int j0 = 0; j0 < k; j0 += 64) {
        const double v = j0 + lane < k ? shift[j0 + lane] : 0.0;
        const int m = min(64, k - j0);
        for (int l = 0; l < m; ++l) tot = tot + __shfl(v, l, 64);
      }
      if (lane == 0) sh.flag = tot <= tol;
    }
    __syncthreads();
    const int stop = sh.flag;
    __syncthreads();
    if (stop) break;
  }
  if (!strict) {
    for (int i = tid; i < n; i += kKmThreads) {
      const uint32_t kk = P[i];
      const double x0 = (double)key_r(kk) - m0, x1 = (double)key_g(kk) - m1, x2 = (double)key_b(kk) - m2;
      double bd = km64_dist(x0, x1, x2, C);
      int bj = 0;
      for (int j = 1; j < k; ++j) {
        const double d = km64_dist(x0, x1, x2, C + 4 * j);
        if (d < bd) { bd = d; bj = j; }
      }
      aux[i] = (uint32_t)bj;
    }
    __syncthreads();
  }
  for (int i = tid; i < n; i += kKmThreads) lab_out[i] = (int32_t)aux[i];
  if (tid == 0) {
    info[p * 4 + 0] = n_iter; info[p * 4 + 1] = strict; info[p * 4 + 2] = relocated; info[p * 4 + 3] = 0;
  }
}

// ---- K2: integer sums per label -> floor mean ---------------------------------------------------
__global__ __launch_bounds__(256) void cluster_sums_kernel(const uint32_t* __restrict__ keys, const int32_t* __restrict__ labels, int64_t n,
                                                           unsigned long long* __restrict__ sums) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t l = labels[i];
    if (l < 0) continue;
    const uint32_t kk = keys[i];
    atomicAdd(&sums[(size_t)l * 4 + 0], (unsigned long long)key_r(kk));
    atomicAdd(&sums[(size_t)l * 4 + 1], (unsigned long long)key_g(kk));
    atomicAdd(&sums[(size_t)l * 4 + 2], (unsigned long long)key_b(kk));
    atomicAdd(&sums[(size_t)l * 4 + 3], 1ull);
  }
}

// The same sums for up to 2^24 points (a palette: 255 x 2^24 < 2^32, so two 32-bit fields share one 64-bit word without carries):
// (r | g) and (b | count) = TWO atomics per flush instead of four, and a thread walks 8 consecutive points and flushes only when
// the label changes -- a palette is sorted by colour and clusters are compact, so neighbours mostly share their label.  The
// scattered 64-bit atomics execute at the memory side: 12 M of them were 1 ms between level 1 and level 2 of a 4K frame.
// cluster_unpack_kernel then spreads the two words of every label over the four documented fields, in place.
constexpr int kSumRun = 8;
__global__ __launch_bounds__(256) void cluster_sums_packed_kernel(const uint32_t* __restrict__ keys, const int32_t* __restrict__ labels, int64_t n,
                                                                  unsigned long long* __restrict__ sums) {
  const int64_t chunks = (n + kSumRun - 1) / kSumRun;
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < chunks; c += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i0 = c * kSumRun;
    int32_t lab[kSumRun];
    uint32_t kk[kSumRun];
#pragma unroll
    for (int q = 0; q < kSumRun; ++q) {
      const int64_t i = min(i0 + q, n - 1);
      lab[q] = i0 + q < n ? labels[i] : -1;
      kk[q] = keys[i];
    }
    int32_t cur = -1;
    unsigned long long w0 = 0, w1 = 0;
#pragma unroll
    for (int q = 0; q < kSumRun; ++q) {
      if (lab[q] != cur) {
        if (cur >= 0) { atomicAdd(&sums[(size_t)cur * 4 + 0], w0); atomicAdd(&sums[(size_t)cur * 4 + 1], w1); }
        cur = lab[q]; w0 = 0; w1 = 0;
      }
      w0 += ((unsigned long long)key_r(kk[q]) << 32) | key_g(kk[q]);
      w1 += ((unsigned long long)key_b(kk[q]) << 32) | 1ull;
    }
    if (cur >= 0) { atomicAdd(&sums[(size_t)cur * 4 + 0], w0); atomicAdd(&sums[(size_t)cur * 4 + 1], w1); }
  }
}

__global__ void cluster_unpack_kernel(unsigned long long* __restrict__ sums, int64_t k) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= k) return;
  const unsigned long long w0 = sums[j * 4 + 0], w1 = sums[j * 4 + 1];
  sums[j * 4 + 0] = w0 >> 32; sums[j * 4 + 1] = w0 & 0xffffffffull;
  sums[j * 4 + 2] = w1 >> 32; sums[j * 4 + 3] = w1 & 0xffffffffull;
}

__global__ void cluster_means_kernel(const unsigned long long* __restrict__ sums, int64_t k, uint32_t* __restrict__ keys_out) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= k) return;
  const unsigned long long c = sums[j * 4 + 3];
  uint32_t out = 0;
  if (c) out = ((uint32_t)(sums[j * 4 + 0] / c) << 16) | ((uint32_t)(sums[j * 4 + 1] / c) << 8) | (uint32_t)(sums[j * 4 + 2] / c);
  keys_out[j] = out;
}

}  // namespace rhccq

using namespace rhccq;

extern "C" {

int rhccq_kmeans(rhccq_ctx* ctx, const uint32_t* keys, const int32_t* desc, const int64_t* koff, const double* rand,
                 int32_t n_prob, int32_t max_n, double* work, int32_t* labels_out, int32_t* info) {
  if (!ctx || !keys || !desc || !koff || !rand || !work || !labels_out || !info || n_prob <= 0 || max_n < 0)
    return rhccq_fail(ctx, RHCCQ_E_ARG, "kmeans: bad argument");
  uint32_t* gpts = nullptr;
  int stride = 0;
  if (max_n > RHCCQ_KM_LDS_MAX) {
    stride = (max_n + 63) & ~63;
    const size_t bytes = (size_t)n_prob * 2 * stride * sizeof(uint32_t);
    if (ctx->scratch_bytes < bytes) {
      if (ctx->scratch) RHCCQ_HIP(ctx, hipFree(ctx->scratch));
      ctx->scratch = nullptr;
      ctx->scratch_bytes = 0;
      RHCCQ_HIP(ctx, hipMalloc(&ctx->scratch, bytes));
      ctx->scratch_bytes = bytes;
    }
    gpts = (uint32_t*)ctx->scratch;
  }
  hipLaunchKernelGGL(kmeans_kernel, dim3(n_prob), dim3(kKmThreads), 0, ctx->stream, keys, desc, koff, rand, work, labels_out, info, gpts, stride, 300);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_cluster_sums(rhccq_ctx* ctx, const uint32_t* keys, const int32_t* labels, int64_t n, int64_t k, unsigned long long* sums) {
  if (!ctx || n < 0 || k < 0 || (n > 0 && (!keys || !labels || !sums))) return rhccq_fail(ctx, RHCCQ_E_ARG, "cluster_sums: bad argument");
  if (n == 0) return 0;
  if (n <= (1ll << 24) && k > 0) {
    int64_t b = ((n + kSumRun - 1) / kSumRun + 255) / 256;
    if (b > 2048) b = 2048;
    hipLaunchKernelGGL(cluster_sums_packed_kernel, dim3((int)b), dim3(256), 0, ctx->stream, keys, labels, n, sums);
    hipLaunchKernelGGL(cluster_unpack_kernel, dim3((int)((k + 255) / 256)), dim3(256), 0, ctx->stream, sums, k);
  } else {
    int64_t b = (n + 255) / 256;
    if (b > 2048) b = 2048;
    hipLaunchKernelGGL(cluster_sums_kernel, dim3((int)b), dim3(256), 0, ctx->stream, keys, labels, n, sums);
  }
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_cluster_means(rhccq_ctx* ctx, const unsigned long long* sums, int64_t k, uint32_t* keys_out) {
  if (!ctx || k < 0 || (k > 0 && (!sums || !keys_out))) return rhccq_fail(ctx, RHCCQ_E_ARG, "cluster_means: bad argument");
  if (k == 0) return 0;
  hipLaunchKernelGGL(cluster_means_kernel, dim3((int)((k + 255) / 256)), dim3(256), 0, ctx->stream, sums, k, keys_out);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

}  // extern "C"
