// K8: the MiniBatchKMeans branch the reference takes for palettes of >= 10 000 colours
// (encoder/compression/clustering.py:207-230; sklearn 1.7.2 MiniBatchKMeans(n_clusters, batch_size=1000,
// random_state=42, n_init='auto')).  Tier B parity: the algorithm is sklearn's, the random stream of the
// mini-batch loop and the tie-breaking of its argsort are the canonical, reproducible ones of
// oracle/rhccq_oracle.py::minibatch_kmeans_labels, which these kernels reproduce bit-for-bit.
//
// MI355X design
//   init   greedy k-means++ over the (sorted) init sample in EXACT integers.  The chain of k picks is
//          sequential, so one 1024-thread workgroup owns a problem; samples are grouped in blocks of 64
//          consecutive (key-sorted => spatially coherent) samples with a bounding box and the block's
//          max / sum of closest distances, and a candidate only visits blocks whose box is nearer than
//          that max (exact pruning: the skipped samples cannot change).  One wave per candidate,
//          64 boxes tested per wave step (ballot), 64 samples of a hit block evaluated per step.
//   steps  per step two launches for all problems: (A) brute-force E-step of the 1000-point batch
//          against all centres, centres tiled 64 per workgroup through LDS, partial arg-mins written per
//          tile; (B) one workgroup per problem reduces the partials in tile order (first arg-min),
//          accumulates exact integer member sums in an LDS hash keyed by centre, applies the per-centre
//          learning-rate update, the low-count reassignment and sklearn's EWA early-stopping rule.
//   assign final E-step over all N points, brute force, 4 points per thread, centres (pre-scaled by -2,
//          exact) tiled through LDS; float64 VALU bound (K = 3 is not an MFMA shape).
#include "rhccq_common.h"

namespace rhccq {

// ------------------------------------------------------------------------------------------------
// shared helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long counter_hash(unsigned long long seed, unsigned long long stream, unsigned long long counter) {
  unsigned long long z = seed * 0x9E3779B97F4A7C15ull + stream * 0xD1B54A32D192ED03ull + counter * 0x2545F4914F6CDD1Dull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ unsigned long long bounded(unsigned long long z, unsigned long long n) { return ((z >> 32) * n) >> 32; }

struct MbkP {  // device copy of rhccq_mbk_problem
  long long off, n, k, koff, init_off, init_n, rand_off;
  int first, T;
};

constexpr int kInitThreads = 1024;
constexpr int kInitWaves = kInitThreads / 64;
constexpr int kTMaxI = 24;

struct InitShared {
  unsigned long long scan_red[kInitWaves + 1];
  unsigned long long red64[kInitWaves];
  unsigned long long pots[kTMaxI];
  int cand[kTMaxI];
};

__device__ __forceinline__ unsigned long long init_exscan64(unsigned long long v, InitShared& sh, unsigned long long* total) {
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
  for (int i = 0; i < kInitWaves; ++i) {
    if (i < w) base += sh.scan_red[i];
    tot += sh.scan_red[i];
  }
  *total = tot;
  return base + inc - v;
}

// squared distance from a colour to a box [lo, hi] (packed keys), exact integers
__device__ __forceinline__ int box_dist2(uint32_t k, uint32_t lo, uint32_t hi) {
  const int r = key_r(k), g = key_g(k), b = key_b(k);
  const int dr = max(max((int)key_r(lo) - r, r - (int)key_r(hi)), 0);
  const int dg = max(max((int)key_g(lo) - g, g - (int)key_g(hi)), 0);
  const int db = max(max((int)key_b(lo) - b, b - (int)key_b(hi)), 0);
  return __mul24(dr, dr) + __mul24(dg, dg) + __mul24(db, db);
}

// per problem scratch layout (u32): skey[np] closest[np] and, when the block tables do not fit LDS,
// lo[nb] hi[nb] bmax[nb] bsum[nb]   (nb = ceil(init_n / 64), np = nb * 64)
constexpr int kInitLdsBlocks = 4096;   // block tables in LDS up to 262144 init samples (64 KB)

struct InitTables {
  uint32_t* lo;     // per block: packed min corner of the bounding box
  uint32_t* hi;     // packed max corner
  uint32_t* bmax;   // max closest distance in the block
  uint32_t* bsum;   // sum of closest distances in the block
};

// distance work of one candidate / the winner over the blocks [chunk0, nb) visited with stride
// `chunk_stride`; kCommit = false: returns sum over samples of max(closest - d, 0);
// kCommit = true: lowers closest[] and refreshes bmax / bsum of the touched blocks.
template <bool kCommit>
__device__ __forceinline__ unsigned long long visit_blocks(uint32_t ck, int nb, int chunk0, int chunk_stride, const uint32_t* skey,
                                                           uint32_t* closest, InitTables tb) {
  const int lane = threadIdx.x & 63;
  unsigned long long delta = 0;
  for (int chunk = chunk0; chunk < nb; chunk += chunk_stride) {
    const int b = chunk + lane;
    const bool hit = b < nb && (unsigned)box_dist2(ck, tb.lo[b], tb.hi[b]) < tb.bmax[b];
    unsigned long long mask = __ballot(hit);
    while (mask) {
      // up to four hit blocks per round so that their loads are in flight together
      int bb[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (mask) { bb[q] = chunk + __ffsll((long long)mask) - 1; mask &= mask - 1; }
        else bb[q] = -1;
      }
      uint32_t kk[4], cl[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = (max(bb[q], 0) << 6) + lane;
        kk[q] = skey[i];
        cl[q] = closest[i];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (bb[q] < 0) continue;
        const unsigned d = (unsigned)dist2_keys(ck, kk[q]);
        if (!kCommit) {
          delta += cl[q] > d ? cl[q] - d : 0u;
        } else {
          unsigned c2 = cl[q];
          if (d < c2) { c2 = d; closest[(bb[q] << 6) + lane] = d; }
          unsigned dm = c2, ds = c2;
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) {
            dm = max(dm, (unsigned)__shfl_down(dm, o, 64));
            ds += __shfl_down(ds, o, 64);
          }
          if (lane == 0) { tb.bmax[bb[q]] = dm; tb.bsum[bb[q]] = ds; }
        }
      }
    }
  }
  return delta;
}

__global__ __launch_bounds__(kInitThreads) void mbk_init_kernel(const uint32_t* __restrict__ keys, const MbkP* __restrict__ probs,
                                                                const int32_t* __restrict__ init_idx, const double* __restrict__ rand,
                                                                double* __restrict__ centres, int32_t* __restrict__ chosen,
                                                                uint32_t* scratch, const long long* __restrict__ scratch_off) {
  __shared__ InitShared sh;
  __shared__ uint32_t s_tab[4 * kInitLdsBlocks];
  __shared__ int s_cblock[kTMaxI];
  __shared__ unsigned long long s_cbase[kTMaxI];
  const MbkP P = probs[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = (int)P.init_n, k = (int)P.k, T = P.T;
  const int nb = (n + 63) >> 6, np = nb << 6;
  uint32_t* skey = scratch + scratch_off[blockIdx.x];
  uint32_t* closest = skey + np;
  InitTables tb;
  if (nb <= kInitLdsBlocks) {
    tb.lo = s_tab; tb.hi = s_tab + kInitLdsBlocks; tb.bmax = s_tab + 2 * kInitLdsBlocks; tb.bsum = s_tab + 3 * kInitLdsBlocks;
  } else {
    tb.lo = closest + np; tb.hi = tb.lo + nb; tb.bmax = tb.hi + nb; tb.bsum = tb.bmax + nb;
  }
  int32_t* cho = chosen + P.koff;
  // ---- gather the sample, boxes ------------------------------------------------------------------
  for (int i = tid; i < np; i += kInitThreads) {
    const int src = i < n ? i : n - 1;                  // padding repeats the last sample
    skey[i] = keys[P.off + init_idx[P.init_off + src]];
  }
  __syncthreads();
  const uint32_t kf = skey[P.first];
  for (int b = wave; b < nb; b += kInitWaves) {
    const int i = (b << 6) + lane;
    const uint32_t kk = skey[i];
    unsigned r0 = key_r(kk), r1 = r0, g0 = key_g(kk), g1 = g0, b0 = key_b(kk), b1 = b0;
    unsigned d = i < n ? (unsigned)dist2_keys(kk, kf) : 0u;
    closest[i] = d;
    unsigned dm = d, ds = d;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      r0 = min(r0, (unsigned)__shfl_down(r0, o, 64)); r1 = max(r1, (unsigned)__shfl_down(r1, o, 64));
      g0 = min(g0, (unsigned)__shfl_down(g0, o, 64)); g1 = max(g1, (unsigned)__shfl_down(g1, o, 64));
      b0 = min(b0, (unsigned)__shfl_down(b0, o, 64)); b1 = max(b1, (unsigned)__shfl_down(b1, o, 64));
      dm = max(dm, (unsigned)__shfl_down(dm, o, 64));
      ds += __shfl_down(ds, o, 64);
    }
    if (lane == 0) {
      tb.lo[b] = (r0 << 16) | (g0 << 8) | b0;
      tb.hi[b] = (r1 << 16) | (g1 << 8) | b1;
      tb.bmax[b] = dm;
      tb.bsum[b] = ds;                                  // <= 64 * 195075 fits 32 bits
    }
  }
  if (tid == 0) cho[0] = P.first;
  __syncthreads();
  const int perb = (nb + kInitThreads - 1) / kInitThreads;
  const int blo = min(tid * perb, nb), bhi = min(blo + perb, nb);
  for (int c = 1; c < k; ++c) {
    // ---- locate the block of each of the T thresholds u * pot in the cumulative sum ---------------
    unsigned long long loc = 0;
    for (int b = blo; b < bhi; ++b) loc += tb.bsum[b];
    unsigned long long pot;
    const unsigned long long base = init_exscan64(loc, sh, &pot);
    const double dpot = (double)pot;
    const double* u = rand + P.rand_off + (size_t)(c - 1) * T;
    if (tid < T) s_cblock[tid] = -1;
    __syncthreads();
    if (loc > 0) {
      for (int t = 0; t < T; ++t) {
        const double r = u[t] * dpot;
        if ((double)base < r && r <= (double)(base + loc)) {
          unsigned long long cum = base;
          int b = blo;
          for (; b < bhi - 1; ++b) {
            if ((double)(cum + tb.bsum[b]) >= r) break;
            cum += tb.bsum[b];
          }
          s_cblock[t] = b;
          s_cbase[t] = cum;
        }
      }
    }
    __syncthreads();
    // ---- one wave per candidate: in-block search (np.searchsorted 'left'), then its potential ------
    for (int t = wave; t < T; t += kInitWaves) {
      const double r = u[t] * dpot;
      const int b = s_cblock[t];
      int cand;
      if (b < 0) {
        cand = r <= 0.0 ? 0 : n - 1;
      } else {
        const int i = (b << 6) + lane;
        unsigned long long inc = i < n ? closest[i] : 0u;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const unsigned long long tv = __shfl_up(inc, o, 64);
          if (lane >= o) inc += tv;
        }
        const unsigned long long m = __ballot(i < n && (double)(s_cbase[t] + inc) >= r);
        cand = m ? (b << 6) + __ffsll((long long)m) - 1 : min((b << 6) + 63, n - 1);
      }
      const uint32_t ck = skey[cand];
      unsigned long long delta = visit_blocks<false>(ck, nb, 0, 64, skey, closest, tb);
      delta = wave_sum(delta);
      if (lane == 0) { sh.pots[t] = pot - delta; sh.cand[t] = cand; }
    }
    __syncthreads();
    int best = 0;
    unsigned long long bp = sh.pots[0];
    for (int t = 1; t < T; ++t)
      if (sh.pots[t] < bp) { bp = sh.pots[t]; best = t; }
    const int bi = sh.cand[best];
    // ---- commit the winner (all waves, disjoint block ranges) ------------------------------------------
    visit_blocks<true>(skey[bi], nb, wave * 64, kInitWaves * 64, skey, closest, tb);
    if (tid == 0) cho[c] = bi;
    __syncthreads();
  }
  __syncthreads();
  for (int j = tid; j < k; j += kInitThreads) {
    const uint32_t kk = skey[cho[j]];
    const double c0 = (double)key_r(kk), c1 = (double)key_g(kk), c2 = (double)key_b(kk);
    double* C = centres + (P.koff + j) * 4;
    C[0] = c0; C[1] = c1; C[2] = c2; C[3] = (c0 * c0 + c1 * c1) + c2 * c2;
  }
}

// ------------------------------------------------------------------------------------------------
// mini-batch steps
// ------------------------------------------------------------------------------------------------
constexpr int kBatch = 1024;        // padded batch (sklearn batch_size = 1000)
constexpr int kTileC = 64;          // centres per workgroup in the batch E-step

// state[p][8] = {ewa, ewa_min, no_improvement, since_reassign, done, steps_done, have_ewa, have_min}
__global__ __launch_bounds__(256) void mbk_batch_estep_kernel(const uint32_t* __restrict__ keys, const MbkP* __restrict__ probs,
                                                              const double* __restrict__ centres, const double* __restrict__ state,
                                                              long long step, unsigned long long seed, double* __restrict__ pdist,
                                                              int32_t* __restrict__ pidx, const long long* __restrict__ part_off) {
  const int p = blockIdx.y;
  const MbkP P = probs[p];
  if (state[p * 8 + 4] != 0.0) return;
  const int n_tiles = (int)((P.k + kTileC - 1) / kTileC);
  const int tile = blockIdx.x;
  if (tile >= n_tiles) return;
  const int bs = (int)min((long long)1000, P.n);
  __shared__ double sc[kTileC * 4];
  const int j0 = tile * kTileC, nj = (int)min((long long)kTileC, P.k - j0);
  for (int i = threadIdx.x; i < nj * 4; i += blockDim.x) {
    const double v = centres[(P.koff + j0) * 4 + i];
    sc[i] = (i & 3) == 3 ? v : -2.0 * v;               // pre-scale by -2 (exact): dist = csq + dot'
  }
  __syncthreads();
  double* pd = pdist + part_off[p] + (size_t)tile * kBatch;
  int32_t* pi = pidx + part_off[p] + (size_t)tile * kBatch;
  for (int b = threadIdx.x; b < bs; b += blockDim.x) {
    const unsigned long long src = bounded(counter_hash(seed, 2ull * (unsigned long long)step, (unsigned long long)b), (unsigned long long)P.n);
    const uint32_t kk = keys[P.off + src];
    const double x0 = (double)key_r(kk), x1 = (double)key_g(kk), x2 = (double)key_b(kk);
    double bd = sc[3] + ((x0 * sc[0] + x1 * sc[1]) + x2 * sc[2]);
    int bj = 0;
    for (int j = 1; j < nj; ++j) {
      const double d = sc[j * 4 + 3] + ((x0 * sc[j * 4] + x1 * sc[j * 4 + 1]) + x2 * sc[j * 4 + 2]);
      if (d < bd) { bd = d; bj = j; }
    }
    pd[b] = bd;
    pi[b] = j0 + bj;
  }
}

constexpr int kUpdThreads = 1024;
constexpr int kHashSlots = 2048;

struct UpdShared {
  int lab[kBatch];
  uint32_t bkey[kBatch];
  double per[kBatch];
  int hkey[kHashSlots];
  unsigned hsum[kHashSlots][4];
  unsigned long long hk[kBatch];   // reassignment hash keys
  int perm[kBatch];
  double dred[16];
  int ired[17];
  double wmax, wmin_keep, thr_w;
  int n_cand, n_re, flag;
  double sel_w;
  int sel_take;
};

__device__ __forceinline__ double block_max_d(double v, UpdShared& sh) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
  __syncthreads();
  if (lane == 0) sh.dred[w] = v;
  __syncthreads();
  double t = sh.dred[0];
  for (int i = 1; i < kUpdThreads / 64; ++i) t = fmax(t, sh.dred[i]);
  return t;
}
__device__ __forceinline__ double block_min_d(double v, UpdShared& sh) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_down(v, o, 64));
  __syncthreads();
  if (lane == 0) sh.dred[w] = v;
  __syncthreads();
  double t = sh.dred[0];
  for (int i = 1; i < kUpdThreads / 64; ++i) t = fmin(t, sh.dred[i]);
  return t;
}
__device__ __forceinline__ int block_sum_i(int v, UpdShared& sh) { return block_sum<int>(v, sh.ired); }

__global__ __launch_bounds__(kUpdThreads) void mbk_update_kernel(const uint32_t* __restrict__ keys, const MbkP* __restrict__ probs,
                                                                 double* __restrict__ centres, double* __restrict__ weights,
                                                                 double* __restrict__ state, long long step, unsigned long long seed,
                                                                 const double* __restrict__ pdist, const int32_t* __restrict__ pidx,
                                                                 const long long* __restrict__ part_off) {
  __shared__ UpdShared sh;
  const int p = blockIdx.x, tid = threadIdx.x;
  const MbkP P = probs[p];
  double* st = state + p * 8;
  if (st[4] != 0.0) return;
  const int k = (int)P.k;
  const long long n = P.n;
  const int bs = (int)min((long long)1000, n);
  const long long n_steps_max = (100 * n) / bs;
  if (step >= n_steps_max) {
    if (tid == 0) st[4] = 2.0;                           // ran out of steps
    return;
  }
  const int n_tiles = (k + kTileC - 1) / kTileC;
  double* C = centres + P.koff * 4;
  double* W = weights + P.koff;
  // ---- reduce the per-tile partial arg-mins in tile order (first arg-min) --------------------------
  for (int i = tid; i < kHashSlots; i += kUpdThreads) {
    sh.hkey[i] = -1;
    sh.hsum[i][0] = sh.hsum[i][1] = sh.hsum[i][2] = sh.hsum[i][3] = 0;
  }
  double per = 0.0;
  if (tid < bs) {
    const double* pd = pdist + part_off[p] + tid;
    const int32_t* pi = pidx + part_off[p] + tid;
    double bd = pd[0];
    int bj = pi[0];
    for (int t = 1; t < n_tiles; ++t) {
      const double d = pd[(size_t)t * kBatch];
      if (d < bd) { bd = d; bj = pi[(size_t)t * kBatch]; }
    }
    const unsigned long long src = bounded(counter_hash(seed, 2ull * (unsigned long long)step, (unsigned long long)tid), (unsigned long long)n);
    const uint32_t kk = keys[P.off + src];
    sh.lab[tid] = bj;
    sh.bkey[tid] = kk;
    const double d0 = (double)key_r(kk) - C[bj * 4], d1 = (double)key_g(kk) - C[bj * 4 + 1], d2 = (double)key_b(kk) - C[bj * 4 + 2];
    per = (d0 * d0 + d1 * d1) + d2 * d2;
  }
  if (tid < kBatch) sh.per[tid] = per;
  // reassignment decision uses the weights BEFORE this step's update (sklearn _random_reassign)
  int zero = 0;
  for (int j = tid; j < k; j += kUpdThreads) zero |= W[j] == 0.0;
  __syncthreads();
  zero = block_sum_i(zero, sh) > 0;
  double since = st[3] + (double)bs;
  const bool do_reassign = zero || since >= 10.0 * (double)k;
  if (do_reassign) since = 0.0;
  // ---- batch inertia: fixed 1024-leaf tree ----------------------------------------------------------
  for (int s = 512; s >= 1; s >>= 1) {
    __syncthreads();
    if (tid < s) sh.per[tid] = sh.per[tid] + sh.per[tid + s];
  }
  __syncthreads();
  const double inertia = sh.per[0];
  // ---- exact integer member sums per touched centre (LDS hash) -------------------------------------
  if (tid < bs) {
    const int j = sh.lab[tid];
    unsigned h = ((unsigned)j * 2654435761u) >> 21;     // 11 bits
    while (true) {
      int cur = sh.hkey[h];
      if (cur == -1) {
        const int old = atomicCAS(&sh.hkey[h], -1, j);
        cur = old == -1 ? j : old;
      }
      if (cur == j) break;
      h = (h + 1) & (kHashSlots - 1);
    }
    const uint32_t kk = sh.bkey[tid];
    atomicAdd(&sh.hsum[h][0], key_r(kk));
    atomicAdd(&sh.hsum[h][1], key_g(kk));
    atomicAdd(&sh.hsum[h][2], key_b(kk));
    atomicAdd(&sh.hsum[h][3], 1u);
  }
  __syncthreads();
  for (int h = tid; h < kHashSlots; h += kUpdThreads) {
    const int j = sh.hkey[h];
    if (j < 0) continue;
    const double w = W[j], wn = w + (double)sh.hsum[h][3];
    const double alpha = 1.0 / wn;
    const double c0 = (C[j * 4] * w + (double)sh.hsum[h][0]) * alpha;
    const double c1 = (C[j * 4 + 1] * w + (double)sh.hsum[h][1]) * alpha;
    const double c2 = (C[j * 4 + 2] * w + (double)sh.hsum[h][2]) * alpha;
    C[j * 4] = c0; C[j * 4 + 1] = c1; C[j * 4 + 2] = c2;
    C[j * 4 + 3] = (c0 * c0 + c1 * c1) + c2 * c2;
    W[j] = wn;
  }
  __syncthreads();
  // ---- low-count reassignment (sklearn _mini_batch_step) -----------------------------------------------
  if (do_reassign) {
    double wm = 0.0;
    for (int j = tid; j < k; j += kUpdThreads) wm = fmax(wm, W[j]);
    wm = block_max_d(wm, sh);
    const double thr = 0.01 * wm;
    int cnt = 0;
    for (int j = tid; j < k; j += kUpdThreads) cnt += W[j] < thr;
    cnt = block_sum_i(cnt, sh);
    const int cap = (int)(0.5 * (double)bs);
    // selection threshold: keep candidates with (W, index) among the `cap` smallest when cnt > cap
    double sel_w = thr;                                 // candidates: W < sel_w, plus `take` of W == sel_w
    int take = 0;
    if (cnt > 0.5 * (double)bs) {
      // weights are integer-valued: bisect the smallest integer v with #(W <= v, W < thr) >= cap
      double lo_v = -1.0, hi_v = floor(thr);            // #(W <= lo_v) < cap <= #(W <= hi_v) (hi_v >= all candidates)
      if (hi_v >= thr) hi_v -= 1.0;
      while (hi_v - lo_v > 1.0) {
        const double mid = floor((lo_v + hi_v) * 0.5);
        int c2 = 0;
        for (int j = tid; j < k; j += kUpdThreads) c2 += (W[j] < thr) && (W[j] <= mid);
        c2 = block_sum_i(c2, sh);
        if (c2 >= cap) hi_v = mid; else lo_v = mid;
      }
      int below = 0;
      for (int j = tid; j < k; j += kUpdThreads) below += (W[j] < thr) && (W[j] < hi_v);
      below = block_sum_i(below, sh);
      sel_w = hi_v;
      take = cap - below;
    }
    // rank the candidates in ascending index order: thread owns a contiguous index range
    const int perj = (k + kUpdThreads - 1) / kUpdThreads;
    const int jlo = min(tid * perj, k), jhi = min(jlo + perj, k);
    const bool capped = cnt > 0.5 * (double)bs;
    int eqc = 0;
    if (capped)
      for (int j = jlo; j < jhi; ++j) eqc += (W[j] < thr) && (W[j] == sel_w);
    int eq_tot;
    int eq_base = block_exscan<int>(eqc, sh.ired, &eq_tot);
    int mine = 0;
    {
      int e = eq_base;
      for (int j = jlo; j < jhi; ++j) {
        const double w = W[j];
        bool sel;
        if (!capped) sel = w < thr;
        else if (w < thr && w < sel_w) sel = true;
        else if (w < thr && w == sel_w) { sel = e < take; ++e; }
        else sel = false;
        mine += sel;
      }
    }
    int n_re;
    const int rbase = block_exscan<int>(mine, sh.ired, &n_re);
    // min weight among the centres that are NOT reassigned (computed before any overwrite)
    double wmin = INFINITY;
    {
      int e = eq_base;
      for (int j = jlo; j < jhi; ++j) {
        const double w = W[j];
        bool sel;
        if (!capped) sel = w < thr;
        else if (w < thr && w < sel_w) sel = true;
        else if (w < thr && w == sel_w) { sel = e < take; ++e; }
        else sel = false;
        if (!sel) wmin = fmin(wmin, w);
      }
    }
    wmin = block_min_d(wmin, sh);
    if (n_re > 0) {
      // perm = batch positions ordered by (hash key, position)
      if (tid < bs) sh.hk[tid] = counter_hash(seed, 2ull * (unsigned long long)step + 1ull, (unsigned long long)tid);
      __syncthreads();
      if (tid < bs) {
        const unsigned long long me = sh.hk[tid];
        int rank = 0;
        for (int b = 0; b < bs; ++b) {
          const unsigned long long o = sh.hk[b];
          rank += (o < me) || (o == me && b < tid);
        }
        sh.perm[rank] = tid;
      }
      __syncthreads();
    }
    {
      int e = eq_base, r = rbase;
      for (int j = jlo; j < jhi; ++j) {
        const double w = W[j];
        bool sel;
        if (!capped) sel = w < thr;
        else if (w < thr && w < sel_w) sel = true;
        else if (w < thr && w == sel_w) { sel = e < take; ++e; }
        else sel = false;
        if (sel) {
          const uint32_t kk = sh.bkey[sh.perm[r]];
          ++r;
          const double c0 = (double)key_r(kk), c1 = (double)key_g(kk), c2 = (double)key_b(kk);
          C[j * 4] = c0; C[j * 4 + 1] = c1; C[j * 4 + 2] = c2;
          C[j * 4 + 3] = (c0 * c0 + c1 * c1) + c2 * c2;
          W[j] = wmin;
        }
      }
    }
  }
  // ---- sklearn _mini_batch_convergence (EWA early stopping) -------------------------------------------
  if (tid == 0) {
    st[3] = since;
    st[5] = (double)(step + 1);
    const double binert = inertia / (double)bs;
    if (step + 1 != 1) {
      double ewa;
      if (st[6] == 0.0) { ewa = binert; st[6] = 1.0; }
      else {
        double a = (double)bs * 2.0 / ((double)n + 1.0);
        a = a < 1.0 ? a : 1.0;
        ewa = st[0] * (1.0 - a) + binert * a;
      }
      st[0] = ewa;
      if (st[7] == 0.0 || ewa < st[1]) { st[2] = 0.0; st[1] = ewa; st[7] = 1.0; }
      else st[2] += 1.0;
      if (st[2] >= 10.0) st[4] = 1.0;
    }
    if (step + 1 >= n_steps_max && st[4] == 0.0) st[4] = 2.0;
  }
}

// ------------------------------------------------------------------------------------------------
// final E-step over all points
// ------------------------------------------------------------------------------------------------
constexpr int kAsgThreads = 256;
constexpr int kAsgPts = 4;
constexpr int kAsgTile = 512;       // centres per LDS tile (16 KB)

__global__ __launch_bounds__(kAsgThreads) void mbk_assign_kernel(const uint32_t* __restrict__ keys, const MbkP* __restrict__ probs,
                                                                 const double* __restrict__ centres, int32_t* __restrict__ labels,
                                                                 const long long* __restrict__ blk_off) {
  // blockIdx.x -> (problem, chunk of 1024 points) through the prefix blk_off[n_prob+1]
  __shared__ double sc[kAsgTile * 4];
  int p = 0;
  while (blockIdx.x >= blk_off[p + 1]) ++p;
  const MbkP P = probs[p];
  const long long i0 = ((long long)blockIdx.x - blk_off[p]) * (kAsgThreads * kAsgPts) + threadIdx.x;
  double x0[kAsgPts], x1[kAsgPts], x2[kAsgPts], bd[kAsgPts];
  int bj[kAsgPts];
#pragma unroll
  for (int q = 0; q < kAsgPts; ++q) {
    const long long i = i0 + (long long)q * kAsgThreads;
    const uint32_t kk = i < P.n ? keys[P.off + i] : 0u;
    x0[q] = (double)key_r(kk); x1[q] = (double)key_g(kk); x2[q] = (double)key_b(kk);
    bd[q] = INFINITY; bj[q] = 0;
  }
  const int k = (int)P.k;
  for (int j0 = 0; j0 < k; j0 += kAsgTile) {
    const int nj = min(kAsgTile, k - j0);
    __syncthreads();
    for (int i = threadIdx.x; i < nj * 4; i += kAsgThreads) {
      const double v = centres[(P.koff + j0) * 4 + i];
      sc[i] = (i & 3) == 3 ? v : -2.0 * v;
    }
    __syncthreads();
    for (int j = 0; j < nj; ++j) {
      const double c0 = sc[j * 4], c1 = sc[j * 4 + 1], c2 = sc[j * 4 + 2], cs = sc[j * 4 + 3];
#pragma unroll
      for (int q = 0; q < kAsgPts; ++q) {
        const double d = cs + ((x0[q] * c0 + x1[q] * c1) + x2[q] * c2);
        if (d < bd[q]) { bd[q] = d; bj[q] = j0 + j; }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < kAsgPts; ++q) {
    const long long i = i0 + (long long)q * kAsgThreads;
    if (i < P.n) labels[P.off + i] = bj[q];
  }
}

static int ensure_scratch(rhccq_ctx* ctx, size_t bytes) {
  if (ctx->scratch_bytes >= bytes) return 0;
  if (ctx->scratch) RHCCQ_HIP(ctx, hipFree(ctx->scratch));
  ctx->scratch = nullptr;
  ctx->scratch_bytes = 0;
  RHCCQ_HIP(ctx, hipMalloc(&ctx->scratch, bytes));
  ctx->scratch_bytes = bytes;
  return 0;
}

// copies a small host table to device memory carved from `dst` (async, stream ordered; the source is
// staged in a pageable buffer so hipMemcpyAsync returns after the copy into the staging area)
static int put(rhccq_ctx* ctx, void* dst, const void* src, size_t bytes) {
  RHCCQ_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  return 0;
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace rhccq

using namespace rhccq;

extern "C" {

// work layout for steps/assign: [probs MbkP[n_prob]] [part_off i64[n_prob]] [blk_off i64[n_prob+1]]
//                               [pdist f64[sum tiles*1024]] [pidx i32[sum tiles*1024]]
int64_t rhccq_mbk_work_bytes(const rhccq_mbk_problem* probs, int32_t n_prob) {
  if (!probs || n_prob <= 0) return 0;
  size_t part = 0;
  for (int i = 0; i < n_prob; ++i) part += (size_t)((probs[i].k + kTileC - 1) / kTileC) * kBatch;
  size_t bytes = align256(sizeof(MbkP) * n_prob) + align256(8 * (size_t)n_prob) + align256(8 * (size_t)(n_prob + 1));
  bytes += align256(part * 8) + align256(part * 4);
  return (int64_t)bytes;
}

struct WorkView {
  MbkP* probs;
  long long* part_off;
  long long* blk_off;
  double* pdist;
  int32_t* pidx;
};

static int layout_work(rhccq_ctx* ctx, const rhccq_mbk_problem* probs, int n_prob, void* work, int64_t work_bytes, WorkView* v,
                       long long* total_blocks, int* max_tiles) {
  if (work_bytes < rhccq_mbk_work_bytes(probs, n_prob)) return rhccq_fail(ctx, RHCCQ_E_ARG, "mbk: work buffer too small");
  char* base = (char*)work;
  v->probs = (MbkP*)base; base += align256(sizeof(MbkP) * n_prob);
  v->part_off = (long long*)base; base += align256(8 * (size_t)n_prob);
  v->blk_off = (long long*)base; base += align256(8 * (size_t)(n_prob + 1));
  size_t part = 0;
  std::string stage;
  stage.resize(sizeof(MbkP) * n_prob + 8 * (size_t)n_prob + 8 * (size_t)(n_prob + 1));
  MbkP* hp = (MbkP*)stage.data();
  long long* hpo = (long long*)(stage.data() + sizeof(MbkP) * n_prob);
  long long* hbo = hpo + n_prob;
  long long blocks = 0;
  int mt = 0;
  for (int i = 0; i < n_prob; ++i) {
    const rhccq_mbk_problem& q = probs[i];
    if (q.n <= 0 || q.k <= 0 || q.k > q.n || q.T <= 0 || q.T > kTMaxI) return rhccq_fail(ctx, RHCCQ_E_ARG, "mbk: bad problem");
    hp[i] = MbkP{q.off, q.n, q.k, q.koff, q.init_off, q.init_n, q.rand_off, q.first, q.T};
    hpo[i] = (long long)part;
    const int tiles = (int)((q.k + kTileC - 1) / kTileC);
    mt = tiles > mt ? tiles : mt;
    part += (size_t)tiles * kBatch;
    hbo[i] = blocks;
    blocks += (q.n + kAsgThreads * kAsgPts - 1) / (kAsgThreads * kAsgPts);
  }
  hbo[n_prob] = blocks;
  v->pdist = (double*)base; base += align256(part * 8);
  v->pidx = (int32_t*)base;
  if (int e = put(ctx, v->probs, hp, sizeof(MbkP) * n_prob)) return e;
  if (int e = put(ctx, v->part_off, hpo, 8 * (size_t)n_prob)) return e;
  if (int e = put(ctx, v->blk_off, hbo, 8 * (size_t)(n_prob + 1))) return e;
  // the staging string dies at return: make sure the copies have been issued from it
  RHCCQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *total_blocks = blocks;
  *max_tiles = mt;
  return 0;
}

int rhccq_mbk_init(rhccq_ctx* ctx, const uint32_t* keys, const rhccq_mbk_problem* probs, int32_t n_prob, const int32_t* init_idx,
                   const double* rand, double* centres, int32_t* chosen) {
  if (!ctx || !keys || !probs || !init_idx || !rand || !centres || !chosen || n_prob <= 0)
    return rhccq_fail(ctx, RHCCQ_E_ARG, "mbk_init: bad argument");
  // scratch: [MbkP table][scratch_off table][per problem sample arrays]
  std::string stage;
  stage.resize((sizeof(MbkP) + 8) * (size_t)n_prob);
  MbkP* hp = (MbkP*)stage.data();
  long long* ho = (long long*)(stage.data() + sizeof(MbkP) * n_prob);
  size_t words = 0;
  for (int i = 0; i < n_prob; ++i) {
    const rhccq_mbk_problem& q = probs[i];
    if (q.init_n <= 0 || q.k <= 0 || q.k > q.init_n || q.T <= 0 || q.T > kTMaxI || q.first < 0 || q.first >= q.init_n)
      return rhccq_fail(ctx, RHCCQ_E_ARG, "mbk_init: bad problem");
    if (q.init_n > (1ll << 30)) return rhccq_fail(ctx, RHCCQ_E_LIMIT, "mbk_init: init sample too large");
    hp[i] = MbkP{q.off, q.n, q.k, q.koff, q.init_off, q.init_n, q.rand_off, q.first, q.T};
    ho[i] = (long long)words;
    const size_t nb = (size_t)((q.init_n + 63) / 64);
    words += 2 * nb * 64 + 4 * nb;
    words = (words + 63) & ~(size_t)63;
  }
  const size_t head = align256(sizeof(MbkP) * n_prob) + align256(8 * (size_t)n_prob);
  if (int e = ensure_scratch(ctx, head + words * 4)) return e;
  char* base = (char*)ctx->scratch;
  MbkP* dp = (MbkP*)base;
  long long* dof = (long long*)(base + align256(sizeof(MbkP) * n_prob));
  uint32_t* dscr = (uint32_t*)(base + head);
  if (int e = put(ctx, dp, hp, sizeof(MbkP) * n_prob)) return e;
  if (int e = put(ctx, dof, ho, 8 * (size_t)n_prob)) return e;
  RHCCQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  hipLaunchKernelGGL(mbk_init_kernel, dim3(n_prob), dim3(kInitThreads), 0, ctx->stream, keys, dp, init_idx, rand, centres, chosen, dscr, dof);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_mbk_steps(rhccq_ctx* ctx, const uint32_t* keys, const rhccq_mbk_problem* probs, int32_t n_prob, int64_t step0,
                    int32_t n_steps, uint64_t seed, double* centres, double* weights, double* state, void* work, int64_t work_bytes) {
  if (!ctx || !keys || !probs || !centres || !weights || !state || !work || n_prob <= 0 || n_steps < 0 || step0 < 0)
    return rhccq_fail(ctx, RHCCQ_E_ARG, "mbk_steps: bad argument");
  WorkView v;
  long long blocks;
  int max_tiles;
  if (int e = layout_work(ctx, probs, n_prob, work, work_bytes, &v, &blocks, &max_tiles)) return e;
  for (int s = 0; s < n_steps; ++s) {
    const long long step = step0 + s;
    hipLaunchKernelGGL(mbk_batch_estep_kernel, dim3(max_tiles, n_prob), dim3(256), 0, ctx->stream, keys, v.probs, centres, state, step,
                       (unsigned long long)seed, v.pdist, v.pidx, v.part_off);
    hipLaunchKernelGGL(mbk_update_kernel, dim3(n_prob), dim3(kUpdThreads), 0, ctx->stream, keys, v.probs, centres, weights, state, step,
                       (unsigned long long)seed, v.pdist, v.pidx, v.part_off);
  }
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_mbk_assign(rhccq_ctx* ctx, const uint32_t* keys, const rhccq_mbk_problem* probs, int32_t n_prob, const double* centres,
                     void* work, int64_t work_bytes, int32_t* labels_out) {
  if (!ctx || !keys || !probs || !centres || !work || !labels_out || n_prob <= 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "mbk_assign: bad argument");
  WorkView v;
  long long blocks;
  int max_tiles;
  if (int e = layout_work(ctx, probs, n_prob, work, work_bytes, &v, &blocks, &max_tiles)) return e;
  if (blocks > 0x7fffffffll) return rhccq_fail(ctx, RHCCQ_E_LIMIT, "mbk_assign: too many points");
  hipLaunchKernelGGL(mbk_assign_kernel, dim3((unsigned)blocks), dim3(kAsgThreads), 0, ctx->stream, keys, v.probs, centres, labels_out, v.blk_off);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

}  // extern "C"
