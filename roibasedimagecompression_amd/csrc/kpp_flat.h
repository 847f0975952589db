// Greedy k-means++ (sklearn _kmeans_plusplus, cluster/_kmeans.py) for SMALL sample sets: every sample lives in a register of
// the workgroup, a pick is two barriers.  Shared by KMeans (k7_kmeans.hip: up to 10 240 points, where it replaced a pick of six
// barriers, ~7 us -> ~3.8 us) and, on request (RHCCQ_OPT_INIT_KERNEL = 4), by the MiniBatchKMeans init (k8_minibatch.hip; see
// rhccq_mbk_init for why the block-tree chain stays the default there).
//
// Same arithmetic and the same picks as the other generations of the chain (exact integers: squared distances, potentials and the
// cumulative sums of the candidate search, so grouping and order do not matter); what differs is the schedule: no pruning, no
// lists, nothing re-read -- T distance evaluations per sample and pick, with per-candidate bookkeeping (a wave scan, a range
// test) that makes it instruction bound at ~9 500 cycles per pick: 3 000 search, 2 900 evaluate, 1 700 choose, 1 900 in barriers.
//
// One pick:
//   search    thread i owns the draw positions [i S, (i + 1) S) and knows the exclusive prefix `base` of their closest
//             distances; np.searchsorted(cumsum(closest), u_t * pot) for the T uniforms of the pick: the thread whose range
//             holds the target walks its S samples; candidate index and colour -> LDS                             (barrier 1)
//   evaluate  every thread: min(closest, d(sample, candidate t)) for its S samples and all T candidates, summed per candidate;
//             one DPP scan per candidate gives the thread's prefix inside its wave and the wave's total -> LDS     (barrier 2)
//   choose    every thread adds up the wave totals per candidate (the candidate potentials), takes the first smallest, lowers its
//             closest distances against that candidate and gets its NEXT `base` from the same wave totals: no scan of its own.
// (included inside namespace rhccq, behind rhccq_common.h)

constexpr int kFlatT = 16;          // n_local_trials = 2 + int(ln k) <= 16
constexpr int kFlatWaves = 16;      // workgroups of up to 1 024 threads

struct __align__(16) FlatShared {
  unsigned wtot[kFlatT][kFlatWaves];     // per candidate: the waves' sums of min(closest, d)
  int cand[2][kFlatT];                   // (double buffered by pick parity: thread 0 reads the winner's while the next search writes)
  uint32_t ckey[2][kFlatT];
  double u[2][kFlatT];                   // the uniforms of this pick and of the next (a cold line in HBM: fetched one pick ahead)
  unsigned long long red[kFlatWaves + 1];
};

__device__ __forceinline__ unsigned flat_incscan_u32(unsigned v) {   // inclusive scan over the wave on DPP (see wave_incscan_u32)
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
  return v;
}

// skey(i): colour of the sample at draw position i (0 <= i < n); n <= kS * blockDim.x; rand: (k - 1) * T uniforms in pick order;
// chosen[0 .. k): the draw positions picked (chosen[0] = first).  All threads of the workgroup call; ends with a barrier.
// kW: the waves that WORK (the first kW of the workgroup; the others only keep the barriers company).  A pick is bound by the instructions
// the waves of a SIMD issue together, and every working wave pays the per-candidate bookkeeping (a DPP scan, a range test, the arg-min) once
// per pick whatever its share of the samples: 3 900 samples over 4 waves x 16 samples per lane instead of 16 x 4 is the same arithmetic with
// a quarter of the bookkeeping.
template <int kS, int kW, typename KeyFn>
__device__ __forceinline__ void kpp_flat(KeyFn skey, const int n, const int k, const int T, const int first, const double* __restrict__ rand,
                                         int* __restrict__ chosen, FlatShared& sh) {
  static_assert(kW % 4 == 0 && kW <= kFlatWaves, "wave totals are read four at a time");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int nw = kW;
  const bool act = wave < kW;                                       // (wave-uniform)
  const int S = (n + 64 * kW - 1) / (64 * kW);                      // samples per thread (<= kS)
  const int i0 = tid * S;
  // per sample: the colour, |x|^2 and c' = closest - |x|^2, so that min(closest, d(x, c)) = |x|^2 + min(c', |c|^2 - 2 <x, c>):
  // a dot product, a multiply-add, a minimum and an addition per (sample, candidate) pair
  uint32_t key[kS];
  int na[kS], cp[kS];
  const uint32_t kf = skey(first);
  unsigned loc = 0, sna = 0;                                        // this thread's sum of closest distances / of |x|^2
#pragma unroll
  for (int s = 0; s < kS; ++s) {
    const int i = i0 + s;
    const bool in = act && s < S && i < n;
    key[s] = in ? skey(i) : kf;
    na[s] = (int)norm2_key(key[s]);
    const int cl = in ? dist2_keys(key[s], kf) : 0;                 // (samples beyond n: closest 0, never a candidate, never a gain)
    cp[s] = cl - na[s];
    loc += (unsigned)cl;                                            // <= 16 x 195 075
    sna += (unsigned)na[s];
  }
  // potential and the threads' exclusive prefixes of the closest distances, once by a block scan
  unsigned long long pot, base;
  {
    const unsigned inc = flat_incscan_u32(loc);                       // <= 64 x 16 x 195 075 < 2^32
    if (lane == 63 && act) sh.red[wave] = inc;
    if (tid < T && k > 1) sh.u[1][tid] = rand[tid];
    __syncthreads();
    unsigned long long before = 0, tot = 0;
    for (int w = 0; w < nw; ++w) { const unsigned long long v = sh.red[w]; if (w < wave) before += v; tot += v; }
    pot = tot;
    base = before + (inc - loc);
    if (tid == 0) chosen[0] = first;
  }
#ifdef KPP_STAMPS
  unsigned long long _ka[8] = {0, 0, 0, 0, 0, 0, 0, 0}, _kl = clock64();
#define KSTAMP(i) do { const unsigned long long _t = clock64(); _ka[i] += _t - _kl; _kl = _t; } while (0)
#else
#define KSTAMP(i) do {} while (0)
#endif
  for (int c = 1; c < k; ++c) {
    const int par = c & 1;
    if (!act) {                                                        // (the idle waves: two barriers per pick, as everybody)
      __syncthreads();
      __syncthreads();
      continue;
    }
    const double un = (tid < T && c + 1 < k) ? rand[(size_t)c * T + tid] : 0.0;     // next pick's uniforms: in flight during this pick
    const double dpot = (double)pot;
    // ---- search: np.searchsorted(cumsum(closest), u * pot, 'left') clipped to n - 1, over the draw order.  The sums are integers
    // below 2^53, so "cum >= r" is "cum >= ceil(r)": one conversion per target, integer compares for the rest ---------------------
    const unsigned long long end = base + loc;
    // the sums are integers below 2^53: exact in float64, so the range test costs two conversions per pick and two compares per target;
    // lane t forms target t once per wave
    const double rl = sh.u[par][lane & (kFlatT - 1)] * dpot;
    const double dbase = (double)base, dend = (double)end;
#pragma unroll
    for (int t = 0; t < kFlatT; ++t) {
      if (t < T) {
        const double r = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(rl), t), __builtin_amdgcn_readlane(__double2loint(rl), t));
        if (r <= 0.0) {
          if (tid == 0) { sh.cand[par][t] = 0; sh.ckey[par][t] = key[0]; }
        } else if (dbase < r && r <= dend) {                    // (then loc > 0)
          // first sample with cum >= r, i.e. cum - base >= ceil(r) - base: 32-bit sums, no chain of compares
          const unsigned rel = (unsigned)((unsigned long long)ceil(r) - base);
          unsigned run = 0;
          int hit = 0;
#pragma unroll
          for (int s = 0; s < kS; ++s) {
            run += (unsigned)(cp[s] + na[s]);
            hit += run < rel ? 1 : 0;
          }
          uint32_t hk = key[0];
#pragma unroll
          for (int s = 1; s < kS; ++s) hk = hit == s ? key[s] : hk;
          sh.cand[par][t] = i0 + hit;
          sh.ckey[par][t] = hk;
        }
      }
    }
    KSTAMP(0);
    __syncthreads();
    KSTAMP(1);
    // ---- evaluate every candidate against this thread's samples ---------------------------------------------------------------------
    unsigned acc[kFlatT], inc[kFlatT];
#pragma unroll
    for (int t = 0; t < kFlatT; ++t) {
      acc[t] = 0;
      if (t < T) {
        const uint32_t ck = sh.ckey[par][t];
        const int nb = (int)norm2_key(ck);
        int sm = 0;
#pragma unroll
        for (int s = 0; s < kS; ++s) sm += min(cp[s], nb - 2 * (int)__builtin_amdgcn_udot4(ck, key[s], 0u, false));
        acc[t] = sna + (unsigned)sm;
        inc[t] = flat_incscan_u32(acc[t]);
        if (lane == 63) sh.wtot[t][wave] = inc[t];
      }
    }
    KSTAMP(2);
    if (tid < T) sh.u[par ^ 1][tid] = un;
    KSTAMP(3);
    __syncthreads();
    KSTAMP(4);
    // ---- choose: candidate potentials from the wave totals; the first smallest wins.  Lane t adds up candidate t's totals (all
    // waves, and the waves before this one: the cross-wave part of the next prefix); the minimum is a scalar walk over T lanes ------
    unsigned ltot = 0, lbefore = 0;
    {
      const int t = lane & (kFlatT - 1);
#pragma unroll
      for (int q = 0; q < kW / 4; ++q) {
        const uint4 v = reinterpret_cast<const uint4*>(sh.wtot[t])[q];
        const unsigned vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ltot += vv[e];                                        // (<= 10 240 samples x 195 075 < 2^32)
          lbefore += (4 * q + e < wave) ? vv[e] : 0u;
        }
      }
    }
    unsigned bp = (unsigned)__builtin_amdgcn_readlane((int)ltot, 0);
    int best = 0;
#pragma unroll
    for (int t = 1; t < kFlatT; ++t) {
      if (t < T) {
        const unsigned v = (unsigned)__builtin_amdgcn_readlane((int)ltot, t);
        if (v < bp) { bp = v; best = t; }
      }
    }
    const unsigned long long bbefore = (unsigned)__shfl((int)lbefore, best, 64);
    unsigned binc = 0, bacc = 0;
#pragma unroll
    for (int t = 0; t < kFlatT; ++t)
      if (t == best) { binc = inc[t]; bacc = acc[t]; }
    const uint32_t kb = sh.ckey[par][best];
    const int nbb = (int)norm2_key(kb);
#pragma unroll
    for (int s = 0; s < kS; ++s) cp[s] = min(cp[s], nbb - 2 * (int)__builtin_amdgcn_udot4(kb, key[s], 0u, false));
    loc = bacc;
    pot = bp;
    base = bbefore + (binc - bacc);
    if (tid == 0) chosen[c] = sh.cand[par][best];
    KSTAMP(5);
  }
#ifdef KPP_STAMPS
  if (tid == 0 && blockIdx.x == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&g_init_stamps[i], _ka[i]);
#endif
#undef KSTAMP
  __syncthreads();
}

