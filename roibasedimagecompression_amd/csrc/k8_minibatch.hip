// K8: the MiniBatchKMeans branch the reference takes for palettes of >= 10 000 colours
// (encoder/compression/clustering.py:207-230; sklearn 1.7.2 MiniBatchKMeans(n_clusters, batch_size=1000,
// random_state=42, n_init='auto')).  These kernels follow sklearn's fit operation for operation -- the
// RandomState(42) stream (randint batches, choice() reassignment rows, k-means++ uniforms) is replayed on the
// device from its raw MT19937 words, k-means++ runs over the init sample in DRAW order, centre updates add the
// batch members in batch order, and np.argsort's UNSTABLE tie order in the low-count reassignment is the one
// numpy's scalar sort kernel produces (k8_npysort.h) -- so that the result equals scikit-learn's untouched
// fit_predict bit for bit under numpy's scalar sort kernels, the host setting of record (RHCCQ_OPT_REASSIGN_ORDER
// = 0: the stable order of rounds 1-3; oracle/rhccq_oracle.py::minibatch_kmeans_labels; both pinned against
// sklearn itself in tests/golden/).
//
// MI355X design
//   init   greedy k-means++ over the init sample in EXACT integers.  The chain of k picks is sequential, so one
//          1024-thread workgroup owns a problem.  Two orders of the same samples are kept: the DRAW order
//          (sklearn's: the cumulative-sum search for candidates runs over per-block sums of it) and, purely as
//          an internal pruning index, the Morton order of the colours: blocks of 64 Morton-consecutive samples
//          have a tight bounding box and carry the max of their closest distances, and a candidate only
//          visits blocks whose box is nearer than that max (exact pruning: the skipped samples cannot
//          change).  A committed improvement is scattered to the draw-order copy and its block sums.
//   steps  per step two launches for all problems: (A) brute-force E-step of the 1000-point batch
//          against all centres, centres tiled 64 per workgroup through LDS, partial arg-mins written per
//          tile; (B) one workgroup per problem reduces the partials in tile order (first arg-min),
//          accumulates exact integer member sums in an LDS hash keyed by centre, applies the per-centre
//          learning-rate update, the low-count reassignment and sklearn's EWA early-stopping rule.
//   assign final E-step over all N points, brute force, 4 points per thread, centres (pre-scaled by -2,
//          exact) tiled through LDS; float64 VALU bound (K = 3 is not an MFMA shape).
#include <hipcub/hipcub.hpp>

#include <cstring>
#include <vector>

#include "rhccq_common.h"

namespace rhccq {

// ------------------------------------------------------------------------------------------------
// shared helpers
// ------------------------------------------------------------------------------------------------
struct MbkP {  // device copy of rhccq_mbk_problem
  long long off, n, k, koff, init_off, init_n, rand_off;
  int first, T;
};

constexpr int kInitThreads = 1024;
constexpr int kInitWaves = kInitThreads / 64;
constexpr int kTMaxI = 16;   // n_local_trials = 2 + int(ln k) <= 16 for k < 1.2e6; one wave per candidate

struct InitShared {
  unsigned long long scan_red[kInitWaves + 1];
  unsigned long long red64[kInitWaves];
  unsigned long long pots[kTMaxI];
  int cand[kTMaxI];
  uint32_t ckey[kTMaxI];
  uint32_t ckp[kTMaxI][4];                                // (-k, k) channel pairs of the candidates (box tests)
  unsigned long long delta[kTMaxI];
  unsigned long long pot;
  int n_touch2[2], n_items, overflow, n_hits;
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

// Squared distance from a colour to a bounding box, exact integers, in 9 instructions: per channel the box is
// kept as the i16 pair (lo, -hi) and the colour as (-k, k); their packed sum is (lo - k, k - hi), at most one
// half of which is positive, so clamping at 0 and a packed dot product with itself adds that axis' square.
typedef short rhccq_s2 __attribute__((ext_vector_type(2)));
typedef unsigned short rhccq_us2 __attribute__((ext_vector_type(2)));
struct CandP { uint32_t r, g, b; };                       // (-k, k) pairs of one colour
__device__ __forceinline__ uint32_t box_pair(unsigned lo, unsigned hi) { return (lo & 0xffffu) | ((0u - hi) << 16); }
__device__ __forceinline__ int pair_lo(uint32_t x) { return (int)(short)(x & 0xffffu); }
__device__ __forceinline__ int pair_hi(uint32_t x) { return -(int)(short)(x >> 16); }
__device__ __forceinline__ CandP cand_pairs(uint32_t k) {
  const unsigned r = key_r(k), g = key_g(k), b = key_b(k);
  return CandP{((0u - r) & 0xffffu) | (r << 16), ((0u - g) & 0xffffu) | (g << 16), ((0u - b) & 0xffffu) | (b << 16)};
}
__device__ __forceinline__ unsigned box_axis2(uint32_t box, uint32_t kp, unsigned acc) {
  rhccq_s2 x = __builtin_bit_cast(rhccq_s2, box) + __builtin_bit_cast(rhccq_s2, kp);
  x = __builtin_elementwise_max(x, (rhccq_s2)(short)0);
  const rhccq_us2 y = __builtin_bit_cast(rhccq_us2, x);
  return __builtin_amdgcn_udot2(y, y, acc, false);
}
__device__ __forceinline__ unsigned box_dist2(const CandP& c, uint32_t xr, uint32_t xg, uint32_t xb) {
  return box_axis2(xr, c.r, box_axis2(xg, c.g, box_axis2(xb, c.b, 0u)));
}

// per problem scratch layout: samp[np] (uint2 = {key, closest}, Morton order), dsamp[np] (the same pairs in
// sklearn's draw order), mperm[np] (Morton position -> draw position) and, when the tables do not fit LDS,
// xr[nb] xg[nb] xb[nb] bmax[nb] dsum[nb] sxr[nsb] sxg[nsb] sxb[nsb] sbmax[nsb] dssum[nsb]
//   nb = ceil(init_n / 64) blocks of 64 samples, np = nb * 64, nsb = ceil(nb / 16) super-blocks
constexpr int kInitLdsBlocks = 4096;   // tables in LDS up to 262144 init samples (80 KB + 5 KB)
constexpr int kInitLdsSuper = kInitLdsBlocks / 16;

struct InitTables {
  uint32_t* xr;     // per MORTON block: bounding box, one (lo, -hi) i16 pair per channel
  uint32_t* xg;
  uint32_t* xb;
  uint32_t* bmax;   // max closest distance in the Morton block
  uint32_t* dsum;   // per DRAW block (64 consecutive draws): exact sum of closest distances
  uint32_t* sxr;    // per Morton super-block (16 blocks): box, max of bmax (may lag high)
  uint32_t* sxg;
  uint32_t* sxb;
  uint32_t* sbmax;
  uint32_t* dssum;  // per DRAW super-block (16 draw blocks): exact sum of dsum
};

#ifdef RHCCQ_STAMPS   // diagnostic build only (tools/stamps.py): per-phase cycle shares of the init chain
__device__ unsigned long long g_init_stamps[16];
__device__ unsigned long long g_wave_stamps[8][16];       // [phase][wave]: cycles each wave spent in the phase
#define STAMP(slot)                                  \
  do {                                               \
    const unsigned long long _t = clock64();        \
    _acc[slot] += _t - _last;                        \
    _last = _t;                                      \
  } while (0)
#define SUBSTART() unsigned long long _sub = clock64()
#define SUB(slot) do { const unsigned long long _t = clock64(); _acc[slot] += _t - _sub; _sub = _t; } while (0)
#else
#define STAMP(slot) do {} while (0)
#define SUBSTART() do {} while (0)
#define SUB(slot) do {} while (0)
#endif

// ---- wave helpers on DPP (VALU, no LDS crossbar): sums / max of one u32 per lane ---------------------
__device__ __forceinline__ unsigned dpp_row_sum(unsigned v) {
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);   // row_half_mirror
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);   // row_mirror
  return v;                                                                        // every lane: sum of its row of 16
}
__device__ __forceinline__ unsigned dpp_row_max(unsigned v) {
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true));
  return v;
}
// wave-uniform results (scalar): sum (< 2^32 required) / max over the 64 lanes
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
  v = dpp_row_sum(v);
  return (unsigned)__builtin_amdgcn_readlane((int)v, 0) + (unsigned)__builtin_amdgcn_readlane((int)v, 16) +
         (unsigned)__builtin_amdgcn_readlane((int)v, 32) + (unsigned)__builtin_amdgcn_readlane((int)v, 48);
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  v = dpp_row_max(v);
  return max(max((unsigned)__builtin_amdgcn_readlane((int)v, 0), (unsigned)__builtin_amdgcn_readlane((int)v, 16)),
             max((unsigned)__builtin_amdgcn_readlane((int)v, 32), (unsigned)__builtin_amdgcn_readlane((int)v, 48)));
}
// inclusive scan of one u32 per lane over the wave on DPP (gfx9 row_shr / row_bcast forms): 6 shifted adds on
// the VALU instead of 6 LDS-crossbar permutes.  The wave total must fit 32 bits.
__device__ __forceinline__ unsigned wave_incscan_u32(unsigned v) {
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);   // row_shr:1
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);   // row_shr:2
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);   // row_shr:4
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);   // row_shr:8  -> scan inside each row of 16
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
  return v;
}
// exact u64 inclusive scan of values < 2^32 from two u32 scans of 16-bit limbs (limb totals < 2^32)
__device__ __forceinline__ unsigned long long wave_incscan_limbs(unsigned v) {
  const unsigned hi = wave_incscan_u32(v >> 16), lo = wave_incscan_u32(v & 0xffffu);
  return ((unsigned long long)hi << 16) + lo;
}

__device__ __forceinline__ unsigned long long wave_incscan_u64(unsigned long long v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

// Exact two-level pruning: a (super-)block is skipped when the new centre `ck` is at least as far from its
// bounding box as the largest closest-distance inside it (then no sample of it can improve).
// enumerate_hits: one wave, all super-blocks; calls f(block) wave-uniformly for every block that may improve.
template <typename F>
__device__ __forceinline__ void enumerate_hits(const CandP& ck, int nb, int nsb, const InitTables& tb, F&& f) {
  const int lane = threadIdx.x & 63;
  for (int base = 0; base < nsb; base += 64) {
    const int sb = base + lane;
    const bool hsb = sb < nsb && box_dist2(ck, tb.sxr[sb], tb.sxg[sb], tb.sxb[sb]) < tb.sbmax[sb];
    unsigned long long msb = __ballot(hsb);
    while (msb) {
      // four hit super-blocks per round: lane -> (which super-block, which of its 16 blocks)
      int sbi[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (msb) { sbi[q] = base + __ffsll((long long)msb) - 1; msb &= msb - 1; }
        else sbi[q] = -1;
      }
      const int which = lane >> 4;
      const int my_sb = which == 0 ? sbi[0] : which == 1 ? sbi[1] : which == 2 ? sbi[2] : sbi[3];
      const int b = my_sb * 16 + (lane & 15);
      const bool hb = my_sb >= 0 && b < nb && box_dist2(ck, tb.xr[b], tb.xg[b], tb.xb[b]) < tb.bmax[b];
      f(__ballot(hb), b, hb);
    }
  }
}

// lower closest[] of Morton block b against centre ck (samples already in `sv`, their draw positions in `dpos`);
// refresh bmax and the draw-order sums; note the super-block as touched
__device__ __forceinline__ void commit_block(int b, uint32_t ck, uint2 sv, uint32_t dpos, uint2* samp, uint2* dsamp, const InitTables& tb,
                                             int* touch, int* n_touch, int touch_cap) {
  const int lane = threadIdx.x & 63;
  const unsigned d = (unsigned)dist2_keys(ck, sv.x);
  unsigned c2 = sv.y;
  if (d < c2) {
    // the improvement goes to both copies of the sample and to the draw-order sums the candidate search reads
    const unsigned delta = c2 - d;
    c2 = d;
    samp[(b << 6) + lane].y = d;
    dsamp[dpos].y = d;
    atomicSub(&tb.dsum[dpos >> 6], delta);
    atomicSub(&tb.dssum[dpos >> 10], delta);
  }
  const unsigned dm = wave_max_u32(c2);
  if (lane == 0) {
    tb.bmax[b] = dm;
    const int slot = atomicAdd(n_touch, 1);
    if (slot < touch_cap) touch[slot] = b >> 4;
  }
}

constexpr int kMaxItems = 12288;           // (candidate, block) work items per step held in LDS
constexpr int kMaxHits = kTMaxI * kInitLdsSuper;   // (candidate, super-block) pairs per step
constexpr int kMaxTouch = 1024;
constexpr int kEvalItems = 8;             // (candidate, block) items a wave keeps in flight per round of the evaluation

__device__ __forceinline__ void init_body(const uint32_t* __restrict__ keys, const MbkP& P, const int32_t* __restrict__ init_idx,
                                          const int32_t* __restrict__ perm, const double* __restrict__ rand, double* __restrict__ centres,
                                          int32_t* __restrict__ cho, uint2* samp, uint2* dsamp, uint32_t* mperm, InitTables tb, InitShared& sh, double* s_u /* [2][kTMaxI] */, int* s_touch,
                                          uint32_t* s_items /* nullptr: no work list (tables in global memory) */, uint32_t* s_hits,
                                          int max_items) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = (int)P.init_n, k = (int)P.k, T = P.T;
  const int nb = (n + 63) >> 6, np = nb << 6, nsb = (nb + 15) >> 4;
  // ---- gather the sample, first centre, block tables -----------------------------------------------
  const uint32_t kf = keys[P.off + init_idx[P.init_off + P.first]];
  for (int i = tid; i < np; i += kInitThreads) {
    // Morton position i holds the sample drawn at position d = perm[i]; padding (i >= n) repeats the last
    // Morton sample with closest = 0 and sits at the unused draw positions n .. np-1
    const int d = i < n ? perm[P.init_off + i] : i;
    const uint32_t kk = keys[P.off + init_idx[P.init_off + (i < n ? d : perm[P.init_off + n - 1])]];
    const uint2 sv = make_uint2(kk, i < n ? (unsigned)dist2_keys(kk, kf) : 0u);
    samp[i] = sv;
    dsamp[d] = sv;
    mperm[i] = (uint32_t)d;
  }
  __syncthreads();
  for (int b = wave; b < nb; b += kInitWaves) {
    const uint2 sv = samp[(b << 6) + lane];
    unsigned r0 = key_r(sv.x), r1 = r0, g0 = key_g(sv.x), g1 = g0, b0 = key_b(sv.x), b1 = b0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      r0 = min(r0, (unsigned)__shfl_down(r0, o, 64)); r1 = max(r1, (unsigned)__shfl_down(r1, o, 64));
      g0 = min(g0, (unsigned)__shfl_down(g0, o, 64)); g1 = max(g1, (unsigned)__shfl_down(g1, o, 64));
      b0 = min(b0, (unsigned)__shfl_down(b0, o, 64)); b1 = max(b1, (unsigned)__shfl_down(b1, o, 64));
    }
    const unsigned dm = wave_max_u32(sv.y);
    const unsigned ds = wave_sum_u32(dsamp[(b << 6) + lane].y);        // draw block b; <= 64 * 195075 fits 32 bits
    if (lane == 0) {
      tb.xr[b] = box_pair(r0, r1);
      tb.xg[b] = box_pair(g0, g1);
      tb.xb[b] = box_pair(b0, b1);
      tb.bmax[b] = dm;
      tb.dsum[b] = ds;
    }
  }
  __syncthreads();
  unsigned long long psum = 0;
  for (int sb = tid; sb < nsb; sb += kInitThreads) {
    int r0 = 255, g0 = 255, b0 = 255, r1 = 0, g1 = 0, b1 = 0;
    unsigned m = 0, sum = 0;
    for (int b = sb * 16; b < min(sb * 16 + 16, nb); ++b) {
      const uint32_t xr = tb.xr[b], xg = tb.xg[b], xb = tb.xb[b];
      r0 = min(r0, pair_lo(xr)); g0 = min(g0, pair_lo(xg)); b0 = min(b0, pair_lo(xb));
      r1 = max(r1, pair_hi(xr)); g1 = max(g1, pair_hi(xg)); b1 = max(b1, pair_hi(xb));
      m = max(m, tb.bmax[b]);
      sum += tb.dsum[b];                                // <= 16 * 64 * 195075 < 2^32
    }
    tb.sxr[sb] = box_pair((unsigned)r0, (unsigned)r1);
    tb.sxg[sb] = box_pair((unsigned)g0, (unsigned)g1);
    tb.sxb[sb] = box_pair((unsigned)b0, (unsigned)b1);
    tb.sbmax[sb] = m;
    tb.dssum[sb] = sum;
    psum += sum;
  }
  psum = block_sum<unsigned long long>(psum, sh.red64);
  if (tid == 0) { cho[0] = P.first; sh.n_touch2[0] = 0; sh.n_touch2[1] = 0; sh.pot = psum; sh.n_items = 0; sh.overflow = 0; sh.n_hits = 0; }
  if (tid < kTMaxI) sh.delta[tid] = 0;
  if (tid < T && k > 1) s_u[T + tid] = rand[P.rand_off + tid];         // uniforms of step 1 -> buffer 1
  __syncthreads();
  const int persb = (nsb + 63) >> 6;                     // super-blocks per lane in the level-1 search
#ifdef RHCCQ_STAMPS
  unsigned long long _acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long _last = clock64();
#endif
  for (int c = 1; c < k; ++c) {
    const double* u = s_u + (c & 1) * T;
    // the next step's uniforms are a cold line in HBM: fetch them now, park them in LDS at the end of the step
    double u_next = 0.0;
    if (tid < T && c + 1 < k) u_next = rand[P.rand_off + (size_t)c * T + tid];
    const unsigned long long pot = sh.pot;
    // the reductions of this step start from zero; reset here (between the previous step's closing barrier and
    // this step's first one) -- NOT at the end of the previous step, where slower waves may still be reading
    // them for the arg-max
    if (tid < T) sh.delta[tid] = 0;
    const int* touch_r = s_touch + (((c - 1) & 1) ? kMaxTouch : 0);   // list written by the previous step's commit
    const int n_touched = min(sh.n_touch2[(c - 1) & 1], kMaxTouch);
    // ================= phase 1: waves t < T -- pick candidate t, list the blocks it can improve ===========
    if (wave < T) {
      const int t = wave;
      const double r = u[t] * (double)pot;
      // np.searchsorted(cumsum(closest), r, 'left') over the DRAW order, through its super-block / block / sample
      // sums (all exact integers, so the grouping does not matter)
      int cand = r <= 0.0 ? 0 : n - 1;
      uint32_t ck = 0;
      bool found = false;
      {
        unsigned long long loc = 0;
        const int s0 = lane * persb;
        for (int i = 0; i < persb; ++i) loc += (s0 + i < nsb) ? tb.dssum[s0 + i] : 0u;
        // persb <= 21 super-blocks of < 2e8 each keep a lane's sum below 2^32 (kInitLdsSuper / 64 = 4 in LDS)
        const unsigned long long inc = loc < 0x100000000ull && persb <= 21 ? wave_incscan_limbs((unsigned)loc) : wave_incscan_u64(loc);
        const unsigned long long exc = inc - loc;
        const unsigned long long m1 = __ballot(loc > 0 && (double)exc < r && r <= (double)inc);
        if (m1) {
          const int src = __ffsll((long long)m1) - 1;
          unsigned long long cum = exc;
          int sb = s0;
          for (int i = 0; i < persb - 1; ++i) {          // walk inside the owning lane's chunk
            const unsigned v = (sb < nsb) ? tb.dssum[sb] : 0u;
            if ((double)(cum + v) >= r) break;
            cum += v;
            ++sb;
          }
          sb = __builtin_amdgcn_readlane(sb, src);
          cum = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(cum >> 32), src) << 32) |
                (unsigned)__builtin_amdgcn_readlane((int)(unsigned)cum, src);
          // level 2: the 16 blocks of the super-block
          const int b2 = sb * 16 + (lane & 15);
          const unsigned long long v2 = (lane < 16 && b2 < nb) ? tb.dsum[b2] : 0u;
          const unsigned long long inc2 = wave_incscan_u32((unsigned)v2);     // 16 blocks x < 1.25e7 fits 32 bits
          const unsigned long long m2 = __ballot(lane < 16 && v2 > 0 && (double)(cum + inc2 - v2) < r && r <= (double)(cum + inc2));
          if (m2) {
            const int l2 = __ffsll((long long)m2) - 1;
            const int b = sb * 16 + l2;
            const unsigned long long cum2 = cum + (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(inc2 - v2), l2);
            // level 3: the 64 samples of the block
            const int i = (b << 6) + lane;
            const uint2 sv = dsamp[i];
            const unsigned long long inc3 = wave_incscan_u32(i < n ? sv.y : 0u);   // 64 x 195075 fits 32 bits
            const unsigned long long m3 = __ballot(i < n && (double)(cum2 + inc3) >= r);
            const int l3 = m3 ? __ffsll((long long)m3) - 1 : min(63, n - 1 - (b << 6));
            cand = (b << 6) + l3;
            ck = (uint32_t)__builtin_amdgcn_readlane((int)sv.x, l3);
            found = true;
          }
        }
      }
      if (!found) ck = dsamp[cand].x;
      STAMP(0);
      if (lane == 0) {
        sh.cand[t] = cand;
        sh.ckey[t] = ck;
        const CandP cp = cand_pairs(ck);
        sh.ckp[t][0] = cp.r; sh.ckp[t][1] = cp.g; sh.ckp[t][2] = cp.b;
      }
    } else {
      // idle waves refresh the super-block maxima the previous winner touched (a stale, larger sbmax is
      // conservative, so the candidate waves may read either value)
      for (int i = (tid - T * 64); i < n_touched * 16; i += kInitThreads - T * 64) {
        const int sb = touch_r[i >> 4], b = sb * 16 + (i & 15);
        unsigned m = b < nb ? tb.bmax[b] : 0u;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
        if ((i & 15) == 0) tb.sbmax[sb] = m;
      }
    }
    if (s_items) {
      // ============ phase 1b: which blocks can each candidate improve?  All 16 waves, two balanced stages ======
      SUBSTART();
      __syncthreads();
      SUB(13);
      // stage 1: (candidate, 64 super-blocks) units -> list of (candidate, super-block) pairs that pass the box test
      const int n_chunks = (nsb + 63) >> 6;
      const float inv_chunks = 1.0f / (float)n_chunks;
      // (two units / two rounds are kept in flight per wave: each is a chain of dependent LDS accesses)
      for (int u0 = wave; u0 < T * n_chunks; u0 += 2 * kInitWaves) {
        bool hit[2];
        uint32_t word[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int u = u0 + q * kInitWaves;
          const bool uv = u < T * n_chunks;
          const int t = uv ? (int)(((float)u + 0.5f) * inv_chunks) : 0, chunk = uv ? u - t * n_chunks : 0;
          const CandP cp{sh.ckp[t][0], sh.ckp[t][1], sh.ckp[t][2]};
          const int sb = chunk * 64 + lane;
          const int sbc = min(sb, nsb - 1);
          hit[q] = uv && sb < nsb && box_dist2(cp, tb.sxr[sbc], tb.sxg[sbc], tb.sxb[sbc]) < tb.sbmax[sbc];
          word[q] = ((uint32_t)t << 24) | (uint32_t)sb;
        }
        const unsigned long long m0 = __ballot(hit[0]), m1 = __ballot(hit[1]);
        const int c0 = __popcll(m0), c1 = __popcll(m1);
        if (c0 + c1) {
          int base = 0;
          if (lane == 0) base = atomicAdd(&sh.n_hits, c0 + c1);
          base = __builtin_amdgcn_readfirstlane(base);
          const unsigned long long below = (1ull << lane) - 1ull;
          if (hit[0]) s_hits[base + __popcll(m0 & below)] = word[0];
          if (hit[1]) s_hits[base + c0 + __popcll(m1 & below)] = word[1];
        }
      }
      SUB(14);
      __syncthreads();
      SUB(15);
      // stage 2: four pairs per wave and round, one lane per block of the super-block -> (candidate, block) items
      const int n_hits = __builtin_amdgcn_readfirstlane(sh.n_hits);
      for (int h0 = wave * 4; h0 < n_hits; h0 += 2 * kInitWaves * 4) {
        bool hb[2];
        uint32_t word[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int h = h0 + q * kInitWaves * 4 + (lane >> 4);
          const uint32_t pv = h < n_hits ? s_hits[h] : 0xffffffffu;
          const int t = (int)(pv >> 24) & (kTMaxI - 1);
          const int b = (int)(pv & 0xffffffu) * 16 + (lane & 15);
          const int bc = min(b, nb - 1);
          const CandP cp{sh.ckp[t][0], sh.ckp[t][1], sh.ckp[t][2]};
          hb[q] = pv != 0xffffffffu && b < nb && box_dist2(cp, tb.xr[bc], tb.xg[bc], tb.xb[bc]) < tb.bmax[bc];
          word[q] = ((uint32_t)t << 24) | (uint32_t)b;
        }
        const unsigned long long m0 = __ballot(hb[0]), m1 = __ballot(hb[1]);
        const int c0 = __popcll(m0), c1 = __popcll(m1);
        if (c0 + c1) {
          int base = 0;
          if (lane == 0) base = atomicAdd(&sh.n_items, c0 + c1);
          base = __builtin_amdgcn_readfirstlane(base);
          if (base + c0 + c1 > max_items) { if (lane == 0) sh.overflow = 1; }
          else {
            const unsigned long long below = (1ull << lane) - 1ull;
            if (hb[0]) s_items[base + __popcll(m0 & below)] = word[0];
            if (hb[1]) s_items[base + c0 + __popcll(m1 & below)] = word[1];
          }
        }
      }
    }
    STAMP(1);
    __syncthreads();
    STAMP(2);
    const bool use_list = s_items != nullptr && sh.overflow == 0;
    const int n_items = sh.n_items;
#ifdef RHCCQ_STAMPS
    _acc[5] += (unsigned long long)n_items;
    _acc[7] += n_items > 128 ? 1 : 0;
#endif
    int* touch_w = s_touch + ((c & 1) ? kMaxTouch : 0);      // list written by this step's commit
    int* n_touch_w = &sh.n_touch2[c & 1];
    // ================= phase 2: potentials ==================================================================
    // all waves share the (candidate, block) items evenly, kEvalItems (x 512 B of samples) in flight per wave
    if (use_list) {
      // Items, candidates and block numbers are wave-uniform: they are moved to scalar registers (readlane) so
      // that the per-item work is a handful of vector instructions -- one CU's four SIMDs issue for all 16
      // waves, and the instruction count, not memory latency, sets the length of this phase.
      const uint32_t ckv = lane < T ? sh.ckey[lane] : 0u;              // lane t: key of candidate t
      const unsigned cnv = norm2_key(ckv);
      const int n_it = __builtin_amdgcn_readfirstlane(n_items);
      SUBSTART();
      for (int i0 = wave * kEvalItems; i0 < n_it; i0 += kInitWaves * kEvalItems) {
        const uint32_t itv = (lane < kEvalItems && i0 + lane < n_it) ? s_items[i0 + lane] : 0xffffffffu;
        uint2 sv[kEvalItems];
#pragma unroll
        for (int q = 0; q < kEvalItems; ++q) {
          const uint32_t it = (uint32_t)__builtin_amdgcn_readlane((int)itv, q);
          sv[q] = samp[((it == 0xffffffffu ? 0u : (it & 0xffffffu)) << 6) + lane];
        }
        SUB(8);
        // consecutive items mostly belong to one candidate (a candidate appends its blocks in runs): keep a
        // per-lane partial sum while the candidate does not change, reduce across the wave only on a change
        int cur_t = -1;
        unsigned acc = 0;                                // <= kEvalItems * 195075 per lane
#pragma unroll
        for (int q = 0; q < kEvalItems; ++q) {
          const uint32_t it = (uint32_t)__builtin_amdgcn_readlane((int)itv, q);
          if (it == 0xffffffffu) break;                  // items are dense: the first gap ends the batch
          const int t = (int)(it >> 24);
          if (t != cur_t) {
            if (cur_t >= 0) {
              const unsigned sdel = wave_sum_u32(acc);
              if (lane == 0 && sdel) atomicAdd(&sh.delta[cur_t], (unsigned long long)sdel);
            }
            cur_t = t;
            acc = 0;
          }
          const uint32_t ck = (uint32_t)__builtin_amdgcn_readlane((int)ckv, t);
          const unsigned na = (unsigned)__builtin_amdgcn_readlane((int)cnv, t);
          // closest - d, clamped at 0, with d = na + |x|^2 - 2 <ck, x>
          const unsigned nbna = __builtin_amdgcn_udot4(sv[q].x, sv[q].x, na, false);
          const unsigned c2d = sv[q].y + 2u * __builtin_amdgcn_udot4(ck, sv[q].x, 0u, false);
          acc += c2d > nbna ? c2d - nbna : 0u;
        }
        if (cur_t >= 0) {
          const unsigned sdel = wave_sum_u32(acc);
          if (lane == 0 && sdel) atomicAdd(&sh.delta[cur_t], (unsigned long long)sdel);
        }
        SUB(9);
      }
    } else {
      for (int t = wave; t < T; t += kInitWaves) {
        const uint32_t ck = sh.ckey[t];
        unsigned long long delta = 0;
        enumerate_hits(cand_pairs(ck), nb, nsb, tb, [&](unsigned long long mb, int b, bool hb) {
          while (mb) {
            const int p = __ffsll((long long)mb) - 1;
            mb &= mb - 1;
            const int bb = __shfl(b, p, 64);
            const uint2 s2 = samp[(bb << 6) + lane];
            const unsigned d = (unsigned)dist2_keys(ck, s2.x);
            delta += s2.y > d ? s2.y - d : 0u;
          }
        });
        delta = wave_sum(delta);
        if (lane == 0) sh.delta[t] = delta;
      }
    }
    __syncthreads();
    STAMP(3);
    // ================= phase 3: greedy choice + commit ======================================================
#ifdef RHCCQ_STAMPS
    const unsigned long long _am0 = clock64();
#endif
    // largest reduction == smallest potential; the first candidate wins ties.  Lane t holds delta[t]: the
    // wave maximum of the high words, then of the low words among the lanes that share it, then the first lane
    // (a serial loop over T LDS reads was 10 % of the step)
    const unsigned long long dv = lane < T ? sh.delta[lane] : 0ull;
    const unsigned dhi = (unsigned)(dv >> 32), dlo = (unsigned)dv;
    const unsigned mhi = wave_max_u32(dhi);
    const unsigned mlo = wave_max_u32(dhi == mhi ? dlo : 0u);
    const unsigned long long bd = ((unsigned long long)mhi << 32) | mlo;
    const int best = __ffsll((long long)__ballot(lane < T && dv == bd)) - 1;
    const uint32_t kb = (uint32_t)__builtin_amdgcn_readlane((int)(lane < T ? sh.ckey[lane] : 0u), best);
#ifdef RHCCQ_STAMPS
    _acc[10] += clock64() - _am0 + (kb & 0);
#endif
    SUBSTART();
    if (false) {
    } else if (use_list) {
      // lane l of wave w looks at item w + 16 l (one LDS read covers 1024 items); the wave then commits the
      // winner's blocks among them one after the other
      for (int i0 = 0; i0 < n_items; i0 += kInitWaves * 64) {
        const int i = i0 + wave + kInitWaves * lane;
        const uint32_t itx = i < n_items ? s_items[i] : 0xffffffffu;
        unsigned long long m = __ballot(itx != 0xffffffffu && (int)(itx >> 24) == best);
        while (m) {
          const int l = __ffsll((long long)m) - 1;
          m &= m - 1;
          const int b = (int)((uint32_t)__shfl((int)itx, l, 64) & 0xffffffu);
          commit_block(b, kb, samp[(b << 6) + lane], mperm[(b << 6) + lane], samp, dsamp, tb, touch_w, n_touch_w, kMaxTouch);
        }
      }
    } else {
      // wave w owns super-blocks sb = w (mod 16)
      const CandP kbp = cand_pairs(kb);
      for (int base = 0; base * kInitWaves + wave < nsb; base += 64) {
        const int sb = (base + lane) * kInitWaves + wave;
        const bool hsb = sb < nsb && box_dist2(kbp, tb.sxr[sb], tb.sxg[sb], tb.sxb[sb]) < tb.sbmax[sb];
        unsigned long long msb = __ballot(hsb);
        while (msb) {
          const int sx = (base + __ffsll((long long)msb) - 1) * kInitWaves + wave;
          msb &= msb - 1;
          const int b = sx * 16 + (lane & 15);
          const bool hb = lane < 16 && b < nb && box_dist2(kbp, tb.xr[b], tb.xg[b], tb.xb[b]) < tb.bmax[b];
          unsigned long long mb = __ballot(hb);
          while (mb) {
            const int p = __ffsll((long long)mb) - 1;
            mb &= mb - 1;
            const int bb = sx * 16 + p;
            commit_block(bb, kb, samp[(bb << 6) + lane], mperm[(bb << 6) + lane], samp, dsamp, tb, touch_w, n_touch_w, kMaxTouch);
          }
        }
      }
    }
    SUB(11);
#ifdef RHCCQ_STAMPS
    __syncthreads();
    _acc[6] += (unsigned long long)*n_touch_w;
#endif
    SUB(12);
    if (tid == 0) { cho[c] = sh.cand[best]; sh.pot = pot - bd; sh.n_items = 0; sh.overflow = 0; sh.n_hits = 0; sh.n_touch2[(c + 1) & 1] = 0; }
    if (tid < T) s_u[((c + 1) & 1) * T + tid] = u_next;
    __syncthreads();
    STAMP(4);
    // more touched super-blocks than the list holds (only in the first steps): refresh all of them
    if (sh.n_touch2[c & 1] > kMaxTouch) {
      for (int sb = tid; sb < nsb; sb += kInitThreads) {
        unsigned m = 0;
        for (int b = sb * 16; b < min(sb * 16 + 16, nb); ++b) m = max(m, tb.bmax[b]);
        tb.sbmax[sb] = m;
      }
      __syncthreads();
      if (tid == 0) sh.n_touch2[c & 1] = 0;
      __syncthreads();
    }
  }
#ifdef RHCCQ_STAMPS
  if (tid == 0 && blockIdx.x == gridDim.x - 1)
    for (int i = 0; i < 16; ++i) g_init_stamps[i] += _acc[i];
#endif
  for (int j = tid; j < k; j += kInitThreads) {
    const uint32_t kk = dsamp[cho[j]].x;
    const double c0 = (double)key_r(kk), c1 = (double)key_g(kk), c2 = (double)key_b(kk);
    double* C = centres + (P.koff + j) * 4;
    C[0] = c0; C[1] = c1; C[2] = c2; C[3] = km64_csq(c0, c1, c2);
  }
}

__global__ __launch_bounds__(kInitThreads) void mbk_init_kernel(const uint32_t* __restrict__ keys, const MbkP* __restrict__ probs,
                                                                const int32_t* __restrict__ init_idx, const int32_t* __restrict__ perm,
                                                                const double* __restrict__ rand, double* __restrict__ centres,
                                                                int32_t* __restrict__ chosen,
                                                                uint32_t* scratch, const long long* __restrict__ scratch_off,
                                                                int lds_blocks, int max_items) {
  __shared__ InitShared sh;
  __shared__ uint32_t s_tab[5 * kInitLdsBlocks + 5 * kInitLdsSuper];
  __shared__ uint32_t s_items[kMaxItems];
  __shared__ uint32_t s_hits[kMaxHits];
  __shared__ double s_u[2 * kTMaxI];
  __shared__ int s_touch[2 * kMaxTouch];
  const MbkP P = probs[blockIdx.x];
  const int nb = ((int)P.init_n + 63) >> 6, np = nb << 6, nsb = (nb + 15) >> 4;
  uint2* samp = reinterpret_cast<uint2*>(scratch + scratch_off[blockIdx.x]);
  uint2* dsamp = samp + np;
  uint32_t* mperm = reinterpret_cast<uint32_t*>(dsamp + np);
  InitTables tb;
  if (nb <= lds_blocks) {
    tb.xr = s_tab; tb.xg = s_tab + kInitLdsBlocks; tb.xb = s_tab + 2 * kInitLdsBlocks;
    tb.bmax = s_tab + 3 * kInitLdsBlocks; tb.dsum = s_tab + 4 * kInitLdsBlocks;
    tb.sxr = s_tab + 5 * kInitLdsBlocks; tb.sxg = tb.sxr + kInitLdsSuper; tb.sxb = tb.sxg + kInitLdsSuper;
    tb.sbmax = tb.sxb + kInitLdsSuper; tb.dssum = tb.sbmax + kInitLdsSuper;
    init_body(keys, P, init_idx, perm, rand, centres, chosen + P.koff, samp, dsamp, mperm, tb, sh, s_u, s_touch, s_items, s_hits, max_items);
  } else {
    uint32_t* g = mperm + np;
    tb.xr = g; tb.xg = g + nb; tb.xb = g + 2 * nb; tb.bmax = g + 3 * nb; tb.dsum = g + 4 * nb;
    tb.sxr = g + 5 * nb; tb.sxg = tb.sxr + nsb; tb.sxb = tb.sxg + nsb; tb.sbmax = tb.sxb + nsb; tb.dssum = tb.sbmax + nsb;
    init_body(keys, P, init_idx, perm, rand, centres, chosen + P.koff, samp, dsamp, mperm, tb, sh, s_u, s_touch, nullptr, nullptr, 0);
  }
}

// =====================================================================================================
// k-means++ chain, second generation.  Same arithmetic and the same picks as init_body() above (both are exact
// integers); what changes is how one pick is scheduled on the CU, because the chain is bound by instruction issue:
//   * enumeration in ONE phase: a (candidate, 64 super-blocks) unit tests its boxes and expands its hits to blocks
//     on the spot, appending (candidate, block) items to the shared list -- two barriers and one list less;
//   * evaluation by quarter waves: 16 lanes x 4 samples per item, 4 items per wave instruction stream (half the
//     instructions per item); the stored per-sample value is c' = closest - |x|^2, so that the improvement
//     closest - d = c' - |c|^2 + 2<c, x> costs one dot product;
//   * the samples a wave evaluated STAY in its registers until the winner is known: the commit needs no second
//     trip to the L2, and the draw position of every sample travels in the spare bits of its (key, c') pair.
// =====================================================================================================
constexpr int kJThreads = 1024;
constexpr int kJWaves = kJThreads / 64;
constexpr int kJMaxItems = 4096;          // (candidate, block) items per pick in LDS; beyond: per-candidate enumeration
constexpr int kJKeep = 6;                 // evaluation instructions (4 items each) a wave keeps in registers
constexpr int kJTouch = 1024;

struct JShared {
  unsigned long long delta[kTMaxI];
  unsigned long long R[kTMaxI];           // integer search targets: ceil(u * pot)
  unsigned long long pot;
  unsigned long long red64[kJWaves];
  int cand[kTMaxI];
  uint32_t ckey[kTMaxI];
  uint32_t cna[kTMaxI];
  uint32_t ckp[kTMaxI][4];
  int n_items, overflow, n_touch2[2], n_hits;
  // sharded chain only
  unsigned long long tot[8];              // every shard's sum of closest (exact)
  unsigned long long gd[8][kTMaxI];       // every shard's improvement per candidate of this pick
  int owner[kTMaxI];
  int abort;
};

// ---- sharded chain: C workgroups (one CU each) per problem; shard s owns the draws [s D, (s + 1) D) with its own Morton
// index.  Per pick two exchanges through 8-byte data-tagged granules (tag = pick number, agent-scope relaxed = sc1 accesses):
// the owner of a search target publishes the candidate's colour; every shard publishes its T partial improvements.
constexpr int kJMaxShards = 8;
constexpr unsigned kJTagInit = 0xffffffu;
struct JExchange {                        // global memory, zeroed before every launch
  unsigned long long cand[2][kTMaxI];
  unsigned long long delta[2][kJMaxShards][kTMaxI];
  unsigned long long tot[kJMaxShards];
  unsigned long long abort;
  unsigned long long pad;
};
__device__ __forceinline__ unsigned long long jx_load(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void jx_store(unsigned long long* p, unsigned tag, unsigned long long value) {
  __hip_atomic_store(p, ((unsigned long long)tag << 40) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every lane with `on` re-reads its granule until the tags of all of them match; false = gave up (a partner aborted or
// never arrived): the caller raises the abort words and every shard leaves the chain at its next barrier
__device__ __forceinline__ bool jx_wait(const unsigned long long* p, bool on, unsigned tag, const unsigned long long* abort_word,
                                        unsigned long long& value) {
  for (unsigned spins = 0;; ++spins) {
    unsigned long long x = 0;
    if (on) x = jx_load(p);
    const bool ok = !on || (unsigned)(x >> 40) == tag;
    if (__all(ok)) { value = x & ((1ull << 40) - 1ull); return true; }
    if ((spins & 255u) == 255u && (spins > (1u << 21) || jx_load(abort_word) != 0)) return false;
  }
}

// sample pair, Morton order: x = key | (dpos & 255) << 24 ; y = (c' & 0x7ffff) | (dpos >> 8) << 19, c' = closest - |key|^2
__device__ __forceinline__ uint2 jpack(uint32_t key, int cprime, uint32_t dpos) {
  return make_uint2(key | (dpos << 24), ((uint32_t)cprime & 0x7ffffu) | ((dpos >> 8) << 19));
}
__device__ __forceinline__ int j_cprime(uint32_t y) { return (int)(y << 13) >> 13; }
__device__ __forceinline__ uint32_t j_dpos(uint32_t x, uint32_t y) { return (x >> 24) | ((y >> 19) << 8); }

__device__ __forceinline__ unsigned row_incscan_u32(unsigned v) {                 // inclusive scan inside each row of 16 lanes
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
  return v;
}
__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int l) {
  return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
}
template <typename T>
__device__ __forceinline__ T sel4(int q, const T (&a)[4]) { return q == 0 ? a[0] : q == 1 ? a[1] : q == 2 ? a[2] : a[3]; }

// improvement of the 4 samples a lane holds against candidate (ck, na = |ck|^2): sum of max(0, c' - na + 2 <ck, x>)
__device__ __forceinline__ unsigned j_eval4(uint32_t ck, int na, const uint4& a, const uint4& b) {
  const int i0 = j_cprime(a.y) - na + 2 * (int)__builtin_amdgcn_udot4(ck, a.x, 0u, false);
  const int i1 = j_cprime(a.w) - na + 2 * (int)__builtin_amdgcn_udot4(ck, a.z, 0u, false);
  const int i2 = j_cprime(b.y) - na + 2 * (int)__builtin_amdgcn_udot4(ck, b.x, 0u, false);
  const int i3 = j_cprime(b.w) - na + 2 * (int)__builtin_amdgcn_udot4(ck, b.z, 0u, false);
  return (unsigned)(max(i0, 0) + max(i1, 0)) + (unsigned)(max(i2, 0) + max(i3, 0));
}

// commit of one improved sample (rare: ~30 per pick): both copies of the sample, the draw-order sums
__device__ __forceinline__ void j_store1(int imp, int newcp, unsigned nx, uint32_t x, uint32_t y, int m, uint2* samp, uint2* dsamp, uint32_t* dsum,
                                         uint32_t* dssum) {
  if (imp > 0) {
    const uint32_t dpos = j_dpos(x, y);
    samp[m].y = ((uint32_t)newcp & 0x7ffffu) | (y & 0xfff80000u);
    dsamp[dpos].y = (uint32_t)(newcp + (int)nx);
    atomicSub(&dsum[dpos >> 6], (unsigned)imp);
    atomicSub(&dssum[dpos >> 10], (unsigned)imp);
  }
}

// one row of 16 lanes = one block of 64 samples (lane j holds samples 4j .. 4j+3 in a, b): lower closest[] against
// the new centre, refresh the block's max, note its super-block as touched.  Everything but the stores is computed
// unconditionally (straight-line code: the chain is bound by instruction issue, and divergent branches cost more than
// the few operations they would skip).
__device__ __forceinline__ void j_commit_row(bool on, int b, uint32_t ck, int na, const uint4& a, const uint4& bb, uint2* samp, uint2* dsamp,
                                             uint4* blk, uint32_t* dsum, uint32_t* dssum, int* touch, int* n_touch) {
  const int j = threadIdx.x & 15;
  const int e0 = 2 * (int)__builtin_amdgcn_udot4(ck, a.x, 0u, false), e1 = 2 * (int)__builtin_amdgcn_udot4(ck, a.z, 0u, false);
  const int e2 = 2 * (int)__builtin_amdgcn_udot4(ck, bb.x, 0u, false), e3 = 2 * (int)__builtin_amdgcn_udot4(ck, bb.z, 0u, false);
  const int c0 = j_cprime(a.y), c1 = j_cprime(a.w), c2 = j_cprime(bb.y), c3 = j_cprime(bb.w);
  const int i0 = c0 - na + e0, i1 = c1 - na + e1, i2 = c2 - na + e2, i3 = c3 - na + e3;       // improvements (> 0: the sample moves)
  const unsigned n0 = norm2_key(a.x & 0xffffffu), n1 = norm2_key(a.z & 0xffffffu), n2 = norm2_key(bb.x & 0xffffffu), n3 = norm2_key(bb.z & 0xffffffu);
  const int w0 = i0 > 0 ? na - e0 : c0, w1 = i1 > 0 ? na - e1 : c1, w2 = i2 > 0 ? na - e2 : c2, w3 = i3 > 0 ? na - e3 : c3;   // new c' = d - |x|^2
  if (on && max(max(i0, i1), max(i2, i3)) > 0) {
    const int m0 = (b << 6) + 4 * j;
    j_store1(i0, w0, n0, a.x, a.y, m0, samp, dsamp, dsum, dssum);
    j_store1(i1, w1, n1, a.z, a.w, m0 + 1, samp, dsamp, dsum, dssum);
    j_store1(i2, w2, n2, bb.x, bb.y, m0 + 2, samp, dsamp, dsum, dssum);
    j_store1(i3, w3, n3, bb.z, bb.w, m0 + 3, samp, dsamp, dsum, dssum);
  }
  unsigned mx = max(max((unsigned)(w0 + (int)n0), (unsigned)(w1 + (int)n1)), max((unsigned)(w2 + (int)n2), (unsigned)(w3 + (int)n3)));
  mx = dpp_row_max(on ? mx : 0u);
  if (on && j == 0) {
    blk[b].w = mx;
    const int slot = atomicAdd(n_touch, 1);
    if (slot < kJTouch) touch[slot] = b >> 4;
  }
}

// calls f(my_block, on) for the blocks named by the set bits of `mb` (bit l <-> the block lane l holds in `b`), four
// blocks per call: row q of the wave gets the q-th of them
template <typename F>
__device__ __forceinline__ void j_for_rows(unsigned long long mb, int b, F&& f) {
  const int q = (threadIdx.x & 63) >> 4;
  while (mb) {
    int b4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (mb) { b4[i] = __builtin_amdgcn_readlane(b, __ffsll((long long)mb) - 1); mb &= mb - 1; }
      else b4[i] = -1;
    }
    const int my_b = sel4(q, b4);
    f(my_b, my_b >= 0);
  }
}

// box tests of one candidate against 64 super-blocks (one per lane), hits expanded to blocks: calls g(mask, b) with the
// ballot of the blocks that may improve and the block index each lane tested
template <typename G>
__device__ __forceinline__ void j_enumerate(const CandP& cp, int sb_first, int nsb, int nb, const uint4* sup, const uint4* blk, G&& g) {
  const int lane = threadIdx.x & 63;
  const int sbi = sb_first + lane;
  bool hs = false;
  if (sbi < nsb) {
    const uint4 se = sup[sbi];
    hs = box_dist2(cp, se.x, se.y, se.z) < se.w;
  }
  unsigned long long ms = __ballot(hs);
  while (ms) {
    int s4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (ms) { s4[i] = sb_first + __ffsll((long long)ms) - 1; ms &= ms - 1; }
      else s4[i] = -1;
    }
    const int my_s = sel4(lane >> 4, s4);
    const int b = my_s * 16 + (lane & 15);
    bool hb = my_s >= 0 && b < nb;
    if (hb) {
      const uint4 be = blk[b];
      hb = box_dist2(cp, be.x, be.y, be.z) < be.w;
    }
    g(__ballot(hb), b, hb);
  }
}

template <bool kSh>
__global__ __launch_bounds__(kJThreads) void mbk_init2_kernel(const uint32_t* __restrict__ keys, const MbkP* __restrict__ probs,
                                                               const int32_t* __restrict__ init_idx, const int32_t* __restrict__ perm,
                                                               const double* __restrict__ rand, double* __restrict__ centres,
                                                               int32_t* __restrict__ chosen, uint32_t* scratch,
                                                               const long long* __restrict__ scratch_off, int max_items, int nshard,
                                                               JExchange* xch_all) {
  __shared__ JShared sh;
  __shared__ uint4 blk[kInitLdsBlocks];                  // per Morton block: box (3 pairs), max closest
  __shared__ uint4 sup[kInitLdsSuper];                   // per Morton super-block: box, max of the blocks' max (may lag high)
  __shared__ uint32_t dsum[kInitLdsBlocks];              // per draw block: sum of closest
  __shared__ uint32_t dssum[kInitLdsSuper];
  __shared__ uint32_t items[kJMaxItems];
  __shared__ uint32_t hits[kTMaxI * kInitLdsSuper];      // (candidate, super-block) pairs whose box test passed
  __shared__ int s_touch[2 * kJTouch];
  const int prob = kSh ? (int)blockIdx.x / nshard : (int)blockIdx.x;
  const int me = kSh ? (int)blockIdx.x % nshard : 0;
  const MbkP P = probs[prob];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform BY CONSTRUCTION: loops and branches on it stay scalar
  const int n_all = (int)P.init_n, k = (int)P.k, T = P.T;
  // shard `me` owns the draws [lo, lo + n): whole draw super-blocks, so that its sums are the global ones restricted to it
  const int shard_d = kSh ? ((((n_all + nshard - 1) / nshard) + 1023) & ~1023) : n_all;
  const int lo = kSh ? me * shard_d : 0;
  const int n = kSh ? min(shard_d, n_all - lo) : n_all;          // (the launcher makes sure every shard is non-empty)
  const int nb = (n + 63) >> 6, np = nb << 6, nsb = (nb + 15) >> 4;
  uint2* samp = reinterpret_cast<uint2*>(scratch + scratch_off[blockIdx.x]);
  uint2* dsamp = samp + np;
  int32_t* cho = chosen + P.koff;
  JExchange* xch = kSh ? xch_all + prob : nullptr;
  // ---- gather the sample (both orders), first centre, tables ---------------------------------------------
  const uint32_t kf = keys[P.off + init_idx[P.init_off + P.first]];
  if (!kSh) {
    const int last_d = perm[P.init_off + n - 1];
    for (int i = tid; i < np; i += kJThreads) {
      const int d = i < n ? perm[P.init_off + i] : i;     // padding: the last Morton sample again, closest = 0, unused draw slots
      const uint32_t kk = keys[P.off + init_idx[P.init_off + (i < n ? d : last_d)]];
      const unsigned cl = i < n ? (unsigned)dist2_keys(kk, kf) : 0u;
      samp[i] = jpack(kk, (int)cl - (int)norm2_key(kk), (uint32_t)d);
      dsamp[d] = make_uint2(kk, cl);
    }
  } else {
    // the problem's Morton order, filtered to this shard's draws (a stable compaction keeps the order)
    __shared__ int s_cnt[kJWaves + 1];
    int base = 0;
    uint32_t last_key = 0;
    for (int i0 = 0; i0 < n_all; i0 += kJThreads) {
      const int i = i0 + tid;
      const int d = i < n_all ? perm[P.init_off + i] : -1;
      const bool in = d >= lo && d < lo + n;
      const unsigned long long mb = __ballot(in);
      if (lane == 0) s_cnt[wave] = __popcll(mb);
      __syncthreads();
      int before = 0, total = 0;
      for (int w = 0; w < kJWaves; ++w) {
        const int cw = s_cnt[w];
        before += w < wave ? cw : 0;
        total += cw;
      }
      if (in) {
        const uint32_t kk = keys[P.off + init_idx[P.init_off + d]];
        const unsigned cl = (unsigned)dist2_keys(kk, kf);
        const int m = base + before + __popcll(mb & ((1ull << lane) - 1ull));
        samp[m] = jpack(kk, (int)cl - (int)norm2_key(kk), (uint32_t)(d - lo));
        dsamp[d - lo] = make_uint2(kk, cl);
        if (m == n - 1) s_cnt[kJWaves] = (int)kk;
      }
      base += total;
      __syncthreads();
    }
    last_key = (uint32_t)s_cnt[kJWaves];
    for (int i = n + tid; i < np; i += kJThreads) {       // padding as above
      samp[i] = jpack(last_key, -(int)norm2_key(last_key), (uint32_t)i);
      dsamp[i] = make_uint2(last_key, 0u);
    }
  }
  __syncthreads();
  for (int b = wave; b < nb; b += kJWaves) {
    const uint2 sv = samp[(b << 6) + lane];
    const uint32_t kk = sv.x & 0xffffffu;
    unsigned r0 = key_r(kk), r1 = r0, g0 = key_g(kk), g1 = g0, b0 = key_b(kk), b1 = b0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      r0 = min(r0, (unsigned)__shfl_down(r0, o, 64)); r1 = max(r1, (unsigned)__shfl_down(r1, o, 64));
      g0 = min(g0, (unsigned)__shfl_down(g0, o, 64)); g1 = max(g1, (unsigned)__shfl_down(g1, o, 64));
      b0 = min(b0, (unsigned)__shfl_down(b0, o, 64)); b1 = max(b1, (unsigned)__shfl_down(b1, o, 64));
    }
    const unsigned dm = wave_max_u32((unsigned)(j_cprime(sv.y) + (int)norm2_key(kk)));
    const unsigned ds = wave_sum_u32(dsamp[(b << 6) + lane].y);
    if (lane == 0) {
      blk[b] = make_uint4(box_pair(r0, r1), box_pair(g0, g1), box_pair(b0, b1), dm);
      dsum[b] = ds;
    }
  }
  __syncthreads();
  unsigned long long psum = 0;
  for (int sb = tid; sb < nsb; sb += kJThreads) {
    int r0 = 255, g0 = 255, b0 = 255, r1 = 0, g1 = 0, b1 = 0;
    unsigned m = 0, sum = 0;
    for (int b = sb * 16; b < min(sb * 16 + 16, nb); ++b) {
      const uint4 be = blk[b];
      r0 = min(r0, pair_lo(be.x)); g0 = min(g0, pair_lo(be.y)); b0 = min(b0, pair_lo(be.z));
      r1 = max(r1, pair_hi(be.x)); g1 = max(g1, pair_hi(be.y)); b1 = max(b1, pair_hi(be.z));
      m = max(m, be.w);
      sum += dsum[b];
    }
    sup[sb] = make_uint4(box_pair((unsigned)r0, (unsigned)r1), box_pair((unsigned)g0, (unsigned)g1), box_pair((unsigned)b0, (unsigned)b1), m);
    dssum[sb] = sum;
    psum += sum;
  }
  psum = block_sum<unsigned long long>(psum, sh.red64);
  if (kSh) {
    if (tid == 0) { sh.abort = 0; jx_store(&xch->tot[me], kJTagInit, psum); }
    if (wave == 0) {
      unsigned long long v = 0;
      const bool ok = jx_wait(&xch->tot[lane < nshard ? lane : 0], lane < nshard, kJTagInit, &xch->abort, v);
      if (!ok && lane == 0) { sh.abort = 1; __hip_atomic_store(&xch->abort, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      if (lane < nshard) sh.tot[lane] = v;
    }
    __syncthreads();
    psum = 0;
    for (int s2 = 0; s2 < nshard; ++s2) psum += sh.tot[s2];
    if (tid == 0 && me == 0) {
      const double c0 = (double)key_r(kf), c1 = (double)key_g(kf), c2 = (double)key_b(kf);
      double* C = centres + P.koff * 4;
      C[0] = c0; C[1] = c1; C[2] = c2; C[3] = km64_csq(c0, c1, c2);
    }
  }
  if (tid == 0) { if (me == 0) cho[0] = P.first; sh.n_touch2[0] = 0; sh.n_touch2[1] = 0; sh.pot = psum; sh.n_items = 0; sh.n_hits = 0; sh.overflow = 0; }
  if (tid < kTMaxI) sh.delta[tid] = 0;
  if (tid < T && k > 1) sh.R[tid] = (unsigned long long)ceil(rand[P.rand_off + tid] * (double)psum);
  __syncthreads();
  const int nch = (nsb + 63) >> 6;
  const unsigned long long below = (1ull << lane) - 1ull;
  const int rq = lane >> 4, rj = lane & 15;
#ifdef RHCCQ_STAMPS
  unsigned long long _acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long _last = clock64();
  unsigned long long _wacc[4] = {0, 0, 0, 0}, _wt = 0;
#define WBEGIN() _wt = clock64()
#define WEND(ph) _wacc[ph] += clock64() - _wt
#else
#define WBEGIN() do {} while (0)
#define WEND(ph) do {} while (0)
#endif
  for (int c = 1; c < k; ++c) {
    if (kSh && sh.abort) break;                           // (uniform: read behind the barrier that closed the previous pick)
    WBEGIN();
    // the next pick's uniforms are a cold line in HBM: fetch them now, use them at the end of the pick
    double u_next = 0.0;
    if (tid < T && c + 1 < k) u_next = rand[P.rand_off + (size_t)c * T + tid];
    const unsigned long long pot = sh.pot;
    if (tid < T) sh.delta[tid] = 0;                      // (read for the arg-max before the previous pick's closing barrier)
    const int* touch_r = s_touch + (((c - 1) & 1) ? kJTouch : 0);
    const int n_touched = min(sh.n_touch2[(c - 1) & 1], kJTouch);
    // ================= phase 1: waves t < T -- candidate t ====================================================
    if (wave < T) {
      // np.searchsorted(cumsum(closest), r, 'left') in DRAW order; cum and the target R = ceil(r) are exact integers
      const int t = wave;
      const unsigned long long rv = sh.R[t];
      const unsigned long long R = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(rv >> 32)) << 32) |
                                   (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)rv);
      int cand = R == 0 ? 0 : n - 1;                      // R = 0: position 0; a target beyond the total (cannot happen): the last
      uint32_t ck = 0;
      bool found = false;
      // sharded: the shard whose range of the global cumulative sum holds the target searches; the others wait for its answer
      bool searcher = true;
      unsigned long long Rl = R;
      if (kSh) {
        int own = R == 0 ? 0 : nshard - 1;
        unsigned long long cum = 0, before = 0;
        bool hit = R == 0;
        for (int s2 = 0; s2 < nshard; ++s2) {
          const unsigned long long v = sh.tot[s2];
          if (!hit && v > 0 && cum < R && R <= cum + v) { own = s2; before = cum; hit = true; }
          if (!hit && s2 == nshard - 1) before = cum;
          cum += v;
        }
        searcher = own == me;
        Rl = R - before;
        if (lane == 0) sh.owner[t] = own;
      }
      if (searcher && R != 0) {
        const unsigned long long R = Rl;                  // (the local target; shadows the global one inside the search)
        unsigned long long carry = 0;
        for (int ch = 0; ch < nch && !found; ++ch) {
          const int sb = ch * 64 + lane;
          const unsigned v = sb < nsb ? dssum[sb] : 0u;
          const unsigned long long inc = carry + wave_incscan_limbs(v);
          const unsigned long long exc = inc - v;
          const unsigned long long m1 = __ballot(v > 0 && exc < R && R <= inc);
          if (m1) {
            const int l1 = __ffsll((long long)m1) - 1;
            const int sbh = ch * 64 + l1;
            const unsigned long long base = readlane64(exc, l1);
            // level 2: the 16 draw blocks of the super-block
            const int b2 = sbh * 16 + (lane & 15);
            const unsigned v2 = (lane < 16 && b2 < nb) ? dsum[b2] : 0u;
            const unsigned long long cum2 = base + row_incscan_u32(v2);
            const unsigned long long m2 = __ballot(lane < 16 && v2 > 0 && (cum2 - v2) < R && R <= cum2);
            if (m2) {
              const int l2 = __ffsll((long long)m2) - 1;
              const int bh = sbh * 16 + l2;
              const unsigned rr = (unsigned)(R - readlane64(cum2 - v2, l2));     // <= the block's sum
              // level 3: the 64 samples of the draw block
              const int i = (bh << 6) + lane;
              const uint2 sv = dsamp[i];
              const unsigned inc3 = wave_incscan_u32(i < n ? sv.y : 0u);          // 64 x 195075 fits 32 bits
              const unsigned long long m3 = __ballot(i < n && inc3 >= rr);
              const int l3 = m3 ? __ffsll((long long)m3) - 1 : min(63, n - 1 - (bh << 6));
              cand = (bh << 6) + l3;
              ck = (uint32_t)__builtin_amdgcn_readlane((int)sv.x, l3);
              found = true;
            }
            break;
          }
          carry = readlane64(inc, 63);
        }
      }
      if (kSh) {
        if (searcher) {
          if (!found) ck = dsamp[cand].x;
          if (lane == 0) jx_store(&xch->cand[c & 1][t], (unsigned)c, (unsigned long long)ck);
        } else {
          unsigned long long v = 0;
          if (!jx_wait(&xch->cand[c & 1][t], true, (unsigned)c, &xch->abort, v) && lane == 0) {
            sh.abort = 1;
            __hip_atomic_store(&xch->abort, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          ck = (uint32_t)v & 0xffffffu;
        }
      } else if (!found) ck = dsamp[cand].x;
      const CandP cp = cand_pairs(ck);
      if (lane == 0) {
        sh.cand[t] = cand;
        sh.ckey[t] = ck;
        sh.cna[t] = norm2_key(ck);
        sh.ckp[t][0] = cp.r; sh.ckp[t][1] = cp.g; sh.ckp[t][2] = cp.b;
      }
      // ... and, while the candidate is in registers, the super-blocks it may improve (their maxima may be mid-refresh by
      // the idle waves: a stale, larger maximum is conservative); two chunks of 64 per round, one list append for both
      for (int ch = 0; ch < nch; ch += 2) {
        const int s0 = ch * 64 + lane, s1 = s0 + 64;
        bool h0 = false, h1 = false;
        if (s0 < nsb) {
          const uint4 se = sup[s0];
          h0 = box_dist2(cp, se.x, se.y, se.z) < se.w;
        }
        if (s1 < nsb) {
          const uint4 se = sup[s1];
          h1 = box_dist2(cp, se.x, se.y, se.z) < se.w;
        }
        const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1);
        const int c0 = __popcll(m0), c1 = __popcll(m1);
        if (c0 + c1) {
          int base = 0;
          if (lane == 0) base = atomicAdd(&sh.n_hits, c0 + c1);
          base = __builtin_amdgcn_readfirstlane(base);
          if (h0) hits[base + __popcll(m0 & below)] = ((uint32_t)t << 24) | (uint32_t)s0;
          if (h1) hits[base + c0 + __popcll(m1 & below)] = ((uint32_t)t << 24) | (uint32_t)s1;
        }
      }
    } else {
      // the other waves refresh the super-block maxima the previous winner touched (a stale, larger maximum is
      // conservative, so the enumeration may read either value)
      for (int i = tid - T * 64; i < n_touched * 16; i += kJThreads - T * 64) {
        const int sb = touch_r[i >> 4], b = sb * 16 + (i & 15);
        unsigned m = b < nb ? blk[b].w : 0u;
        m = dpp_row_max(m);
        if ((i & 15) == 0) sup[sb].w = m;
      }
    }
    WEND(0);
    STAMP(0);
    __syncthreads();
    STAMP(1);
    WBEGIN();
    // ================= phase 2: hit super-blocks -> blocks ====================================================
    // four (candidate, super-block) hits per wave instruction stream, one row of lanes each, one lane per block
    {
      const int n_hits = sh.n_hits;
      for (int h0 = 4 * wave; h0 < n_hits; h0 += 4 * kJWaves) {
        const uint32_t hw = h0 + rq < n_hits ? hits[h0 + rq] : 0xffffffffu;
        const int t = (int)(hw >> 24) & (kTMaxI - 1);
        const int b = (int)(hw & 0xffffffu) * 16 + rj;
        bool hb = hw != 0xffffffffu && b < nb;
        if (hb) {
          const CandP cp{sh.ckp[t][0], sh.ckp[t][1], sh.ckp[t][2]};
          const uint4 be = blk[b];
          hb = box_dist2(cp, be.x, be.y, be.z) < be.w;
        }
        const unsigned long long mb = __ballot(hb);
        const int cnt = __popcll(mb);
        if (cnt) {
          int base = 0;
          if (lane == 0) base = atomicAdd(&sh.n_items, cnt);
          base = __builtin_amdgcn_readfirstlane(base);
          if (base + cnt > max_items) { if (lane == 0) sh.overflow = 1; }
          else if (hb) items[base + __popcll(mb & below)] = ((uint32_t)t << 24) | (uint32_t)b;
        }
      }
    }
    WEND(1);
    STAMP(2);
    __syncthreads();
    STAMP(3);
    WBEGIN();
    // ================= phase 3: potentials ===================================================================
    const bool use_list = sh.overflow == 0;
    const int n_items = use_list ? sh.n_items : 0;
    const int n_ops = (n_items + 3) >> 2;                  // one evaluation instruction stream = 4 items, one per row
    const bool kept = n_ops <= kJWaves * kJKeep;           // every item's samples stay in the registers of its wave
    int* touch_w = s_touch + ((c & 1) ? kJTouch : 0);
    int* n_touch_w = &sh.n_touch2[c & 1];
    uint4 ka[kJKeep], kb[kJKeep];
    uint32_t kw[kJKeep];
    if (use_list) {
      // (wave-uniform guards: an unused slot costs two scalar instructions; a row beyond the list re-reads the last item
      // and contributes nothing -- no divergent control flow inside a slot)
#pragma unroll
      for (int s = 0; s < kJKeep; ++s) {
        kw[s] = 0xffffffffu;
        ka[s] = make_uint4(0, 0, 0, 0);
        kb[s] = ka[s];
        if (4 * (wave + s * kJWaves) < n_items) {
          const int ii = 4 * (wave + s * kJWaves) + rq;
          const uint32_t w = items[min(ii, n_items - 1)];
          const uint4* p4 = reinterpret_cast<const uint4*>(samp + ((w & 0xffffffu) << 6) + 4 * rj);
          ka[s] = p4[0];
          kb[s] = p4[1];
          kw[s] = ii < n_items ? w : (w | 0xf0000000u);    // candidate numbers are < 16: the high nibble marks a padding row
        }
      }
#pragma unroll
      for (int s = 0; s < kJKeep; ++s) {
        if (4 * (wave + s * kJWaves) < n_items) {
          const int t = (int)(kw[s] >> 24) & (kTMaxI - 1);
          unsigned imp = j_eval4(sh.ckey[t], (int)sh.cna[t], ka[s], kb[s]);
          imp = dpp_row_sum((kw[s] >> 28) ? 0u : imp);
          if (rj == 0 && imp) atomicAdd(&sh.delta[t], (unsigned long long)imp);
        }
      }
      for (int o = wave + kJKeep * kJWaves; o < n_ops; o += kJWaves) {   // more items than the registers hold
        const int ii = 4 * o + rq;
        const uint32_t w = ii < n_items ? items[ii] : 0xffffffffu;
        const bool on = w != 0xffffffffu;
        unsigned imp = 0;
        if (on) {
          const uint4* p4 = reinterpret_cast<const uint4*>(samp + ((w & 0xffffffu) << 6) + 4 * rj);
          const int t = (int)(w >> 24);
          imp = j_eval4(sh.ckey[t], (int)sh.cna[t], p4[0], p4[1]);
        }
        imp = dpp_row_sum(imp);
        if (on && rj == 0 && imp) atomicAdd(&sh.delta[w >> 24], (unsigned long long)imp);
      }
    } else {
      // the item list overflowed (the first picks, when every block can still improve): each candidate is enumerated
      // and evaluated by one wave, no list
      for (int t = wave; t < T; t += kJWaves) {
        const CandP cp{sh.ckp[t][0], sh.ckp[t][1], sh.ckp[t][2]};
        const uint32_t ck = sh.ckey[t];
        const int na = (int)sh.cna[t];
        unsigned long long acc = 0;
        for (int ch = 0; ch < nch; ++ch) {
          j_enumerate(cp, ch * 64, nsb, nb, sup, blk, [&](unsigned long long mb, int b, bool hb) {
            j_for_rows(mb, b, [&](int my_b, bool on) {
              if (on) {
                const uint4* p4 = reinterpret_cast<const uint4*>(samp + (my_b << 6) + 4 * rj);
                acc += j_eval4(ck, na, p4[0], p4[1]);
              }
            });
          });
        }
        acc = wave_sum(acc);
        if (lane == 0) sh.delta[t] = acc;
      }
    }
    WEND(2);
    STAMP(4);
#ifdef RHCCQ_STAMPS
    _acc[10] += (unsigned long long)n_items;
    _acc[11] += use_list ? (kept ? 0 : 1) : 0;
    _acc[12] += use_list ? 0 : 1;
#endif
    __syncthreads();
    STAMP(5);
    WBEGIN();
    if (kSh) {
      // every shard's partial improvements: publish mine, collect the others' (wave w collects shard w)
      if (wave == 0 && lane < T) jx_store(&xch->delta[c & 1][me][lane], (unsigned)c, sh.delta[lane]);
      if (wave < nshard) {
        unsigned long long v = lane < T ? sh.delta[lane] : 0ull;
        if (wave != me) {
          if (!jx_wait(&xch->delta[c & 1][wave][lane < T ? lane : 0], lane < T, (unsigned)c, &xch->abort, v) && lane == 0) {
            sh.abort = 1;
            __hip_atomic_store(&xch->abort, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        if (lane < T) sh.gd[wave][lane] = v;
      }
      __syncthreads();
    }
    // ================= phase 4: greedy choice + commit =======================================================
    // largest reduction == smallest potential; the first candidate wins ties
    unsigned long long dv = lane < T ? sh.delta[lane] : 0ull;
    if (kSh) {
      dv = 0;
      if (lane < T)
        for (int s2 = 0; s2 < nshard; ++s2) dv += sh.gd[s2][lane];
    }
    const unsigned dhi = (unsigned)(dv >> 32), dlo = (unsigned)dv;
    const unsigned mhi = wave_max_u32(dhi);
    const unsigned mlo = wave_max_u32(dhi == mhi ? dlo : 0u);
    const unsigned long long bd = ((unsigned long long)mhi << 32) | mlo;
    const int best = __ffsll((long long)__ballot(lane < T && dv == bd)) - 1;
    const uint32_t kbest = (uint32_t)__builtin_amdgcn_readlane((int)(lane < T ? sh.ckey[lane] : 0u), best);
    const int nabest = (int)norm2_key(kbest);
    if (use_list && kept) {
#pragma unroll
      for (int s = 0; s < kJKeep; ++s) {
        if (4 * (wave + s * kJWaves) < n_items) {
          const bool mine = (int)(kw[s] >> 24) == best;     // (a padding row's high nibble never matches)
          if (__ballot(mine))
            j_commit_row(mine, (int)(kw[s] & 0xffffffu), kbest, nabest, ka[s], kb[s], samp, dsamp, blk, dsum, dssum, touch_w, n_touch_w);
        }
      }
    } else if (use_list) {
      for (int o = wave; o < n_ops; o += kJWaves) {
        const int ii = 4 * o + rq;
        const uint32_t w = ii < n_items ? items[ii] : 0xffffffffu;
        const bool mine = w != 0xffffffffu && (int)(w >> 24) == best;
        if (!__ballot(mine)) continue;
        uint4 a = make_uint4(0, 0, 0, 0), bb = a;
        if (mine) {
          const uint4* p4 = reinterpret_cast<const uint4*>(samp + ((w & 0xffffffu) << 6) + 4 * rj);
          a = p4[0];
          bb = p4[1];
        }
        j_commit_row(mine, (int)(w & 0xffffffu), kbest, nabest, a, bb, samp, dsamp, blk, dsum, dssum, touch_w, n_touch_w);
      }
    } else {
      const CandP cp = cand_pairs(kbest);
      for (int ch = wave; ch < nch; ch += kJWaves) {
        j_enumerate(cp, ch * 64, nsb, nb, sup, blk, [&](unsigned long long mb, int b, bool hb) {
          j_for_rows(mb, b, [&](int my_b, bool on) {
            uint4 a = make_uint4(0, 0, 0, 0), bb = a;
            if (on) {
              const uint4* p4 = reinterpret_cast<const uint4*>(samp + (my_b << 6) + 4 * rj);
              a = p4[0];
              bb = p4[1];
            }
            j_commit_row(on, my_b, kbest, nabest, a, bb, samp, dsamp, blk, dsum, dssum, touch_w, n_touch_w);
          });
        });
      }
    }
    if (kSh) {
      if (tid == 0 && sh.owner[best] == me) {             // the searcher of the winner knows its draw position
        cho[c] = lo + sh.cand[best];
        const double c0 = (double)key_r(kbest), c1 = (double)key_g(kbest), c2 = (double)key_b(kbest);
        double* C = centres + (P.koff + c) * 4;
        C[0] = c0; C[1] = c1; C[2] = c2; C[3] = km64_csq(c0, c1, c2);
      }
      if (tid >= 64 && tid < 64 + nshard) sh.tot[tid - 64] -= sh.gd[tid - 64][best];
      if (tid == 0) { sh.pot = pot - bd; sh.n_items = 0; sh.n_hits = 0; sh.overflow = 0; sh.n_touch2[(c + 1) & 1] = 0; }
    } else if (tid == 0) { cho[c] = sh.cand[best]; sh.pot = pot - bd; sh.n_items = 0; sh.n_hits = 0; sh.overflow = 0; sh.n_touch2[(c + 1) & 1] = 0; }
    if (tid < T) sh.R[tid] = (unsigned long long)ceil(u_next * (double)(pot - bd));
    WEND(3);
    STAMP(6);
    __syncthreads();
    STAMP(7);
    // more touched super-blocks than the list holds (only in the first picks): refresh all of them
    if (sh.n_touch2[c & 1] > kJTouch) {
      for (int sb = tid; sb < nsb; sb += kJThreads) {
        unsigned m = 0;
        for (int b = sb * 16; b < min(sb * 16 + 16, nb); ++b) m = max(m, blk[b].w);
        sup[sb].w = m;
      }
      __syncthreads();
      if (tid == 0) sh.n_touch2[c & 1] = 0;
      __syncthreads();
    }
  }
#ifdef RHCCQ_STAMPS
  if (tid == 0 && blockIdx.x == gridDim.x - 1)
    for (int i = 0; i < 16; ++i) g_init_stamps[i] += _acc[i];
  if (lane == 0 && blockIdx.x == gridDim.x - 1)
    for (int i = 0; i < 4; ++i) atomicAdd(&g_wave_stamps[i][wave], _wacc[i]);
#endif
  if (kSh) {                                               // (every centre was written by the shard that found it)
    // a hand-off that never arrived: poison the first centre's norm; rhccq_mbk_steps turns that into state code 4
    if (tid == 0 && (sh.abort || jx_load(&xch->abort) != 0)) centres[P.koff * 4 + 3] = __longlong_as_double(0x7ff8000000000000ll);
    return;
  }
  for (int j = tid; j < k; j += kJThreads) {
    const uint32_t kk = dsamp[cho[j]].x;
    const double c0 = (double)key_r(kk), c1 = (double)key_g(kk), c2 = (double)key_b(kk);
    double* C = centres + (P.koff + j) * 4;
    C[0] = c0; C[1] = c1; C[2] = c2; C[3] = km64_csq(c0, c1, c2);
  }
}

#include "k8_init3.h"
#ifdef RHCCQ_STAMPS
#define KPP_STAMPS
#endif
#include "kpp_flat.h"

// ---- k-means++ chain for small init samples (kpp_flat.h): one workgroup of 512 threads per problem, up to 16 samples per thread ----
constexpr int kFlatThreads = 512;
constexpr int kFlatMaxSamples = kFlatThreads * 16;
template <int kS>
__global__ __launch_bounds__(kFlatThreads) void mbk_init_flat_kernel(const uint32_t* __restrict__ keys, const MbkP* __restrict__ probs,
                                                                     const int32_t* __restrict__ init_idx, const double* __restrict__ rand,
                                                                     double* __restrict__ centres, int32_t* __restrict__ chosen) {
  __shared__ FlatShared sh;
  const MbkP P = probs[blockIdx.x];
  const int n = (int)P.init_n, k = (int)P.k;
  if (n > kS * kFlatThreads) return;                      // (rhccq_mbk_init picks kS from the largest problem of the call)
  const int32_t* idx = init_idx + P.init_off;
  const uint32_t* kp = keys + P.off;
  int32_t* cho = chosen + P.koff;
  kpp_flat<kS, kFlatThreads / 64>([&](int i) { return kp[idx[i]]; }, n, k, P.T, P.first, rand + P.rand_off, cho, sh);
  for (int j = threadIdx.x; j < k; j += kFlatThreads) {
    const uint32_t kk = kp[idx[cho[j]]];
    const double c0 = (double)key_r(kk), c1 = (double)key_g(kk), c2 = (double)key_b(kk);
    double* C = centres + (P.koff + j) * 4;
    C[0] = c0; C[1] = c1; C[2] = c2; C[3] = km64_csq(c0, c1, c2);
  }
}

// ------------------------------------------------------------------------------------------------
// mini-batch steps
// ------------------------------------------------------------------------------------------------
constexpr int kBatch = 1024;        // padded batch (sklearn batch_size = 1000)
constexpr int kTileC = 512;         // centres per workgroup in the batch E-step (16 KB of LDS; 1024 halves the partials the
                                    // update folds but leaves too few workgroups: measured 9 % slower per step)
constexpr int kTileS = 256;         // centres per workgroup of the speculative E-step (k8_overlap.h): the partial arrays are sized for these
constexpr int kPtChunks = kBatch / 512;   // a workgroup takes 512 / split of the batch rows (two per thread)

// Per-problem state, double[16] (see rhccq_mbk_steps).  Every kernel of step `step` (the launch index, equal for all
// problems of a call sequence) reads only slots that no kernel of the same launch writes:
//   [0] ewa  [1] ewa_min  [2] no_improvement  [6] have_ewa  [7] have_min      -- role 1 of the update only
//   [4] why it stopped (1 converged, 2 out of steps, 3 word table exhausted)    [5] steps done   [11] stop_at
//   [3] / [12]  samples since the last reassignment, as seen by an even / odd step
//   [8] / [13]  zero-weight centres,                  "
//   [9] / [14]  MT19937 words consumed,               "          [10] first batch drawn
// a step reads the slots of its parity and writes those of the next step's; stop_at = s + 1 is written during step s and
// means "stopped before step s + 1", so the other roles of step s still see the problem running.
constexpr int kStSince = 3, kStNzero = 8, kStCursor = 9;
__device__ __forceinline__ int st_slot(int even_slot, long long step) {
  return (step & 1) == 0 ? even_slot : (even_slot == kStSince ? 12 : even_slot == kStNzero ? 13 : 14);
}
__device__ __forceinline__ bool mbk_stopped(const double* st, long long step, long long n) {
  const double stop_at = st[11];
  const long long bs = n < 1000 ? n : 1000;
  return (stop_at != 0.0 && (double)step >= stop_at) || st[4] >= 3.0 || step >= (100 * n) / bs;
}
// Batch inertia of step `step` + sklearn _mini_batch_convergence (EWA early stopping), one wave.  The 1 000 terms (fold
// kernel, against the centres before the update) are added ONE AFTER THE OTHER in batch order, as sklearn's
// single-threaded _inertia_dense does -- a chain of ~1 000 dependent float64 additions, ~10 us.  Nothing of step `step`
// needs its outcome, only the NEXT update does, so the chain rides in an extra workgroup of the next step's E-step
// kernel (and in a kernel of its own behind the last step of a launch sequence).
__device__ __forceinline__ void mbk_inertia_block(const MbkP& P, double* st, long long step, const double* __restrict__ pper_p, double* s_per) {
  const int lane = threadIdx.x;
  if (lane >= 64) return;
  const long long n = P.n;
  const int bs = (int)min((long long)1000, n);
  const long long n_steps_max = (100 * n) / bs;
  // The sum must visit the rows in batch order, one rounding per addition (sklearn's _inertia_dense), but nothing says ONE lane
  // has to hold all of them: lane l keeps rows 16 l .. 16 l + 15 in registers and the running sum walks the lanes -- every turn all
  // lanes add their 16 rows to the sum so far, the lane whose turn it is has the true value, a readlane hands it on.  Same 1 024
  // dependent additions in the same order, no LDS round trip between them (a single lane re-reading LDS every 8 additions took
  // 12 us, and the E-step kernel this chain rides in lasted as long as the chain: 14 us of a 30 us step).
  double v[16];
  {
    const double4* src = reinterpret_cast<const double4*>(pper_p + 16 * lane);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double4 u = src[q];
      v[4 * q] = u.x; v[4 * q + 1] = u.y; v[4 * q + 2] = u.z; v[4 * q + 3] = u.w;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = 16 * lane + j < bs ? v[j] : 0.0;     // rows beyond the batch: + 0.0 changes nothing
  }
  (void)s_per;
  double inertia = 0.0;
  for (int t = 0; t < 64; ++t) {
    double x = inertia;
#pragma unroll
    for (int j = 0; j < 16; ++j) x = x + v[j];
    const unsigned long long bits = (unsigned long long)__double_as_longlong(x);
    inertia = __longlong_as_double((long long)readlane64(bits, t));
  }
  if (lane != 0) return;
  st[5] = (double)(step + 1);
  double stop = 0.0;
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
    if (st[2] >= 10.0) stop = 1.0;
  }
  if (step + 1 >= n_steps_max && stop == 0.0) stop = 2.0;
  if (stop != 0.0) { st[4] = stop; st[11] = (double)(step + 1); }
}

// the inertia of the last step of a launch sequence (nothing follows it to ride on)
__global__ __launch_bounds__(64) void mbk_inertia_kernel(const MbkP* __restrict__ probs, double* __restrict__ state, long long step,
                                                         const double* __restrict__ pper) {
  __shared__ double s_per[kBatch];
  const int p = blockIdx.x;
  const MbkP P = probs[p];
  double* st = state + p * 16;
  if (step < 0 || mbk_stopped(st, step, P.n)) return;
  mbk_inertia_block(P, st, step, pper + (size_t)p * kBatch, s_per);
}

// kSplit threads share one batch point, each scanning a contiguous 1/kSplit of the tile's centres: with a single
// straggler problem still running there are only ~160 workgroups for 256 CUs and a thread's serial walk over 512
// centres (one wave per SIMD, ~8 cycles per dependent f64 instruction) is the whole step; splitting the walk keeps
// the arithmetic and the first-arg-min order (lower slices win ties) and shortens the chain.
#ifdef RHCCQ_STAMPS
__device__ unsigned long long g_spec_stamps[8];          // speculative E-step, summed over workgroups: load, exclude, loop, merge; [7] workgroups
#define SSTAMP(slot) do { if (excl_labels != nullptr && threadIdx.x == 0) { const unsigned long long _t = clock64(); atomicAdd(&g_spec_stamps[slot], _t - _sl); _sl = _t; } } while (0)
#else
#define SSTAMP(slot) do {} while (0)
#endif
template <int kSplit>
__device__ __forceinline__ void estep_tile(const int bx, const int p, const MbkP& P, const long long po, const double* __restrict__ centres,
                                           const uint32_t* __restrict__ bkeys, double* __restrict__ pdist, int32_t* __restrict__ pidx,
                                           const int32_t* __restrict__ excl_labels, double* sc, double* s_bd, int* s_bj) {
  // a thread owns TWO batch rows and one slice of the tile's centres: every centre it reads from LDS (a broadcast read, and
  // with four workgroups per CU the LDS pipe was the limit) serves two distance evaluations
  constexpr int kT = 256 / kSplit, kPts = 2 * kT, kSlice = kTileC / kSplit;
  const int n_tiles = (int)((P.k + kTileC - 1) / kTileC);
  const int tile = bx / (kPtChunks * kSplit), chunk = bx % (kPtChunks * kSplit);
  if (tile >= n_tiles) return;
  const int bs = (int)min((long long)1000, P.n);
  const int j0 = tile * kTileC, nj = (int)min((long long)kTileC, P.k - j0);
#ifdef RHCCQ_STAMPS
  unsigned long long _sl = clock64();
  if (excl_labels != nullptr && threadIdx.x == 0) atomicAdd(&g_spec_stamps[7], 1ull);
#endif
  int ex[kBatch / 256];                                  // (requested together with the tile)
  if (excl_labels != nullptr) {
#pragma unroll
    for (int q = 0; q < kBatch / 256; ++q) ex[q] = (int)threadIdx.x + q * 256 < bs ? excl_labels[threadIdx.x + q * 256] - j0 : -1;
  }
  {
    // the tile: all of a thread's loads in flight together (a load -> store loop waited for every one of its 8 reads in turn:
    // 8 us of a 14 us kernel)
    constexpr int kLd = kTileC * 2 / 256;                 // 16 bytes per load: (c0, c1) or (c2, csq)
    const double2* src = reinterpret_cast<const double2*>(centres + (P.koff + j0) * 4);
    double2 cv[kLd];
#pragma unroll
    for (int q = 0; q < kLd; ++q) {
      const int i = (int)threadIdx.x + q * 256;
      cv[q] = i < nj * 2 ? src[i] : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int q = 0; q < kLd; ++q) {
      const int i = (int)threadIdx.x + q * 256;
      // pre-scale by -2 (exact): dist = csq + dot'
      if (i < nj * 2) reinterpret_cast<double2*>(sc)[i] = make_double2(-2.0 * cv[q].x, (i & 1) ? cv[q].y : -2.0 * cv[q].y);
    }
  }
  __syncthreads();
  SSTAMP(0);
  if (excl_labels != nullptr) {
    // speculative E-step (k8_overlap.h): the centres these labels name are being rewritten by the update that runs beside this
    // launch -- distance +inf, they are compared at their new values by the next launch
#pragma unroll
    for (int q = 0; q < kBatch / 256; ++q)
      if (ex[q] >= 0 && ex[q] < nj) sc[ex[q] * 4 + 3] = INFINITY;
    __syncthreads();
  }
  SSTAMP(1);
  double* pd = pdist + po + (size_t)tile * kBatch;
  int32_t* pi = pidx + po + (size_t)tile * kBatch;
  const int pt = threadIdx.x % kT, slice = threadIdx.x / kT;
  const int b0 = chunk * kPts + pt, b1 = b0 + kT;
  double bd0 = INFINITY, bd1 = INFINITY;
  int bj0 = excl_labels != nullptr ? 0 : 0x7fffffff, bj1 = bj0;      // (a tile whose centres are all excluded reports +inf)
  if (b0 < bs) {
    const uint32_t k0 = bkeys[(size_t)p * kBatch + b0], k1 = bkeys[(size_t)p * kBatch + min(b1, bs - 1)];   // rows drawn by an earlier update
    const double x0 = (double)key_r(k0), x1 = (double)key_g(k0), x2 = (double)key_b(k0);
    const double y0 = (double)key_r(k1), y1 = (double)key_g(k1), y2 = (double)key_b(k1);
    const int ja = slice * kSlice, jb = min(ja + kSlice, nj);
    for (int j = ja; j < jb; ++j) {
      const double c0 = sc[j * 4], c1 = sc[j * 4 + 1], c2 = sc[j * 4 + 2], cs = sc[j * 4 + 3];
      const double d0 = cs + km64_dot(x0, x1, x2, c0, c1, c2);
      const double d1 = cs + km64_dot(y0, y1, y2, c0, c1, c2);
      if (d0 < bd0) { bd0 = d0; bj0 = j; }
      if (d1 < bd1) { bd1 = d1; bj1 = j; }
    }
  }
  SSTAMP(2);
  if (kSplit > 1) {
    s_bd[threadIdx.x] = bd0; s_bd[256 + threadIdx.x] = bd1;
    s_bj[threadIdx.x] = bj0; s_bj[256 + threadIdx.x] = bj1;
    __syncthreads();
    if (slice == 0) {
#pragma unroll
      for (int q = 1; q < kSplit; ++q) {                  // ascending slices = ascending centre index: strict '<'
        const double o0 = s_bd[q * kT + pt], o1 = s_bd[256 + q * kT + pt];
        if (o0 < bd0) { bd0 = o0; bj0 = s_bj[q * kT + pt]; }
        if (o1 < bd1) { bd1 = o1; bj1 = s_bj[256 + q * kT + pt]; }
      }
    }
  }
  if (slice == 0) {
    if (b0 < bs) { pd[b0] = bd0; pi[b0] = j0 + bj0; }
    if (b1 < bs) { pd[b1] = bd1; pi[b1] = j0 + bj1; }
  }
  SSTAMP(3);
}

template <int kSplit>
__global__ __launch_bounds__(256) void mbk_batch_estep_kernel(const uint32_t* __restrict__ keys, const MbkP* __restrict__ probs,
                                                              const double* __restrict__ centres, double* __restrict__ state,
                                                              long long step, const uint32_t* __restrict__ bkeys, double* __restrict__ pdist,
                                                              int32_t* __restrict__ pidx, const long long* __restrict__ part_off,
                                                              const double* __restrict__ pper_prev) {
  const int p = blockIdx.y;
  const MbkP P = probs[p];                               // independent table reads, issued together
  const long long po = part_off[p];
  __shared__ __align__(16) double sc[kTileC * 4];
  __shared__ double s_bd[kSplit > 1 ? 512 : 1];
  __shared__ int s_bj[kSplit > 1 ? 512 : 1];
  if (blockIdx.x == gridDim.x - 1) {                     // the extra workgroup: inertia + EWA rule of the PREVIOUS step
    if (pper_prev != nullptr && !mbk_stopped(state + p * 16, step - 1, P.n))
      mbk_inertia_block(P, state + p * 16, step - 1, pper_prev + (size_t)p * kBatch, sc);
    return;
  }
  if (mbk_stopped(state + p * 16, step, P.n)) return;
  estep_tile<kSplit>((int)blockIdx.x, p, P, po, centres, bkeys, pdist, pidx, nullptr, sc, s_bd, s_bj);
}

// Fold of the per-tile partial arg-mins, in tile order (first arg-min), into the slot of tile 0.  One wave per 16
// batch points: lane = (tile group g << 4) | point; a lane loads its tiles g, g + 4, g + 8, ... (16 loads in flight,
// 16 consecutive points = one full line per tile), keeps their first minimum, and two shuffles merge the four
// groups (smaller distance, then smaller tile).  Folding inside the single-workgroup update kernel meant ~700 KB
// through one CU's few outstanding misses (10-14 us of every step); a last-arriving-workgroup fold inside the
// E-step needs an agent-scope release (an L2 write-back) in every one of its ~900 workgroups and was slower still.
__global__ __launch_bounds__(64) void mbk_fold_tiles_kernel(const MbkP* __restrict__ probs, const double* __restrict__ state, long long step,
                                                            const double* __restrict__ centres, const uint32_t* __restrict__ bkeys,
                                                            double* __restrict__ pdist, int32_t* __restrict__ pidx,
                                                            const long long* __restrict__ part_off, double* __restrict__ pper, int tiled) {
  const int p = blockIdx.y;
  const MbkP P = probs[p];
  const long long po = part_off[p];
  if (mbk_stopped(state + p * 16, step, P.n)) return;
  const int n_tiles = tiled ? (int)((P.k + kTileC - 1) / kTileC) : 1;     // the grid E-step leaves its result in the slot of tile 0
  const int bs = (int)min((long long)1000, P.n);
  const int lane = threadIdx.x, g = lane >> 4;
  const int b = min(blockIdx.x * 16 + (lane & 15), bs - 1);      // clamped lanes redo the last point (same value written)
  double bd = INFINITY;
  int bt = 0;
  if (n_tiles > 1) {
    const double* fd = pdist + po + b;
    bt = 0x7fffffff;
    for (int t0 = g; t0 < n_tiles; t0 += 64) {
      double dv[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) dv[q] = t0 + 4 * q < n_tiles ? fd[(size_t)(t0 + 4 * q) * kBatch] : INFINITY;
#pragma unroll
      for (int q = 0; q < 16; ++q)
        if (dv[q] < bd) { bd = dv[q]; bt = t0 + 4 * q; }
    }
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
      const double od = __shfl_xor(bd, o, 64);
      const int ot = __shfl_xor(bt, o, 64);
      if (od < bd || (od == bd && ot < bt)) { bd = od; bt = ot; }
    }
  }
  if (g == 0) {
    const int32_t bj = pidx[po + b + (size_t)bt * kBatch];
    if (bt != 0) {
      pdist[po + b] = bd;
      pidx[po + b] = bj;
    }
    // the row's term of the batch inertia (sklearn _inertia_dense -> _euclidean_dense_dense for 3 features: result = 0;
    // result += d * d per feature; 0 + x is exact), against the centres BEFORE this step's update; the update's second
    // workgroup adds the terms in batch order
    const uint32_t kk = bkeys[(size_t)p * kBatch + b];
    const double* c = centres + (P.koff + bj) * 4;
    const double d0 = (double)key_r(kk) - c[0], d1 = (double)key_g(kk) - c[1], d2 = (double)key_b(kk) - c[2];
    pper[(size_t)p * kBatch + b] = (d0 * d0 + d1 * d1) + d2 * d2;
  }
}

constexpr int kUpdThreads = 1024;
constexpr int kUpdWaves = kUpdThreads / 64;
constexpr int kHashSlots = 2048;
constexpr int kHistBins = 2048;
constexpr int kStateStride = 16;    // doubles per problem, see rhccq_mbk_steps
constexpr int kMemCap = 8;          // batch members listed per touched centre; beyond, the centre's thread scans the batch
constexpr int kStageWords = 2048;   // MT19937 words staged in LDS for the shuffle replay (expected need: ~1 400)
constexpr int kDrawWords = 4;       // words per thread and round of the randint replay (4 096 per round with 1 024 threads)
constexpr long long kWordsMargin = 16384;   // words a step may consume at most (randint ~2 000, shuffle ~1 400 expected)

struct UpdShared {
  int lab[kBatch];                 // labels of the batch; later the indices of the next batch
  uint32_t bkey[kBatch];
  double per[kBatch];
  int hkey[kHashSlots];
  int hcnt[kHashSlots];
  unsigned short hmem[kHashSlots][kMemCap];
  double hold[kHashSlots][4];      // the touched centre as it was read for the labels: c0, c1, c2, weight
  int perm[kBatch];                // reassignment: permutation(batch)[:n_reassign]
  unsigned short jswap[kBatch];    // the shuffle's j_i
  uint32_t stage[kStageWords];
  int hist[kHistBins];
  double dred[kUpdWaves];
  int ired[kUpdWaves + 1];
  int weq[kUpdWaves], wsel[kUpdWaves];
  double sel_w;
  int take;
  long long cursor;
};

__device__ __forceinline__ double block_max_d(double v, UpdShared& sh) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
  __syncthreads();
  if (lane == 0) sh.dred[w] = v;
  __syncthreads();
  double t = sh.dred[0];
  for (int i = 1; i < kUpdWaves; ++i) t = fmax(t, sh.dred[i]);
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
  for (int i = 1; i < kUpdWaves; ++i) t = fmin(t, sh.dred[i]);
  return t;
}
__device__ __forceinline__ int block_sum_i(int v, UpdShared& sh) { return block_sum<int>(v, sh.ired); }

// selection predicate of the low-count reassignment (index-ordered ties resolved by `eq_rank`)
__device__ __forceinline__ bool reassign_sel(double w, double thr, bool capped, double sel_w, int take, int eq_rank) {
  if (!(w < thr)) return false;
  if (!capped) return true;
  if (w < sel_w) return true;
  return w == sel_w && eq_rank < take;
}

// a centre with more batch members than its list holds: += x over the batch rows labelled j, in batch order.  Four labels per
// LDS read, the reads independent of the sums (a row-by-row walk waited ~100 cycles for every one of the 1 000 rows: 27 us)
__device__ __forceinline__ void walk_members(const int* __restrict__ lab, const uint32_t* __restrict__ bkey, int bs, int j, double& a0,
                                             double& a1, double& a2) {
  for (int b = 0; b < bs; b += 16) {
    int4 l[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) l[q] = *reinterpret_cast<const int4*>(lab + b + 4 * q);      // (lab has kBatch entries: rows >= bs hold -1)
    bool any = false;
#pragma unroll
    for (int q = 0; q < 4; ++q) any = any || l[q].x == j || l[q].y == j || l[q].z == j || l[q].w == j;
    if (!any) continue;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int v[4] = {l[q].x, l[q].y, l[q].z, l[q].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (v[e] != j) continue;
        const uint32_t kk = bkey[b + 4 * q + e];
        a0 = a0 + (double)key_r(kk); a1 = a1 + (double)key_g(kk); a2 = a2 + (double)key_b(kk);
      }
    }
  }
}

// ---- numpy's legacy RandomState, replayed from its raw MT19937 words (resident on the device, mt.py) ----------
// RandomState.randint(0, rng + 1, count) at word `cursor` (_bounded_integers.pyx, legacy use_masked path): candidates
// `word & mask` (mask = smallest 2^b - 1 >= rng), one word each, kept when <= rng.  All threads of the workgroup:
// thread t looks at kDrawWords consecutive words per round, a block scan orders the survivors.  out[0 .. count) (LDS)
// receives the values in draw order.  Returns the cursor behind the last consumed word (-1: the table is exhausted).
template <int kDW = kDrawWords>
__device__ __forceinline__ long long replay_randint(const uint32_t* __restrict__ words, long long n_words, long long cursor, unsigned rng,
                                                    int count, int* out, int* ired, long long* s_cursor) {
  const int tid = threadIdx.x;
  if (rng == 0u) {                                       // numpy draws nothing for a one-value range
    for (int i = tid; i < count; i += blockDim.x) out[i] = 0;
    __syncthreads();
    return cursor;
  }
  const uint32_t mask = 0xffffffffu >> __clz(rng);
  int produced = 0;
  while (true) {
    if (cursor + (long long)blockDim.x * kDW > n_words) return -1;
    const long long base = cursor + (long long)tid * kDW;
    uint32_t v[kDW];
    int cnt = 0;
#pragma unroll
    for (int q = 0; q < kDW; ++q) {
      v[q] = words[base + q] & mask;
      cnt += v[q] <= rng;
    }
    int tot;
    int pos = produced + block_exscan<int>(cnt, ired, &tot);
#pragma unroll
    for (int q = 0; q < kDW; ++q) {
      if (v[q] <= rng) {
        if (pos < count) {
          out[pos] = (int)v[q];
          if (pos == count - 1) *s_cursor = base + q + 1;
        }
        ++pos;
      }
    }
    __syncthreads();
    if (produced + tot >= count) return *s_cursor;
    produced += tot;
    cursor += (long long)blockDim.x * kDW;
  }
}

// RandomState.permutation(bs)[:n_take] -- what choice(bs, replace=False, size=n_take) returns (mtrand.pyx legacy):
// shuffle of arange(bs), `for i in reversed(range(1, bs)): j = random_interval(i); swap(x[i], x[j])`, random_interval(i)
// = masked rejection (mask = smallest 2^b - 1 >= i) on one 32-bit word per attempt.
//   phase A (wave 0): the j_i and the cursor behind the last word.  The word a lane looks at is fixed (every attempt
//     consumes one), only the i it faces depends on the rejections before it: lane t first assumes none, the first
//     lane that then rejects is a true rejection, the lanes behind it move up by one, and so on -- one ballot per
//     rejection, 64 words per window;
//   phase B: thread r < n_take follows position r backwards through the swaps (i = 1 .. bs-1) to the arange entry
//     that ends there.
// `stage` holds the words [stage_base, stage_base + kStageWords) (loaded by the whole workgroup beforehand).
__device__ __forceinline__ long long replay_permutation(const uint32_t* __restrict__ words, long long n_words, long long cursor, int bs,
                                                        int n_take, UpdShared& sh, long long stage_base) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave == 0) {
    int i_cur = bs - 1;
    long long c = cursor;
    bool fail = false;
    while (i_cur >= 1) {
      if (c + 64 > n_words) { fail = true; break; }
      const long long wi = c + lane, so = wi - stage_base;
      const uint32_t w = (so >= 0 && so < kStageWords) ? sh.stage[so] : words[wi];
      const unsigned long long below = (1ull << lane) - 1ull;
      // fixed point of "lane t faces i_cur - t + (rejections before t)": lane t's verdict only depends on the lanes below
      // it, so every sweep settles at least one more lane and a sweep that changes nothing has found THE solution; most
      // verdicts do not depend on the few rejections before them, so 2-4 sweeps do (a lane-by-lane walk needed ~17)
      unsigned long long rejm = 0ull;
      int i_t;
      uint32_t m;
      bool valid;
      while (true) {
        i_t = i_cur - lane + __popcll(rejm & below);
        valid = i_t >= 1;
        m = valid ? (0xffffffffu >> __clz((unsigned)i_t)) : 0u;
        const unsigned long long nm = __ballot(valid && (w & m) > (unsigned)i_t);
        if (nm == rejm) break;
        rejm = nm;
      }
      const bool acc = valid && !((rejm >> lane) & 1ull);
      if (acc) sh.jswap[i_t] = (unsigned short)(w & m);
      const unsigned long long am = __ballot(acc);
      const int n_acc = __popcll(am);
      if (i_cur - n_acc < 1) {                           // the shuffle ends inside this window: behind the lane that drew j_1
        c += 64 - __clzll((long long)am);
        i_cur = 0;
      } else {
        c += 64;
        i_cur -= n_acc;
      }
    }
    if (lane == 0) sh.cursor = fail ? -1ll : c;
  }
  __syncthreads();
  const long long out = sh.cursor;
  if (out >= 0 && tid < n_take) {
    int pos = tid;
    for (int i0 = 0; i0 < bs; i0 += 8) {                   // eight swaps per LDS read (entry 0 is no swap)
      const uint4 v = *reinterpret_cast<const uint4*>(&sh.jswap[i0]);
      const unsigned jj[8] = {v.x & 0xffffu, v.x >> 16, v.y & 0xffffu, v.y >> 16, v.z & 0xffffu, v.z >> 16, v.w & 0xffffu, v.w >> 16};
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int i = i0 + q, j = (int)jj[q];
        if (i >= 1 && i < bs) pos = pos == i ? j : (pos == j ? i : pos);
      }
    }
    sh.perm[tid] = pos;
  }
  __syncthreads();
  return out;
}

// the batch of the next step: rows = randint(0, n, bs), their colours -> bkeys (what the E-step kernels read)
template <int kDW = kDrawWords>
__device__ __forceinline__ long long draw_batch(const uint32_t* __restrict__ keys, const MbkP& P, const uint32_t* __restrict__ words,
                                                long long n_words, long long cursor, uint32_t* __restrict__ bkeys_p, int* out, int* ired,
                                                long long* s_cursor) {
  const int bs = (int)min((long long)1000, P.n);
  const long long c = replay_randint<kDW>(words, n_words, cursor, (unsigned)(P.n - 1), bs, out, ired, s_cursor);
  if (c < 0) return c;
  for (int b = threadIdx.x; b < bs; b += blockDim.x) bkeys_p[b] = keys[P.off + out[b]];
  return c;
}

// first batch of every problem (state[10] == 0: not drawn yet)
__global__ __launch_bounds__(kUpdThreads) void mbk_draw0_kernel(const uint32_t* __restrict__ keys, const MbkP* __restrict__ probs,
                                                                 double* __restrict__ state, const uint32_t* __restrict__ words,
                                                                 long long n_words, uint32_t* __restrict__ bkeys, const double* __restrict__ centres) {
  __shared__ int s_out[kBatch];
  __shared__ int s_red[kUpdWaves + 1];
  __shared__ long long s_cursor;
  const int p = blockIdx.x;
  double* st = state + p * kStateStride;
  if (st[10] != 0.0 || st[4] != 0.0) return;
  const MbkP P = probs[p];
  if (centres[P.koff * 4 + 3] != centres[P.koff * 4 + 3]) {   // NaN: the sharded k-means++ chain gave up on a hand-off (code 4)
    if (threadIdx.x == 0) st[4] = 4.0;
    return;
  }
  const long long c = draw_batch(keys, P, words, n_words, (long long)st[kStCursor], bkeys + (size_t)p * kBatch, s_out, s_red, &s_cursor);
  if (threadIdx.x == 0) {
    if (c < 0) st[4] = 3.0;                              // word table exhausted (the host sizes it so that this cannot happen)
    else { st[kStCursor] = (double)c; st[10] = 1.0; }     // batch of step 0: even slots, bkeys buffer 0
  }
}


// One mini-batch step after its E-step (sklearn _mini_batch_step): TWO workgroups per problem (blockIdx.y = role), on two
// CUs, because the step is a chain of latencies and these chains are independent:
//   role 0  centre update (update_center_dense) and, in the rare steps that reassign low-count centres, the reassignment
//           with its choice() replay followed by the next batch's draw;
//   role 1  the next step's batch (randint replay + colour gather) whenever this step does not reassign (then the stream
//           position is known before the step starts).
// (The third chain, batch inertia + EWA rule, rides in the next step's E-step kernel: mbk_inertia_block.)
#ifdef RHCCQ_STAMPS
__device__ unsigned long long g_upd_stamps[16];
#define USTAMP(slot) do { if (tid == 0 && p == 0) { const unsigned long long _t = clock64(); atomicAdd(&g_upd_stamps[slot], _t - _ul); _ul = _t; } } while (0)
#else
#define USTAMP(slot) do {} while (0)
#endif
#ifdef RHCCQ_STAMPS
#define RSTAMP(slot) do { if (tid == 0 && stamp) { const unsigned long long _t = clock64(); atomicAdd(&g_upd_stamps[slot], _t - _rl); _rl = _t; } } while (0)
#else
#define RSTAMP(slot) do {} while (0)
#endif
// The low-count reassignment of one step (sklearn _mini_batch_step with random_reassign): centres whose weight is below 1 % of
// the largest weight -- at most batch / 2 of them, the lightest first -- move to random rows of the batch and take the smallest
// weight of the centres that stay.  Wave w owns the contiguous index range [w R, (w + 1) R), 64 entries at a time.  The five
// sweeps over the weights read an LDS copy when it fits (kLds: the weights are counts, integers below 100 n <= 1.7e9, exact in
// 32 bits; the copy lives in the member tables of the update, which is over by now: k <= 30 720), otherwise global memory (L2)
// -- with 20 dependent reads per thread and sweep that was ~50 us of an 80 us step.
// The step is split over two launches so that the three chains of a reassigning step run side by side instead of one after the
// other (~85 us): launch 1 = the update kernel, role 0 selects (reassign_select: everything up to "which centres, how many")
// while role 1 replays choice() (the shuffle's stream position does not depend on the selection); launch 2
// (mbk_reassign_apply_kernel) moves the selected centres and, beside that, draws the next batch.  What travels between the
// launches is a ReSel record per problem.
#include "k8_npysort.h"
struct ReSel {
  double thr, sel_w, wmin;
  long long cursor_choice;          // MT cursor behind the shuffle (-1: word table exhausted)
  int tag_sel, tag_choice;          // step + 1 of the step the two halves belong to
  int capped, take, n_re, use_mask; // use_mask: a capped selection is the bit mask of npysort_head (numpy's order), not (sel_w, take, rank)
  int eq_base[kUpdWaves], rbase[kUpdWaves];
  int perm[kBatch / 2];             // permutation(batch)[:batch / 2]: more rows are never reassigned
};
constexpr size_t kWLdsOff = offsetof(UpdShared, per);
constexpr int kWLds = (int)((offsetof(UpdShared, perm) - offsetof(UpdShared, per)) / 4);
template <bool kLds>
__device__ __forceinline__ void reassign_select(UpdShared& sh, const double* __restrict__ W, const int k, const int bs, ReSel* __restrict__ rs,
                                                const long long step, const bool stamp, const QsScratch* __restrict__ qs) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef RHCCQ_STAMPS
  unsigned long long _rl = clock64();
#endif
  const int R = (((k + kUpdWaves - 1) / kUpdWaves) + 63) & ~63;
  const int j0 = wave * R, j1 = min(j0 + R, k);
  const int n_it = j1 > j0 ? (j1 - j0 + 63) / 64 : 0;
  unsigned* wl = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(&sh) + kWLdsOff);
  if (kLds) {
#pragma unroll 8
    for (int j = tid; j < k; j += kUpdThreads) wl[j] = (unsigned)W[j];
    __syncthreads();
  }
  RSTAMP(6);
#define RHCCQ_WSWEEP(...)                                                                                                     \
  for (int i = 0; i < n_it; ++i) {                                                                                            \
    const int j = j0 + lane + 64 * i;                                                                                         \
    (void)j;                                                                                                                  \
    const double w = j < j1 ? (kLds ? (double)wl[j] : W[j]) : INFINITY;   /* (+inf: never a candidate, never the minimum) */  \
    __VA_ARGS__                                                                                                               \
  }
  double wm = 0.0;
  RHCCQ_WSWEEP({ if (j < j1) wm = fmax(wm, w); })
  wm = block_max_d(wm, sh);
  const double thr = 0.01 * wm;
  const int cap = (int)(0.5 * (double)bs);
  const int nbins = (int)ceil(thr);                   // candidate weights are integers in [0, thr)
  const bool use_hist = nbins <= kHistBins;
  for (int i = tid; i < kHistBins; i += kUpdThreads) sh.hist[i] = 0;
  __syncthreads();
  int cnt = 0, c0 = 0, c1 = 0;                           // candidates; of them with weight 0 / 1 (the hot bins of the first
  RHCCQ_WSWEEP({                                         // steps: ~k centres share them -- counted in registers, not by atomics)
    if (w < thr) {
      ++cnt;
      if (w == 0.0) ++c0;
      else if (w == 1.0) ++c1;
      else if (use_hist) atomicAdd(&sh.hist[(int)w], 1);
    }
  })
  cnt = block_sum_i(cnt, sh);
  c0 = block_sum_i(c0, sh);
  c1 = block_sum_i(c1, sh);
  if (tid == 0 && use_hist) { sh.hist[0] = c0; if (kHistBins > 1 && thr > 1.0) sh.hist[1] = c1; }
  __syncthreads();
  // more than batch/2 candidates: sklearn keeps np.argsort(weights)[:batch/2] -- an unstable sort over tied counts.
  // qs != nullptr (RHCCQ_OPT_REASSIGN_ORDER = 1, the default): the slots numpy's scalar quicksort fills (k8_npysort.h), as a bit
  // mask; otherwise the stable order (weight, index) of rounds 1-3: a threshold weight + index-ordered ranks among its ties
  const bool capped = cnt > 0.5 * (double)bs;
  const bool use_mask = capped && qs != nullptr;
  double sel_w = thr;
  int take = 0;
  if (use_mask) {
    // (the sort takes over the LDS the weight copy sits in -- the update's member tables, 16 bytes per element: sweep 2 below then reads the
    // weights from global memory, one sweep)
    npysort_head(sh, W, k, cap, *qs, -1, wl, kWLds / 4);
  } else if (capped) {
    if (use_hist) {
      if (tid == 0) {                                 // smallest weight v with #(W <= v) >= cap
        int below = 0, v = 0;
        for (; v < nbins; ++v) {
          if (below + sh.hist[v] >= cap) break;
          below += sh.hist[v];
        }
        sh.sel_w = (double)v;
        sh.take = cap - below;
      }
      __syncthreads();
      sel_w = sh.sel_w;
      take = sh.take;
    } else {
      double lo_v = -1.0, hi_v = floor(thr);
      if (hi_v >= thr) hi_v -= 1.0;
      while (hi_v - lo_v > 1.0) {
        const double mid = floor((lo_v + hi_v) * 0.5);
        int c2 = 0;
        RHCCQ_WSWEEP({ c2 += (w < thr) && (w <= mid); })
        c2 = block_sum_i(c2, sh);
        if (c2 >= cap) hi_v = mid; else lo_v = mid;
      }
      int below = 0;
      RHCCQ_WSWEEP({ below += (w < thr) && (w < hi_v); })
      below = block_sum_i(below, sh);
      sel_w = hi_v;
      take = cap - below;
    }
  }
  // index-ordered ranks inside the wave's range, 64 at a time
  int eq_base = 0;
  if (capped && !use_mask) {
    int eq = 0;
    RHCCQ_WSWEEP({ eq += (w < thr) && (w == sel_w); })
    eq = (int)wave_sum((unsigned long long)eq);
    if (lane == 0) sh.weq[wave] = eq;
    __syncthreads();
    for (int w = 0; w < wave; ++w) eq_base += sh.weq[w];
  }
  RSTAMP(7);
  // sweep 2: selected count per wave, min weight of the centres that stay
  int nsel = 0, eq_run = eq_base;
  double wmin = INFINITY;
#define RHCCQ_WSWEEP2(...)                                                                                                    \
  for (int i = 0; i < n_it; ++i) {                                                                                            \
    const int j = j0 + lane + 64 * i;                                                                                         \
    const double w = j < j1 ? ((kLds && !use_mask) ? (double)wl[j] : W[j]) : INFINITY;                                        \
    __VA_ARGS__                                                                                                               \
  }
  RHCCQ_WSWEEP2({
    const bool is_eq = capped && (w < thr) && (w == sel_w);
    const unsigned long long meq = __ballot(is_eq);
    const int rank = eq_run + __popcll(meq & ((1ull << lane) - 1ull));
    const bool sel = j < j1 && (use_mask ? (w < thr && ((qs->mask[j >> 5] >> (j & 31)) & 1u) != 0u) : reassign_sel(w, thr, capped, sel_w, take, rank));
    nsel += __popcll(__ballot(sel));
    if (j < j1 && !sel) wmin = fmin(wmin, w);
    eq_run += __popcll(meq);
  })
  if (lane == 0) sh.wsel[wave] = nsel;
  wmin = block_min_d(wmin, sh);                        // (two barriers: wsel is visible afterwards)
  int rbase = 0, n_re = 0;
  for (int w = 0; w < kUpdWaves; ++w) { if (w < wave) rbase += sh.wsel[w]; n_re += sh.wsel[w]; }
  RSTAMP(11);
  if (lane == 0) { rs->eq_base[wave] = eq_base; rs->rbase[wave] = rbase; }
  if (tid == 0) {
    rs->thr = thr; rs->sel_w = sel_w; rs->wmin = wmin;
    rs->capped = capped ? 1 : 0; rs->take = take; rs->n_re = n_re; rs->use_mask = use_mask ? 1 : 0;
    rs->tag_sel = (int)(step + 1);
  }
#undef RHCCQ_WSWEEP
#undef RHCCQ_WSWEEP2
}

// Where the draws of a launch go (the batches live in a ring of four buffers, batch b in ring[b & 3]):
//   role 1 draws the batches draw_first .. draw_first + draw_count - 1 (the classic sequence: the next step's batch; the overlapped
//   sequence of rhccq_mbk_steps_overlapped runs one batch ahead, so that the NEXT step's E-step can start beside this step's update);
//   a step that reassigns draws reassign_draws batches behind its choice() itself.
// expect_reassign >= 0: the host scheduled this launch for a step that does (1) / does not (0) reassign; a device state that
// disagrees stops the problem with code 5 before anything is modified.
struct UpdDraws {
  uint32_t* ring[4];
  long long draw_first;
  int draw_count, reassign_draws, expect_reassign, n_prob;
  int lds_weights;                   // RHCCQ_OPT_REASSIGN_LDS
  ReSel* resel;                      // [n_prob] what a reassigning step hands from the update kernel to mbk_reassign_apply_kernel
  unsigned long long* qs_e;          // RHCCQ_OPT_REASSIGN_ORDER = 1: scratch of npysort_head, indexed by koff ([sum k] each); nullptr = stable order
  int* qs_l;
  int* qs_r;
  unsigned* qs_mask;                 // [sum k / 32 + n_prob + 1]: problem p's words start at koff / 32 + p
};
__device__ __forceinline__ QsScratch qs_of(const UpdDraws& dr, const MbkP& P, int p) {
  return QsScratch{dr.qs_e + P.koff, dr.qs_l + P.koff, dr.qs_r + P.koff, dr.qs_mask + (P.koff >> 5) + p};
}

__device__ __forceinline__ void mbk_update_body(UpdShared& sh, const int p, const int role, const uint32_t* __restrict__ keys,
                                                const MbkP* __restrict__ probs, double* __restrict__ centres, double* __restrict__ weights,
                                                double* __restrict__ state, long long step, const uint32_t* __restrict__ words, long long n_words,
                                                const uint32_t* __restrict__ bkeys_cur, const UpdDraws& dr,
                                                const int32_t* __restrict__ labels_p) {
  const int tid = threadIdx.x;
  double* st = state + p * kStateStride;
  // independent table reads issued together: the kernel is a chain of dependent accesses, every cold miss counts
  const double st_since = st[st_slot(kStSince, step)], st_nzero = st[st_slot(kStNzero, step)];
  const double st_cursor = st[st_slot(kStCursor, role == 1 ? dr.draw_first - 1 : step)];
  const MbkP P = probs[p];
  if (mbk_stopped(st, step, P.n)) return;
  const int k = (int)P.k;
  const long long n = P.n;
  const int bs = (int)min((long long)1000, n);
  long long cursor = (long long)st_cursor;
  // reassignment decision uses the weights BEFORE this step's update (sklearn _random_reassign):
  // n_zero = number of zero-weight centres, maintained by the reassignment sweep (it can only be non-zero
  // while every step reassigns)
  double since = st_since + (double)bs;
  const bool do_reassign = st_nzero > 0.0 || since >= 10.0 * (double)k;
  if (do_reassign) since = 0.0;
#ifdef RHCCQ_STAMPS
  const unsigned long long _t_role = clock64();
#define ROLE_END(r) do { if (p == 0) atomicAdd(&g_upd_stamps[8 + (r)], clock64() - _t_role); } while (0)
#else
#define ROLE_END(r) do {} while (0)
#endif
  if (dr.expect_reassign >= 0 && (int)do_reassign != dr.expect_reassign) {   // the host's schedule and the device state disagree
    if (tid == 0) st[4] = 5.0;
    return;
  }
  if (role == 1) {
    if (do_reassign) {
      // ---- new_centers = random_state.choice(batch, replace=False, size=n_reassigns) = permutation(batch)[:n_reassigns]: the whole
      // shuffle is consumed however many rows are taken, so it is replayed beside the selection (role 0) for the batch / 2 rows
      // that may be needed; if nothing is reassigned the record is ignored (numpy then draws nothing)
      ReSel* rs = dr.resel + p;
      long long c = cursor;
      if (c + kWordsMargin > n_words) c = -1;
      else {
        for (int i = tid; i < kStageWords; i += kUpdThreads) sh.stage[i] = words[c + i];
        __syncthreads();
        c = replay_permutation(words, n_words, c, bs, bs / 2, sh, c);
        if (c >= 0 && tid < bs / 2) rs->perm[tid] = sh.perm[tid];
      }
      if (tid == 0) { rs->cursor_choice = c; rs->tag_choice = (int)(step + 1); }
      return;
    }
    // ---- the batches ahead: minibatch_indices = random_state.randint(0, n_samples, batch_size) ------------
    for (int q = 0; q < dr.draw_count; ++q) {
      const long long b = dr.draw_first + q;
      cursor = draw_batch(keys, P, words, n_words, cursor, dr.ring[b & 3] + (size_t)p * kBatch, sh.lab, sh.ired, &sh.cursor);
      if (cursor < 0) break;
      if (tid == 0) st[st_slot(kStCursor, b)] = (double)cursor;
      __syncthreads();                                   // (sh.lab / sh.cursor are reused by the next draw)
    }
    if (tid == 0) {
      if (cursor < 0) st[4] = 3.0;                       // word table exhausted (the host sizes it so that this cannot happen)
      ROLE_END(1);
    }
    return;
  }
  if (do_reassign && cursor + kWordsMargin > n_words) {   // before anything is modified
    if (tid == 0) st[4] = 3.0;
    return;
  }
#ifdef RHCCQ_STAMPS
  unsigned long long _ul = clock64();
  if (tid == 0 && p == 0) atomicAdd(&g_upd_stamps[15], 1ull);
#endif
  double* C = centres + P.koff * 4;
  double* W = weights + P.koff;
  const uint32_t* bkeys_p = bkeys_cur + (size_t)p * kBatch;
  for (int i = tid; i < kHashSlots; i += kUpdThreads) { sh.hkey[i] = -1; sh.hcnt[i] = 0; }
  // ---- labels of the batch (the E-step kernels leave the folded arg-min in the slot of tile 0) ------------
  double cb0 = 0.0, cb1 = 0.0, cb2 = 0.0, wb = 0.0;
  if (tid < bs) {
    const int bj = labels_p[tid];
    const uint32_t kk = bkeys_p[tid];
    sh.lab[tid] = bj;
    sh.bkey[tid] = kk;
    cb0 = C[bj * 4]; cb1 = C[bj * 4 + 1]; cb2 = C[bj * 4 + 2]; wb = W[bj];
  } else {
    sh.lab[tid] = -1;                                    // (walk_members reads whole groups of 16 rows)
  }
  USTAMP(0);
  __syncthreads();
  USTAMP(1);
  // ---- members of every touched centre (LDS hash keyed by centre) -------------------------------------
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
    // the centre and its weight were fetched for the labels: park them in the slot (every member of the cluster
    // writes the same four values), so that the update below does not go back to cold global memory
    sh.hold[h][0] = cb0; sh.hold[h][1] = cb1; sh.hold[h][2] = cb2; sh.hold[h][3] = wb;
    const int pos = atomicAdd(&sh.hcnt[h], 1);
    if (pos < kMemCap) sh.hmem[h][pos] = (unsigned short)tid;
  }
  __syncthreads();
  USTAMP(2);
  // ---- update_center_dense: c*w, += x for the members IN BATCH ORDER, w += count, c *= 1/w (each rounded once) ----
  {
    static_assert(kHashSlots == 2 * kUpdThreads, "apply loop is unrolled for two slots per thread");
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int h = tid + q * kUpdThreads, j = sh.hkey[h];
      if (j < 0) continue;
      const int cnt = sh.hcnt[h];
      const double w = sh.hold[h][3];
      double a0 = sh.hold[h][0] * w, a1 = sh.hold[h][1] * w, a2 = sh.hold[h][2] * w;
      if (cnt <= kMemCap) {
        int prev = -1;
        for (int i = 0; i < cnt; ++i) {                  // next member in ascending batch row
          int best = 0x7fffffff;
#pragma unroll
          for (int m = 0; m < kMemCap; ++m) {
            const int r = m < cnt ? (int)sh.hmem[h][m] : 0x7fffffff;
            if (r > prev && r < best) best = r;
          }
          prev = best;
          const uint32_t kk = sh.bkey[best];
          a0 = a0 + (double)key_r(kk); a1 = a1 + (double)key_g(kk); a2 = a2 + (double)key_b(kk);
        }
      } else {                                           // many members (small k): walk the batch
        walk_members(sh.lab, sh.bkey, bs, j, a0, a1, a2);
      }
      const double wn = w + (double)cnt;
      const double alpha = 1.0 / wn;
      const double c0 = a0 * alpha, c1 = a1 * alpha, c2 = a2 * alpha;
      C[j * 4] = c0; C[j * 4 + 1] = c1; C[j * 4 + 2] = c2;
      C[j * 4 + 3] = km64_csq(c0, c1, c2);
      W[j] = wn;
    }
  }
  __syncthreads();
  USTAMP(3);
  // ---- low-count reassignment (sklearn _mini_batch_step) ----------------------------------------------------------
  if (do_reassign) {
#ifdef RHCCQ_STAMPS
    if (tid == 0 && p == 0) atomicAdd(&g_upd_stamps[14], 1ull);
#endif
    const QsScratch qsv = qs_of(dr, P, p);
    const QsScratch* qs = dr.qs_e != nullptr ? &qsv : nullptr;
    if (dr.lds_weights && k <= kWLds && P.n <= (1ll << 24)) reassign_select<true>(sh, W, k, bs, dr.resel + p, step, p == 0, qs);
    else reassign_select<false>(sh, W, k, bs, dr.resel + p, step, p == 0, qs);
  }
  USTAMP(4);
  // ---- a step that reassigned draws the next batch itself (role 2 stood back) -------------------------------------
  // (a step that reassigns: the centres move, the zero-weight count and the next batch follow in mbk_reassign_apply_kernel)
  if (tid == 0) {
    st[st_slot(kStSince, step + 1)] = since;
    if (!do_reassign) st[st_slot(kStNzero, step + 1)] = st_nzero;        // (0: it stays 0)
  }
  USTAMP(5);
  if (tid == 0) ROLE_END(0);
}

__global__ __launch_bounds__(kUpdThreads) void mbk_update_kernel(const uint32_t* __restrict__ keys, const MbkP* __restrict__ probs,
                                                                 double* __restrict__ centres, double* __restrict__ weights,
                                                                 double* __restrict__ state, long long step,
                                                                 const uint32_t* __restrict__ words, long long n_words,
                                                                 const uint32_t* __restrict__ bkeys_cur, UpdDraws dr,
                                                                 const int32_t* __restrict__ pidx, const long long* __restrict__ part_off,
                                                                 const int32_t* __restrict__ lab) {
  __shared__ UpdShared sh;
  const int p = blockIdx.x;
  mbk_update_body(sh, p, (int)blockIdx.y, keys, probs, centres, weights, state, step, words, n_words, bkeys_cur, dr,
                  lab != nullptr ? lab + (size_t)p * kBatch : pidx + part_off[p]);
}

// Launch 2 of a reassigning step (see ReSel): role 0 moves the selected centres to their batch rows (sklearn _mini_batch_step:
// centers_new[to_reassign] = X[new_centers]; weight_sums[to_reassign] = min(weight_sums[~to_reassign])) and counts the centres
// that still have no weight; role 1 draws the next batch(es) behind the shuffle.  A step that did not reassign left no record
// with its tag: both roles return at once.
struct ApplyShared {
  int perm[kBatch / 2];
  uint32_t bkey[kBatch];
  int out[kBatch];
  int ired[kUpdWaves + 1];
  long long cursor;
};
__global__ __launch_bounds__(kUpdThreads) void mbk_reassign_apply_kernel(const uint32_t* __restrict__ keys, const MbkP* __restrict__ probs,
                                                                         double* __restrict__ centres, double* __restrict__ weights,
                                                                         double* __restrict__ state, long long step,
                                                                         const uint32_t* __restrict__ words, long long n_words,
                                                                         const uint32_t* __restrict__ bkeys_cur, UpdDraws dr) {
  __shared__ ApplyShared sh;
  const int p = blockIdx.x, role = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double* st = state + p * kStateStride;
  const MbkP P = probs[p];
  const ReSel* rs = dr.resel + p;
  const int tag_sel = rs->tag_sel, tag_choice = rs->tag_choice, n_re = rs->n_re;
  const long long cursor_choice = rs->cursor_choice;
  if (mbk_stopped(st, step, P.n) || tag_sel != (int)(step + 1)) return;
  const int k = (int)P.k;
  const int bs = (int)min((long long)1000, P.n);
  if (n_re > 0 && (tag_choice != (int)(step + 1) || cursor_choice < 0)) {      // the shuffle ran out of words (cannot happen
    if (tid == 0) st[4] = 3.0;                                                 // within kWordsMargin)
    return;
  }
  if (role == 1) {
    long long cursor = n_re > 0 ? cursor_choice : (long long)st[st_slot(kStCursor, step)];
    for (int q = 0; q < dr.reassign_draws && cursor >= 0; ++q) {
      const long long b = step + 1 + q;
      cursor = draw_batch(keys, P, words, n_words, cursor, dr.ring[b & 3] + (size_t)p * kBatch, sh.out, sh.ired, &sh.cursor);
      if (cursor >= 0 && tid == 0) st[st_slot(kStCursor, b)] = (double)cursor;
      __syncthreads();
    }
    if (tid == 0 && cursor < 0) st[4] = 3.0;
    return;
  }
  double* C = centres + P.koff * 4;
  double* W = weights + P.koff;
  const double thr = rs->thr, sel_w = rs->sel_w, wmin = rs->wmin;
  const bool capped = rs->capped != 0, use_mask = rs->use_mask != 0;
  const int take = rs->take;
  const unsigned* qmask = dr.qs_mask + (P.koff >> 5) + p;
  if (tid < bs / 2) sh.perm[tid] = n_re > 0 ? rs->perm[tid] : 0;
  if (tid < bs) sh.bkey[tid] = bkeys_cur[(size_t)p * kBatch + tid];
  __syncthreads();
  const int R = (((k + kUpdWaves - 1) / kUpdWaves) + 63) & ~63;
  const int j0 = wave * R, j1 = min(j0 + R, k);
  int nzero = 0, r_run = rs->rbase[wave], eq_run = rs->eq_base[wave];
  for (int jb = j0; jb < j1; jb += 64) {
    const int j = jb + lane;
    const double w = j < j1 ? W[j] : INFINITY;
    const bool is_eq = capped && (w < thr) && (w == sel_w);
    const unsigned long long meq = __ballot(is_eq);
    const int rank = eq_run + __popcll(meq & ((1ull << lane) - 1ull));
    const bool sel = j < j1 && (use_mask ? (w < thr && ((qmask[j >> 5] >> (j & 31)) & 1u) != 0u) : reassign_sel(w, thr, capped, sel_w, take, rank));
    const unsigned long long msel = __ballot(sel);
    double wf = w;
    if (sel) {
      // the i-th reassigned centre (ascending index) takes batch row perm[i]
      const uint32_t kk = sh.bkey[sh.perm[r_run + __popcll(msel & ((1ull << lane) - 1ull))]];
      const double n0 = (double)key_r(kk), n1 = (double)key_g(kk), n2 = (double)key_b(kk);
      C[j * 4] = n0; C[j * 4 + 1] = n1; C[j * 4 + 2] = n2;
      C[j * 4 + 3] = km64_csq(n0, n1, n2);
      W[j] = wmin;
      wf = wmin;
    }
    nzero += (j < j1) && (wf == 0.0);
    eq_run += __popcll(meq);
    r_run += __popcll(msel);
  }
  nzero = block_sum<int>(nzero, sh.ired);
  if (tid == 0) st[st_slot(kStNzero, step + 1)] = (double)nzero;
}

#include "k8_overlap.h"

__global__ __launch_bounds__(kUpdThreads) void npysort_head_kernel(const double* __restrict__ w, int k, int cap, int depth0, unsigned long long* e,
                                                                   int* lpos, int* rpos, unsigned* mask, int use_lds) {
  __shared__ UpdShared sh;
  void* lds = use_lds ? reinterpret_cast<void*>(reinterpret_cast<char*>(&sh) + kWLdsOff) : nullptr;      // the region reassign_select lends it
  npysort_head(sh, w, k, cap, QsScratch{e, lpos, rpos, mask}, depth0, lds, kWLds / 4);
}

// ------------------------------------------------------------------------------------------------
// final E-step over all points
// ------------------------------------------------------------------------------------------------
// ---- exact pruning of the final E-step: uniform 32^3 grid over the centres --------------------------
// A point searches the cells around its own in growing Chebyshev rings and stops once its best true
// distance is below the distance to anything outside the searched cube; every centre it skips is provably
// farther than the winner by a margin far above float64 rounding, so the result equals the brute-force
// first arg-min of csq_j + (-2 * dot) (ties between equal distances go to the smaller j explicitly).
constexpr int kGridG = 32, kGridCells = kGridG * kGridG * kGridG, kCellSide = 8;

__device__ __forceinline__ int grid_axis(double v) { return min(kGridG - 1, max(0, (int)(v * (1.0 / kCellSide)))); }

// One 1024-thread workgroup bins all centres of one problem with LDS atomics only (scattered device-scope
// atomics run at the memory side on this part and cost ~1 ms per step for 32 problems): count per cell, scan,
// fill.  The order inside a cell depends on scheduling; grid_nearest() breaks distance ties by centre index,
// so its result does not.
__global__ __launch_bounds__(1024) void grid_build_kernel(const MbkP* __restrict__ probs, const double* __restrict__ centres,
                                                          uint32_t* __restrict__ cell_start /* [n_prob][cells+1] */,
                                                          uint32_t* __restrict__ order, const double* __restrict__ state, long long step) {
  __shared__ uint32_t cnt[kGridCells];                   // counters, then fill cursors (128 KiB)
  __shared__ unsigned red[17];
  const int p = blockIdx.x;
  const MbkP P = probs[p];
  if (state && mbk_stopped(state + p * 16, step, P.n)) return;
  const double* C = centres + P.koff * 4;
  uint32_t* cs = cell_start + (size_t)p * (kGridCells + 1);
  for (int c = threadIdx.x; c < kGridCells; c += 1024) cnt[c] = 0;
  __syncthreads();
  // eight centres per thread in flight: the loads are issued together, then the LDS atomics
  constexpr int kU = 8;
  for (long long j0 = threadIdx.x; j0 < P.k; j0 += 1024 * kU) {
    int cell[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const long long j = j0 + (long long)u * 1024;
      cell[u] = -1;
      if (j < P.k) {
        const double* c = C + j * 4;
        cell[u] = (grid_axis(c[0]) * kGridG + grid_axis(c[1])) * kGridG + grid_axis(c[2]);
      }
    }
#pragma unroll
    for (int u = 0; u < kU; ++u)
      if (cell[u] >= 0) atomicAdd(&cnt[cell[u]], 1u);
  }
  __syncthreads();
  // exclusive scan of the counters in place: wave w owns 2048 consecutive cells and walks them 64 at a time
  // (consecutive lanes -> consecutive banks), carrying its running offset
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  constexpr int kPerWave = kGridCells / 16;
  unsigned sum = 0;
  for (int it = 0; it < kPerWave / 64; ++it) sum += cnt[w * kPerWave + it * 64 + lane];
  sum = wave_sum_u32(sum);
  if (lane == 0) red[w] = sum;
  __syncthreads();
  unsigned running = 0, tot = 0;
  for (int i = 0; i < 16; ++i) {
    if (i < w) running += red[i];
    tot += red[i];
  }
  for (int it = 0; it < kPerWave / 64; ++it) {
    const int idx = w * kPerWave + it * 64 + lane;
    const unsigned v = cnt[idx];
    const unsigned inc = wave_incscan_u32(v);
    cnt[idx] = running + inc - v;                          // cursor = start of the cell
    running += __shfl(inc, 63, 64);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < kGridCells; c += 1024) cs[c] = cnt[c];
  if (threadIdx.x == 0) cs[kGridCells] = tot;
  __syncthreads();
  for (long long j0 = threadIdx.x; j0 < P.k; j0 += 1024 * kU) {
    int cell[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const long long j = j0 + (long long)u * 1024;
      cell[u] = -1;
      if (j < P.k) {
        const double* c = C + j * 4;
        cell[u] = (grid_axis(c[0]) * kGridG + grid_axis(c[1])) * kGridG + grid_axis(c[2]);
      }
    }
#pragma unroll
    for (int u = 0; u < kU; ++u)
      if (cell[u] >= 0) order[P.koff + atomicAdd(&cnt[cell[u]], 1u)] = (uint32_t)(j0 + (long long)u * 1024);
  }
}

// nearest centre of one point through the grid: first arg-min of csq_j + (-2) <x, c_j> (ties -> smaller j)
__device__ __forceinline__ int grid_nearest(uint32_t kk, const uint32_t* __restrict__ cs, const uint32_t* __restrict__ ord,
                                            const double* __restrict__ C, double* best_out) {
  const double x0 = (double)key_r(kk), x1 = (double)key_g(kk), x2 = (double)key_b(kk);
  const double xsq = (x0 * x0 + x1 * x1) + x2 * x2;
  const int cx = (int)key_r(kk) / kCellSide, cy = (int)key_g(kk) / kCellSide, cz = (int)key_b(kk) / kCellSide;
  double bd = INFINITY;
  int bj = 0x7fffffff;
  for (int r = 0; r < kGridG; ++r) {
    // everything outside the cube of rings <= r-1 is at least (r-1)*8 + (distance to the own cell's wall) >=
    // (r - 1) * 8 away (r >= 1); 1e-6 dwarfs the float64 error of bd + xsq (values <= 4e5)
    if (r >= 2) {
      const double lb = (double)((r - 1) * kCellSide);
      if (bd + xsq <= lb * lb - 1e-6) break;
    }
    const int zlo = max(cz - r, 0), zhi = min(cz + r, kGridG - 1);
    for (int ix = max(cx - r, 0); ix <= min(cx + r, kGridG - 1); ++ix) {
      const int ax = abs(ix - cx);
      for (int iy = max(cy - r, 0); iy <= min(cy + r, kGridG - 1); ++iy) {
        const bool shell_xy = ax == r || abs(iy - cy) == r;
        // on the shell in x or y: the whole z range of the ring; otherwise only its two z faces
        for (int iz = zlo; iz <= zhi; iz += (shell_xy ? 1 : max(zhi - zlo, 1))) {
          if (!shell_xy && abs(iz - cz) != r) continue;
          const int cell = (ix * kGridG + iy) * kGridG + iz;
          const uint32_t e0 = cs[cell], e1 = cs[cell + 1];
          for (uint32_t e = e0; e < e1; ++e) {
            const int j = (int)ord[e];
            const double* c = C + (size_t)j * 4;
            const double d = c[3] + (-2.0 * km64_dot(x0, x1, x2, c[0], c[1], c[2]));
            if (d < bd || (d == bd && j < bj)) { bd = d; bj = j; }
          }
        }
      }
    }
  }
  if (best_out) *best_out = bd;
  return bj;
}

__global__ __launch_bounds__(256) void mbk_assign_grid_kernel(const uint32_t* __restrict__ keys, const MbkP* __restrict__ probs,
                                                              const double* __restrict__ centres, const uint32_t* __restrict__ cell_start,
                                                              const uint32_t* __restrict__ order, int32_t* __restrict__ labels,
                                                              const long long* __restrict__ blk_off) {
  int p = 0;
  while (blockIdx.x >= blk_off[p + 1]) ++p;
  const MbkP P = probs[p];
  const long long i = ((long long)blockIdx.x - blk_off[p]) * 256 + threadIdx.x;
  if (i >= P.n) return;
  labels[P.off + i] = grid_nearest(keys[P.off + i], cell_start + (size_t)p * (kGridCells + 1), order + P.koff, centres + P.koff * 4, nullptr);
}

// Wave-cooperative variant for the batch E-step: only 1000 points per problem are in flight there, so a
// thread-per-point walk would be a chain of dependent loads with nothing to hide their latency.  The 64 lanes
// share one point and take the cells of a ring side by side; same per-centre arithmetic, same arg-min.
__device__ __forceinline__ void grid_scan_shell(int r_lo, int r_hi, int cx, int cy, int cz, double x0, double x1, double x2,
                                                const uint32_t* __restrict__ cs, const uint32_t* __restrict__ ord,
                                                const double* __restrict__ C, double& bd, int& bj) {
  const int lane = threadIdx.x & 63;
  const int side = 2 * r_hi + 1, vol = side * side * side;
  for (int t = lane; t < vol; t += 64) {
    const int dx = t / (side * side) - r_hi, dy = (t / side) % side - r_hi, dz = t % side - r_hi;
    if (max(abs(dx), max(abs(dy), abs(dz))) < r_lo) continue;          // interior: searched in an earlier pass
    const int ix = cx + dx, iy = cy + dy, iz = cz + dz;
    if ((unsigned)ix >= (unsigned)kGridG || (unsigned)iy >= (unsigned)kGridG || (unsigned)iz >= (unsigned)kGridG) continue;
    const int cell = (ix * kGridG + iy) * kGridG + iz;
    const uint32_t e0 = cs[cell], e1 = cs[cell + 1];
    for (uint32_t e = e0; e < e1; ++e) {
      const int j = (int)ord[e];
      const double* c = C + (size_t)j * 4;
      const double d = c[3] + (-2.0 * km64_dot(x0, x1, x2, c[0], c[1], c[2]));
      if (d < bd || (d == bd && j < bj)) { bd = d; bj = j; }
    }
  }
  // wave arg-min, ties to the smaller centre index; every lane ends with the result
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double od = __shfl_xor(bd, o, 64);
    const int oj = __shfl_xor(bj, o, 64);
    if (od < bd || (od == bd && oj < bj)) { bd = od; bj = oj; }
  }
}

// batch E-step through the grid (many problems in flight: the brute-force E-step would be the bottleneck);
// writes the same (distance, label) the tiled kernel + tile reduction produce, into tile 0 of the partials
__global__ __launch_bounds__(256) void mbk_batch_estep_grid_kernel(const uint32_t* __restrict__ keys, const MbkP* __restrict__ probs,
                                                                   const double* __restrict__ centres, double* __restrict__ state,
                                                                   long long step, const uint32_t* __restrict__ bkeys,
                                                                   const double* __restrict__ pper_prev, const uint32_t* __restrict__ cell_start,
                                                                   const uint32_t* __restrict__ order, double* __restrict__ pdist,
                                                                   int32_t* __restrict__ pidx, const long long* __restrict__ part_off) {
  const int p = blockIdx.y;
  const MbkP P = probs[p];
  if (blockIdx.x == gridDim.x - 1) {                     // the extra workgroup: inertia + EWA rule of the PREVIOUS step
    __shared__ double s_per[kBatch];
    if (pper_prev != nullptr && !mbk_stopped(state + p * 16, step - 1, P.n))
      mbk_inertia_block(P, state + p * 16, step - 1, pper_prev + (size_t)p * kBatch, s_per);
    return;
  }
  if (mbk_stopped(state + p * 16, step, P.n)) return;
  const int bs = (int)min((long long)1000, P.n);
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);     // one wave per batch point
  if (b >= bs) return;
  const uint32_t kk = bkeys[(size_t)p * kBatch + b];
  const uint32_t* cs = cell_start + (size_t)p * (kGridCells + 1);
  const uint32_t* ord = order + P.koff;
  const double* C = centres + P.koff * 4;
  const double x0 = (double)key_r(kk), x1 = (double)key_g(kk), x2 = (double)key_b(kk);
  const double xsq = (x0 * x0 + x1 * x1) + x2 * x2;
  const int cx = (int)key_r(kk) / kCellSide, cy = (int)key_g(kk) / kCellSide, cz = (int)key_b(kk) / kCellSide;
  double bd = INFINITY;
  int bj = 0x7fffffff;
  grid_scan_shell(0, 1, cx, cy, cz, x0, x1, x2, cs, ord, C, bd, bj);   // rings 0 and 1: the 27-cell cube
  for (int r = 2; r < kGridG; ++r) {
    const double lb = (double)((r - 1) * kCellSide);     // same stop rule as grid_nearest()
    if (bd + xsq <= lb * lb - 1e-6) break;
    grid_scan_shell(r, r, cx, cy, cz, x0, x1, x2, cs, ord, C, bd, bj);
  }
  if ((threadIdx.x & 63) == 0) {
    pdist[part_off[p] + b] = bd;
    pidx[part_off[p] + b] = bj;
  }
}

// ---- internal pruning order of the init samples: (Morton code of the colour, draw position) sort keys ----------
__device__ __forceinline__ uint32_t spread3(uint32_t v) {
  v &= 0xFFu;
  v = (v | (v << 16)) & 0xFF0000FFu;
  v = (v | (v << 8)) & 0x0F00F00Fu;
  v = (v | (v << 4)) & 0xC30C30C3u;
  v = (v | (v << 2)) & 0x49249249u;
  return v;
}
__global__ __launch_bounds__(256) void sample_sortkey_kernel(const uint32_t* __restrict__ keys, const MbkP* __restrict__ probs, int n_prob,
                                                             const int32_t* __restrict__ init_idx, unsigned long long* __restrict__ out,
                                                             long long base, long long total) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;      // sample base + t of the chunk's first problem
  if (t >= total) return;
  int p = 0;
  while (p + 1 < n_prob && probs[p + 1].init_off <= base + t) ++p;
  const uint32_t idx = (uint32_t)init_idx[base + t];
  const uint32_t kk = keys[probs[p].off + idx];
  const uint32_t m = (spread3(kk >> 16) << 2) | (spread3(kk >> 8) << 1) | spread3(kk);
  out[t] = ((unsigned long long)p << 56) | ((unsigned long long)m << 32) | (uint32_t)(base + t - probs[p].init_off);
}
__global__ __launch_bounds__(256) void sample_unpack_kernel(const unsigned long long* __restrict__ sorted, int32_t* __restrict__ perm,
                                                            long long base, long long total) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t < total) perm[base + t] = (int32_t)(uint32_t)sorted[t];
}

// ---- uniform(size=count) of numpy's legacy RandomState from the resident raw MT19937 words (mt.py) ---------
__global__ __launch_bounds__(256) void mt_uniforms_kernel(const uint32_t* __restrict__ words, long long pos, long long count,
                                                          double* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  const uint2 w = *reinterpret_cast<const uint2*>(words + pos + 2 * i);
  out[i] = ((double)(w.x >> 5) * 67108864.0 + (double)(w.y >> 6)) / 9007199254740992.0;
}
__global__ __launch_bounds__(256) void mt_uniforms_odd_kernel(const uint32_t* __restrict__ words, long long pos, long long count,
                                                              double* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  const uint32_t a = words[pos + 2 * i], b = words[pos + 2 * i + 1];
  out[i] = ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

// hipFree waits for the whole device (every stream, also the other lanes of a pipelined call: a k-means++ chain in flight
// there costs 200 ms), so the scratch starts generous and doubles: a steady stream of frames never reallocates
static int ensure_scratch(rhccq_ctx* ctx, size_t bytes) {
  if (ctx->scratch_bytes >= bytes) return 0;
  size_t want = ctx->scratch_bytes ? 2 * ctx->scratch_bytes : ((size_t)32 << 20);
  while (want < bytes) want *= 2;
  if (ctx->scratch) RHCCQ_HIP(ctx, hipFree(ctx->scratch));
  ctx->scratch = nullptr;
  ctx->scratch_bytes = 0;
  RHCCQ_HIP(ctx, hipMalloc(&ctx->scratch, want));
  ctx->scratch_bytes = want;
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
//                               [pdist f64[sum tiles*1024]] [pidx i32[sum tiles*1024]] [grid tables] [bkeys u32[2][n_prob*1024]] [pper f64[n_prob*1024]]
int64_t rhccq_mbk_work_bytes(const rhccq_mbk_problem* probs, int32_t n_prob) {
  if (!probs || n_prob <= 0) return 0;
  size_t part = 0;
  for (int i = 0; i < n_prob; ++i) part += (size_t)((probs[i].k + kTileS - 1) / kTileS) * kBatch;   // (the speculative E-step's tiles are half as wide)
  size_t bytes = align256(sizeof(MbkP) * n_prob) + align256(8 * (size_t)n_prob) + align256(8 * (size_t)(n_prob + 1));
  bytes += align256(part * 8) + align256(part * 4);
  size_t ksum = 0;
  for (int i = 0; i < n_prob; ++i) ksum += (size_t)probs[i].k;
  bytes += align256((size_t)n_prob * (kGridCells + 1) * 4) + align256((size_t)n_prob * kGridCells * 4) + align256(ksum * 4);
  bytes += 2 * align256((size_t)n_prob * kBatch * 4) + 2 * align256((size_t)n_prob * kBatch * 8);
  bytes += 4 * align256((size_t)n_prob * kBatch * 4);     // batches 2, 3 of the ring; the labels of even / odd steps
  bytes += align256((size_t)n_prob * sizeof(ReSel));        // what a reassigning step hands to its second launch
  bytes += align256(ksum * 8) + 2 * align256(ksum * 4) + align256((ksum / 32 + (size_t)n_prob + 1) * 4);   // npysort_head (k8_npysort.h)
  return (int64_t)bytes;
}

struct WorkView {
  MbkP* probs;
  long long* part_off;
  long long* blk_off;
  double* pdist;
  int32_t* pidx;
  uint32_t* cell_start;   // [n_prob][cells + 1]
  uint32_t* cursor;       // [n_prob][cells] (spare)
  uint32_t* order;        // [sum k] centre indices grouped by cell (problem-relative)
  uint32_t* bkeys[4];     // [n_prob][1024] colours of the batch rows of step s in bkeys[s & 3] (written by the draws)
  double* pper[2];        // [n_prob][1024] the rows' inertia terms of even / odd steps (fold kernel -> mbk_inertia_block)
  int32_t* lab[2];        // [n_prob][1024] labels of even / odd steps (overlapped sequence: mbk_fix_kernel)
  ReSel* resel;           // [n_prob]
  unsigned long long* qs_e;   // [sum k] scratch of npysort_head
  int* qs_l;              // [sum k]
  int* qs_r;              // [sum k]
  unsigned* qs_mask;      // [sum k / 32 + n_prob + 1]
  long long max_k;
};

// upload = false: the tables at the head of `work` are those an earlier call of the same sequence put there (step0 > 0): no
// copies, no stream synchronisation -- the steps of a chunk then queue behind the previous chunk's without a bubble
static int layout_work(rhccq_ctx* ctx, const rhccq_mbk_problem* probs, int n_prob, void* work, int64_t work_bytes, WorkView* v,
                       long long* total_blocks, int* max_tiles, bool upload = true) {
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
    part += (size_t)((q.k + kTileS - 1) / kTileS) * kBatch;
    hbo[i] = blocks;
    blocks += (q.n + 255) / 256;
  }
  hbo[n_prob] = blocks;
  v->pdist = (double*)base; base += align256(part * 8);
  v->pidx = (int32_t*)base; base += align256(part * 4);
  v->cell_start = (uint32_t*)base; base += align256((size_t)n_prob * (kGridCells + 1) * 4);
  v->cursor = (uint32_t*)base; base += align256((size_t)n_prob * kGridCells * 4);
  v->order = (uint32_t*)base;
  {
    size_t ksum = 0;
    for (int i = 0; i < n_prob; ++i) ksum += (size_t)probs[i].k;
    base += align256(ksum * 4);
  }
  v->bkeys[0] = (uint32_t*)base; base += align256((size_t)n_prob * kBatch * 4);
  v->bkeys[1] = (uint32_t*)base; base += align256((size_t)n_prob * kBatch * 4);
  v->pper[0] = (double*)base; base += align256((size_t)n_prob * kBatch * 8);
  v->pper[1] = (double*)base; base += align256((size_t)n_prob * kBatch * 8);
  v->bkeys[2] = (uint32_t*)base; base += align256((size_t)n_prob * kBatch * 4);
  v->bkeys[3] = (uint32_t*)base; base += align256((size_t)n_prob * kBatch * 4);
  v->lab[0] = (int32_t*)base; base += align256((size_t)n_prob * kBatch * 4);
  v->lab[1] = (int32_t*)base; base += align256((size_t)n_prob * kBatch * 4);
  v->resel = (ReSel*)base; base += align256((size_t)n_prob * sizeof(ReSel));
  {
    size_t ksum = 0;
    for (int i = 0; i < n_prob; ++i) ksum += (size_t)probs[i].k;
    v->qs_e = (unsigned long long*)base; base += align256(ksum * 8);
    v->qs_l = (int*)base; base += align256(ksum * 4);
    v->qs_r = (int*)base; base += align256(ksum * 4);
    v->qs_mask = (unsigned*)base;
  }
  v->max_k = 0;
  for (int i = 0; i < n_prob; ++i) v->max_k = probs[i].k > v->max_k ? probs[i].k : v->max_k;
  if (upload) {
    if (int e = put(ctx, v->probs, hp, sizeof(MbkP) * n_prob)) return e;
    if (int e = put(ctx, v->part_off, hpo, 8 * (size_t)n_prob)) return e;
    if (int e = put(ctx, v->blk_off, hbo, 8 * (size_t)(n_prob + 1))) return e;
    RHCCQ_HIP(ctx, hipMemsetAsync(v->resel, 0, sizeof(ReSel) * (size_t)n_prob, ctx->stream));   // (no step has the tag 0)
    // the staging string dies at return: make sure the copies have been issued from it
    RHCCQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  *total_blocks = blocks;
  *max_tiles = mt;
  return 0;
}

#ifdef RHCCQ_STAMPS
int rhccq_debug_upd_stamps(unsigned long long* out16_host) {
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  if (hipMemcpyFromSymbol(out16_host, HIP_SYMBOL(g_upd_stamps), sizeof(unsigned long long) * 16) != hipSuccess) return -2;
  return 0;
}
int rhccq_debug_wave_stamps(unsigned long long* out64_host) {
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  if (hipMemcpyFromSymbol(out64_host, HIP_SYMBOL(g_wave_stamps), sizeof(unsigned long long) * 128) != hipSuccess) return -2;
  return 0;
}
int rhccq_debug_pipe_stamps(unsigned long long* out8_host) {
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  if (hipMemcpyFromSymbol(out8_host, HIP_SYMBOL(g_pipe_stamps), sizeof(unsigned long long) * 8) != hipSuccess) return -2;
  return 0;
}
int rhccq_debug_spec_stamps(unsigned long long* out8_host) {
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  if (hipMemcpyFromSymbol(out8_host, HIP_SYMBOL(g_spec_stamps), sizeof(unsigned long long) * 8) != hipSuccess) return -2;
  return 0;
}
int rhccq_debug_stamps(unsigned long long* out16_host) {
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  if (hipMemcpyFromSymbol(out16_host, HIP_SYMBOL(g_init_stamps), sizeof(unsigned long long) * 16) != hipSuccess) return -2;
  return 0;
}
#endif

// ---- Morton order of the init samples (internal pruning index of rhccq_mbk_init) ------------------------------
// sort key = problem << 56 | Morton code of the sampled colour << 32 | draw position: one device radix sort
// (rocPRIM through hipCUB) per chunk of <= 256 problems orders their samples at once
int64_t rhccq_mbk_order_bytes(int64_t total) { return total <= 0 ? 0 : 16 * total + (8ll << 20); }

int rhccq_mbk_order(rhccq_ctx* ctx, const uint32_t* keys, const rhccq_mbk_problem* probs, int32_t n_prob, const int32_t* init_idx,
                    int32_t* perm, void* tmp, int64_t tmp_bytes) {
  if (!ctx || !keys || !probs || !init_idx || !perm || !tmp || n_prob <= 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "mbk_order: bad argument");
  long long grand = 0;
  for (int i = 0; i < n_prob; ++i) {
    const rhccq_mbk_problem& q = probs[i];
    if (q.init_n <= 0 || q.init_off != grand || q.n <= 0 || q.n > 0x7fffffffll) return rhccq_fail(ctx, RHCCQ_E_ARG, "mbk_order: bad problem");
    grand += q.init_n;
  }
  if (grand > 0x7fffffffll) return rhccq_fail(ctx, RHCCQ_E_LIMIT, "mbk_order: too many samples");
  if (tmp_bytes < rhccq_mbk_order_bytes(grand)) return rhccq_fail(ctx, RHCCQ_E_ARG, "mbk_order: tmp too small");
  constexpr int kChunk = 256;                              // 8 bits of the sort key name the problem
  if (int e = ensure_scratch(ctx, align256(sizeof(MbkP) * kChunk))) return e;
  MbkP* dp = (MbkP*)ctx->scratch;
  std::string stage;
  stage.resize(sizeof(MbkP) * (size_t)kChunk);
  MbkP* hp = (MbkP*)stage.data();
  for (int c0 = 0; c0 < n_prob; c0 += kChunk) {
    const int nc = n_prob - c0 < kChunk ? n_prob - c0 : kChunk;
    long long total = 0;
    for (int i = 0; i < nc; ++i) {
      const rhccq_mbk_problem& q = probs[c0 + i];
      hp[i] = MbkP{q.off, q.n, q.k, q.koff, q.init_off, q.init_n, q.rand_off, q.first, q.T};
      total += q.init_n;
    }
    const long long base = probs[c0].init_off;
    // the previous chunk's kernels may still read the table: the copy is stream ordered behind them
    if (int e = put(ctx, dp, hp, sizeof(MbkP) * nc)) return e;
    RHCCQ_HIP(ctx, hipStreamSynchronize(ctx->stream));      // `stage` is pageable host memory
    unsigned long long* a = (unsigned long long*)tmp;
    unsigned long long* b = a + total;
    void* cub_tmp = (void*)(b + total);
    size_t cub_avail = (size_t)tmp_bytes - 16 * (size_t)total, cub_need = 0;
    hipcub::DoubleBuffer<unsigned long long> buf(a, b);
    int top = 56;
    while ((1 << (top - 56)) < nc) ++top;
    RHCCQ_HIP(ctx, hipcub::DeviceRadixSort::SortKeys(nullptr, cub_need, buf, (int)total, 0, top, ctx->stream));
    if (cub_need > cub_avail) return rhccq_fail(ctx, RHCCQ_E_LIMIT, "mbk_order: sort scratch exceeds tmp");
    const unsigned grid = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(sample_sortkey_kernel, dim3(grid), dim3(256), 0, ctx->stream, keys, dp, nc, init_idx, a, base, total);
    RHCCQ_HIP(ctx, hipcub::DeviceRadixSort::SortKeys(cub_tmp, cub_need, buf, (int)total, 0, top, ctx->stream));
    hipLaunchKernelGGL(sample_unpack_kernel, dim3(grid), dim3(256), 0, ctx->stream, buf.Current(), perm, base, total);
    RHCCQ_LAUNCH_CHECK(ctx);
  }
  return 0;
}

// RandomState.randint(0, n, size) of numpy's legacy generator replayed ON THE HOST from raw MT19937 words (host memory): masked
// rejection, one word per attempt (_bounded_integers.pyx, legacy path for ranges below 2^32).  out (int32[size], may be NULL when
// only the stream position matters: sklearn's validation draw) receives the values; returns the words consumed, -1 when the
// table ends first, -2 for a bad argument.  No HIP call inside: the k-means++ set-up of a frame's problems runs on several host
// threads and this loop, unlike its numpy twin (mt.py), does not hold the interpreter lock.
int64_t rhccq_mt_randint_host(const uint32_t* words, int64_t n_words, int64_t pos, int64_t n, int64_t size, int32_t* out) {
  if (!words || pos < 0 || n <= 0 || n > 0x7fffffffll || size < 0 || n_words < 0) return -2;
  if (size == 0) return 0;
  const uint32_t rng = (uint32_t)(n - 1);
  if (rng == 0u) {                                         // numpy draws nothing for a one-value range
    if (out) memset(out, 0, sizeof(int32_t) * (size_t)size);
    return 0;
  }
  uint32_t mask = rng;
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
  int64_t got = 0;
  if (out) {
    for (int64_t i = pos; i < n_words; ++i) {
      const uint32_t v = words[i] & mask;
      out[got] = (int32_t)v;                               // (kept only when accepted: `got` moves on)
      got += v <= rng;
      if (got == size) return i + 1 - pos;
    }
  } else {
    for (int64_t i = pos; i < n_words; ++i) {
      got += (words[i] & mask) <= rng;
      if (got == size) return i + 1 - pos;
    }
  }
  return -1;
}

int rhccq_mt_uniforms(rhccq_ctx* ctx, const uint32_t* words, int64_t pos, int64_t count, double* out) {
  if (!ctx || !words || !out || pos < 0 || count < 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "mt_uniforms: bad argument");
  if (count == 0) return 0;
  const unsigned grid = (unsigned)((count + 255) / 256);
  if ((pos & 1) == 0 && ((uintptr_t)words & 7u) == 0)
    hipLaunchKernelGGL(mt_uniforms_kernel, dim3(grid), dim3(256), 0, ctx->stream, words, (long long)pos, (long long)count, out);
  else
    hipLaunchKernelGGL(mt_uniforms_odd_kernel, dim3(grid), dim3(256), 0, ctx->stream, words, (long long)pos, (long long)count, out);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_mbk_init(rhccq_ctx* ctx, const uint32_t* keys, const rhccq_mbk_problem* probs, int32_t n_prob, const int32_t* init_idx,
                   const int32_t* perm, const double* rand, double* centres, int32_t* chosen) {
  if (!ctx || !keys || !probs || !init_idx || !perm || !rand || !centres || !chosen || n_prob <= 0)
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
    words += 5 * nb * 64 + 5 * nb + 5 * ((nb + 15) / 16) + 8;       // samp, dsamp (uint2 each), mperm, tables
    words = (words + 63) & ~(size_t)63;
  }
  const int lds_blocks = ctx->opt_init_lds_blocks < 0 ? kInitLdsBlocks : ctx->opt_init_lds_blocks;
  const int max_items = ctx->opt_init_max_items < 0 ? kMaxItems : ctx->opt_init_max_items;
  // second-generation chain (8 waves, tables in LDS) whenever every problem's block tables fit; the first generation keeps
  // the tables of larger problems (more than 262 144 init samples) in global memory
  bool lean = ctx->opt_init_kernel != 1;
  for (int i = 0; i < n_prob && lean; ++i) lean = (probs[i].init_n + 63) / 64 <= lds_blocks;
  // third generation (leaves of 16 samples, k8_init3.h) whenever every problem's leaf table fits LDS: init samples <= 98 304,
  // the regime of 4K frames; RHCCQ_OPT_INIT_KERNEL = 2 keeps the second generation (A/B runs, parity tests)
  bool gen3 = lean && (ctx->opt_init_kernel == 0 || ctx->opt_init_kernel == 3 || ctx->opt_init_kernel == 5) && ctx->opt_init_shards <= 1 && lds_blocks >= kInitLdsBlocks;
  for (int i = 0; i < n_prob && gen3; ++i) gen3 = probs[i].init_n <= kG3MaxSamples;
  // brute force with the samples in registers (kpp_flat.h): RHCCQ_OPT_INIT_KERNEL = 4 only.  MEASURED per pick: alone on the
  // chip 3.2 us at 3 000 init samples (every problem with k <= 3 000) against the third generation's 3.5-3.7, 3.8-4.0 at
  // 3 600-6 000; INSIDE a 4K frame 5.2 us (profiles/r03: its 512 threads share their CU with the other class's step kernels,
  // the third generation's 144 KB of LDS keep a CU to itself) -- so the block-tree chain stays the default at every size and
  // this one serves KMeans (k7_kmeans.hip, where it replaced a six-barrier pick) and as a cross-check
  bool flat = ctx->opt_init_kernel == 4 && ctx->opt_init_shards <= 1 && lds_blocks >= kInitLdsBlocks && max_items >= kMaxItems;
  long long flat_n = 0;
  for (int i = 0; i < n_prob && flat; ++i) {
    flat = probs[i].init_n <= kFlatMaxSamples;
    flat_n = probs[i].init_n > flat_n ? probs[i].init_n : flat_n;
  }
  if (ctx->opt_init_kernel == 4 && !flat) return rhccq_fail(ctx, RHCCQ_E_LIMIT, "mbk_init: RHCCQ_OPT_INIT_KERNEL = 4 takes at most 8192 init samples per problem and the default work-list / table / shard options");
  // sharded chain: C workgroups per problem, all resident at once (they wait for each other), so only for a handful of
  // problems; every shard must own at least one draw super-block and the pick number must fit the 24-bit granule tag
  // MEASURED (MI355X, k = 30 000, 90 000 init samples): 4.84 us per pick on one workgroup, 7.7-8.0 us on 2, 4 or 8 -- each of
  // the two exchanges costs ~3 us with a dozen waves of the receiving CU polling (the price of a hand-off sits in the consumer
  // CU's memory queue), more than the sharded phases save.  Hence opt-in only (RHCCQ_OPT_INIT_SHARDS = 2 / 4 / 8).
  int nshard = 1;
  if (lean && ctx->opt_init_shards > 1) {
    for (int c2 = ctx->opt_init_shards; c2 >= 2 && nshard == 1; c2 >>= 1) {
      bool ok = (long long)n_prob * c2 <= 64;
      for (int i = 0; i < n_prob && ok; ++i) {
        const long long d = ((probs[i].init_n + c2 - 1) / c2 + 1023) & ~1023ll;
        ok = d * (c2 - 1) < probs[i].init_n && probs[i].k < (1 << 24) - 2 && probs[i].init_n >= 4096ll * c2;
      }
      if (ok) nshard = c2;
    }
  }
  const int n_wg = n_prob * nshard;
  if (nshard > 1) {                                        // one sample area per (problem, shard)
    stage.resize(sizeof(MbkP) * (size_t)n_prob + 8 * (size_t)n_wg);
    hp = (MbkP*)stage.data();
    ho = (long long*)(stage.data() + sizeof(MbkP) * n_prob);
    words = 0;
    for (int i = 0; i < n_prob; ++i) {
      const long long d = ((probs[i].init_n + nshard - 1) / nshard + 1023) & ~1023ll;
      for (int s2 = 0; s2 < nshard; ++s2) {
        ho[i * nshard + s2] = (long long)words;
        words += 4 * (size_t)d + 64;                        // samp, dsamp (uint2 each)
      }
    }
  }
  const size_t xbytes = nshard > 1 ? align256(sizeof(JExchange) * (size_t)n_prob) : 0;
  const size_t head = align256(sizeof(MbkP) * n_prob) + align256(8 * (size_t)n_wg) + xbytes;
  if (int e = ensure_scratch(ctx, head + words * 4)) return e;
  char* base = (char*)ctx->scratch;
  MbkP* dp = (MbkP*)base;
  long long* dof = (long long*)(base + align256(sizeof(MbkP) * n_prob));
  JExchange* xch = (JExchange*)(base + align256(sizeof(MbkP) * n_prob) + align256(8 * (size_t)n_wg));
  uint32_t* dscr = (uint32_t*)(base + head);
  if (int e = put(ctx, dp, hp, sizeof(MbkP) * n_prob)) return e;
  if (int e = put(ctx, dof, ho, 8 * (size_t)n_wg)) return e;
  if (nshard > 1) RHCCQ_HIP(ctx, hipMemsetAsync(xch, 0, xbytes, ctx->stream));
  RHCCQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (flat) {
#define RHCCQ_FLAT_LAUNCH(S_) hipLaunchKernelGGL(mbk_init_flat_kernel<S_>, dim3(n_prob), dim3(kFlatThreads), 0, ctx->stream, keys, dp, init_idx, rand, centres, chosen)
    if (flat_n <= 6 * kFlatThreads) RHCCQ_FLAT_LAUNCH(6);          // (3 000 samples: every problem with k <= 1 000)
    else if (flat_n <= 8 * kFlatThreads) RHCCQ_FLAT_LAUNCH(8);
    else if (flat_n <= 10 * kFlatThreads) RHCCQ_FLAT_LAUNCH(10);
    else if (flat_n <= 12 * kFlatThreads) RHCCQ_FLAT_LAUNCH(12);
    else RHCCQ_FLAT_LAUNCH(16);
#undef RHCCQ_FLAT_LAUNCH
  } else if (gen3) {
    const int mi = max_items < kG3MaxItems ? max_items : kG3MaxItems;
    int cw = ctx->opt_init_cands_per_wave;
    bool t16 = false;                                      // (more than 12 trials cannot occur below 98 304 samples; a fourth candidate per wave is not built)
    for (int i = 0; i < n_prob; ++i) t16 = t16 || probs[i].T > 12;
    if (t16) cw = 1;
    bool lds_s = cw == 1;
    for (int i = 0; i < n_prob && lds_s; ++i) lds_s = probs[i].init_n <= kG3LdsSamples;
    const bool in_wave = cw == 1 && ctx->opt_init_kernel == 5;      // RHCCQ_OPT_INIT_KERNEL = 5: evaluation inside the candidate's own wave (measured slower)
    if (lds_s && in_wave)
      hipLaunchKernelGGL((mbk_init3_kernel<1, true, true>), dim3(n_prob), dim3(kG3Threads), 0, ctx->stream, keys, dp, init_idx, perm, rand, centres, chosen, dscr, dof, mi);
    else if (lds_s)
      hipLaunchKernelGGL((mbk_init3_kernel<1, true>), dim3(n_prob), dim3(kG3Threads), 0, ctx->stream, keys, dp, init_idx, perm, rand, centres, chosen, dscr, dof, mi);
    else if (in_wave)
      hipLaunchKernelGGL((mbk_init3_kernel<1, false, true>), dim3(n_prob), dim3(kG3Threads), 0, ctx->stream, keys, dp, init_idx, perm, rand, centres, chosen, dscr, dof, mi);
    else if (cw == 3)
      hipLaunchKernelGGL(mbk_init3_kernel<3>, dim3(n_prob), dim3(kG3Threads), 0, ctx->stream, keys, dp, init_idx, perm, rand, centres, chosen, dscr, dof, mi);
    else if (cw == 2)
      hipLaunchKernelGGL(mbk_init3_kernel<2>, dim3(n_prob), dim3(kG3Threads), 0, ctx->stream, keys, dp, init_idx, perm, rand, centres, chosen, dscr, dof, mi);
    else
      hipLaunchKernelGGL(mbk_init3_kernel<1>, dim3(n_prob), dim3(kG3Threads), 0, ctx->stream, keys, dp, init_idx, perm, rand, centres, chosen, dscr, dof, mi);
  }
  else if (nshard > 1)
    hipLaunchKernelGGL(mbk_init2_kernel<true>, dim3(n_wg), dim3(kJThreads), 0, ctx->stream, keys, dp, init_idx, perm, rand, centres, chosen, dscr,
                       dof, max_items < kJMaxItems ? max_items : kJMaxItems, nshard, xch);
  else if (lean)
    hipLaunchKernelGGL(mbk_init2_kernel<false>, dim3(n_prob), dim3(kJThreads), 0, ctx->stream, keys, dp, init_idx, perm, rand, centres, chosen, dscr,
                       dof, max_items < kJMaxItems ? max_items : kJMaxItems, 1, (JExchange*)nullptr);
  else
    hipLaunchKernelGGL(mbk_init_kernel, dim3(n_prob), dim3(kInitThreads), 0, ctx->stream, keys, dp, init_idx, perm, rand, centres, chosen, dscr, dof,
                       lds_blocks, max_items);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_npysort_head(rhccq_ctx* ctx, const double* w, int32_t k, int32_t cap, int32_t depth0, int32_t use_lds, void* scratch, uint32_t* mask_out) {
  if (!ctx || !w || !scratch || !mask_out || k < 2 || cap < 1 || cap >= k) return rhccq_fail(ctx, RHCCQ_E_ARG, "npysort_head: bad argument");
  unsigned long long* e = (unsigned long long*)scratch;
  int* lpos = (int*)(e + k);
  hipLaunchKernelGGL(npysort_head_kernel, dim3(1), dim3(kUpdThreads), 0, ctx->stream, w, (int)k, (int)cap, (int)depth0, e, lpos, lpos + k, mask_out, (int)use_lds);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_mbk_steps(rhccq_ctx* ctx, const uint32_t* keys, const rhccq_mbk_problem* probs, int32_t n_prob, int64_t step0, int32_t n_steps,
                    const uint32_t* words, int64_t n_words, double* centres, double* weights, double* state, void* work,
                    int64_t work_bytes, int32_t estep_mode, int32_t estep_split) {
  if (!ctx || !keys || !probs || !words || !centres || !weights || !state || !work || n_prob <= 0 || n_steps < 0 || n_words <= 0 || step0 < 0)
    return rhccq_fail(ctx, RHCCQ_E_ARG, "mbk_steps: bad argument");
  WorkView v;
  long long blocks;
  int max_tiles;
  if (int e = layout_work(ctx, probs, n_prob, work, work_bytes, &v, &blocks, &max_tiles, step0 == 0)) return e;
  // many problems in flight (a batch of frames): the brute-force E-step (sum k x 1000 float64 distance
  // evaluations per step) would dominate, so the centres are re-binned every step and the batch is assigned
  // through the grid; with few problems the tiled brute-force kernel has fewer launches per step
  long long ksum = 0;
  for (int i = 0; i < n_prob; ++i) ksum += probs[i].k;
  // RHCCQ_STEPS_NO_REASSIGN: the caller has read the state and knows that no running problem reassigns in this call (no centre
  // without weight, fewer than 10 k samples since the last reassignment throughout): the second launch of a reassigning step is
  // left out; a problem that wants to reassign all the same stops with code 5
  const bool no_reassign = (estep_mode & RHCCQ_STEPS_NO_REASSIGN) != 0;
  estep_mode &= ~RHCCQ_STEPS_NO_REASSIGN;
  if (estep_mode < RHCCQ_ESTEP_AUTO || estep_mode > RHCCQ_ESTEP_GRID) return rhccq_fail(ctx, RHCCQ_E_ARG, "mbk_steps: bad estep_mode");
  if (estep_split != 0 && estep_split != 1 && estep_split != 2 && estep_split != 4 && estep_split != 8)
    return rhccq_fail(ctx, RHCCQ_E_ARG, "mbk_steps: estep_split must be 0, 1, 2, 4 or 8");
  const bool use_grid = estep_mode == RHCCQ_ESTEP_GRID || (estep_mode == RHCCQ_ESTEP_AUTO && ksum >= 200000);
  // problems whose first batch has not been drawn yet (state[10] == 0) draw it now; the others return at once
  if (step0 == 0)
    hipLaunchKernelGGL(mbk_draw0_kernel, dim3(n_prob), dim3(kUpdThreads), 0, ctx->stream, keys, v.probs, state, words, (long long)n_words,
                       v.bkeys[0], centres);
  for (int s = 0; s < n_steps; ++s) {
    const long long step = step0 + s;
    const uint32_t* bk = v.bkeys[step & 3];
    // the previous step's inertia terms (none for the first step of this call: the last step of a call has a kernel of its own)
    const double* pprev = s > 0 ? (const double*)v.pper[(step - 1) & 1] : (const double*)nullptr;
    if (use_grid) {
      hipLaunchKernelGGL(grid_build_kernel, dim3(n_prob), dim3(1024), 0, ctx->stream, v.probs, centres, v.cell_start, v.order,
                         (const double*)state, step);
      hipLaunchKernelGGL(mbk_batch_estep_grid_kernel, dim3(250 + 1, n_prob), dim3(256), 0, ctx->stream, keys, v.probs, centres, state, step, bk,
                         pprev, v.cell_start, v.order, v.pdist, v.pidx, v.part_off);
    } else {
#define RHCCQ_ESTEP_LAUNCH(SS)                                                                                                             \
  hipLaunchKernelGGL(mbk_batch_estep_kernel<SS>, dim3(max_tiles * kPtChunks * SS + 1, n_prob), dim3(256), 0, ctx->stream, keys, v.probs, centres, \
                     state, step, bk, v.pdist, v.pidx, v.part_off, pprev)
      switch (estep_split) {
        case 8: RHCCQ_ESTEP_LAUNCH(8); break;
        case 4: RHCCQ_ESTEP_LAUNCH(4); break;
        case 2: RHCCQ_ESTEP_LAUNCH(2); break;
        default: RHCCQ_ESTEP_LAUNCH(1); break;
      }
#undef RHCCQ_ESTEP_LAUNCH
    }
    // arg-min over the centre tiles (tiled E-step) and every row's inertia term against the centres before the update
    hipLaunchKernelGGL(mbk_fold_tiles_kernel, dim3((1000 + 15) / 16, n_prob), dim3(64), 0, ctx->stream, v.probs, (const double*)state, step,
                       (const double*)centres, bk, v.pdist, v.pidx, v.part_off, v.pper[step & 1], use_grid ? 0 : 1);
    const UpdDraws dr{{v.bkeys[0], v.bkeys[1], v.bkeys[2], v.bkeys[3]}, step + 1, 1, 1, no_reassign ? 0 : -1, n_prob, ctx->opt_reassign_lds, v.resel,
                      ctx->opt_reassign_order == 1 ? v.qs_e : nullptr, v.qs_l, v.qs_r, v.qs_mask};
    hipLaunchKernelGGL(mbk_update_kernel, dim3(n_prob, 2), dim3(kUpdThreads), 0, ctx->stream, keys, v.probs, centres, weights, state, step,
                       words, (long long)n_words, bk, dr, (const int32_t*)v.pidx, v.part_off, (const int32_t*)nullptr);
    if (!no_reassign)
      hipLaunchKernelGGL(mbk_reassign_apply_kernel, dim3(n_prob, 2), dim3(kUpdThreads), 0, ctx->stream, keys, v.probs, centres, weights, state,
                         step, words, (long long)n_words, bk, dr);
  }
  if (n_steps > 0)
    hipLaunchKernelGGL(mbk_inertia_kernel, dim3(n_prob), dim3(64), 0, ctx->stream, v.probs, state, step0 + n_steps - 1,
                       (const double*)v.pper[(step0 + n_steps - 1) & 1]);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

// The same steps for ONE problem whose centres all carry weight (state: no zero-weight centre), with the E-step of step t + 1
// started beside the update of step t (k8_overlap.h).  since0 = the problem's "samples since the last reassignment" as step0 sees
// it (state[3] for an even step0, state[12] for an odd one): with it the host knows which steps of the call reassign.
// *carry (in/out, 0 before the first call and whenever a classic call came in between): bit 0 = batch step0 + 1 has been
// drawn, bit 1 = the speculative tile minima of step0 are in `work`.
int rhccq_mbk_steps_overlapped(rhccq_ctx* ctx, const uint32_t* keys, const rhccq_mbk_problem* probs, int32_t n_prob, int64_t step0,
                               int32_t n_steps, const uint32_t* words, int64_t n_words, double* centres, double* weights, double* state,
                               void* work, int64_t work_bytes, int32_t estep_split, int64_t since0, int32_t* carry) {
  if (!ctx || !keys || !probs || !words || !centres || !weights || !state || !work || !carry || n_steps < 0 || n_words <= 0 || step0 < 0 ||
      since0 < 0 || (*carry & ~3) != 0)
    return rhccq_fail(ctx, RHCCQ_E_ARG, "mbk_steps_overlapped: bad argument");
  if (n_prob != 1) return rhccq_fail(ctx, RHCCQ_E_ARG, "mbk_steps_overlapped: exactly one problem");
  if (estep_split != 0 && estep_split != 1 && estep_split != 2 && estep_split != 4 && estep_split != 8)
    return rhccq_fail(ctx, RHCCQ_E_ARG, "mbk_steps_overlapped: estep_split must be 0, 1, 2, 4 or 8");
  WorkView v;
  long long blocks;
  int max_tiles;
  if (step0 == 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "mbk_steps_overlapped: the first steps of a problem are rhccq_mbk_steps'");
  if (int e = layout_work(ctx, probs, n_prob, work, work_bytes, &v, &blocks, &max_tiles, false)) return e;
  const long long k = probs[0].k, bs = probs[0].n < 1000 ? probs[0].n : 1000;
  const int SS = estep_split == 0 ? 1 : estep_split;
  const int spec_tiles = (int)((k + kTileS - 1) / kTileS);
  // which steps of this call reassign (sklearn _random_reassign with no zero-weight centre left)
  std::vector<char> R((size_t)n_steps + 2, 0);
  {
    long long since = since0;
    for (int s = 0; s < n_steps + 2; ++s) {
      since += bs;
      R[s] = since >= 10 * k;
      if (R[s]) since = 0;
    }
  }
  long long drawn = step0 + ((*carry & 1) ? 1 : 0);        // newest batch in the ring
  bool have_spec = (*carry & 2) != 0;
  UpdDraws dr{{v.bkeys[0], v.bkeys[1], v.bkeys[2], v.bkeys[3]}, 0, 0, 0, 0, n_prob, ctx->opt_reassign_lds, v.resel,
              ctx->opt_reassign_order == 1 ? v.qs_e : nullptr, v.qs_l, v.qs_r, v.qs_mask};
  for (int s = 0; s < n_steps; ++s) {
    const long long step = step0 + s;
    const uint32_t* bk = v.bkeys[step & 3];
    if (!have_spec) {
#define RHCCQ_ESTEP_LAUNCH(S_)                                                                                                             \
  hipLaunchKernelGGL(mbk_batch_estep_kernel<S_>, dim3(max_tiles * kPtChunks * S_ + 1, n_prob), dim3(256), 0, ctx->stream, keys, v.probs, centres, \
                     state, step, bk, v.pdist, v.pidx, v.part_off, (const double*)nullptr)
      switch (SS) {
        case 8: RHCCQ_ESTEP_LAUNCH(8); break;
        case 4: RHCCQ_ESTEP_LAUNCH(4); break;
        case 2: RHCCQ_ESTEP_LAUNCH(2); break;
        default: RHCCQ_ESTEP_LAUNCH(1); break;
      }
#undef RHCCQ_ESTEP_LAUNCH
    }
    hipLaunchKernelGGL(mbk_fix_kernel, dim3((1000 + kFixPts - 1) / kFixPts, n_prob), dim3(256), 0, ctx->stream, v.probs, (const double*)state, step,
                       (const double*)centres, bk, (const double*)v.pdist, (const int32_t*)v.pidx, v.part_off, v.pper[step & 1],
                       have_spec ? (const int32_t*)v.lab[(step - 1) & 1] : (const int32_t*)nullptr, v.lab[step & 1],
                       have_spec ? kTileS : kTileC);
    dr.expect_reassign = R[s];
    dr.draw_first = drawn + 1;
    dr.draw_count = 0;
    dr.reassign_draws = 0;
    if (R[s]) {
      // a reassigning step: the classic update (it draws the batches behind its choice() itself), the inertia in a launch of its own
      if (drawn != step) return rhccq_fail(ctx, RHCCQ_E_LIMIT, "mbk_steps_overlapped: a batch was drawn past a reassignment (internal error)");
      dr.reassign_draws = R[s + 1] ? 1 : 2;
      drawn = step + dr.reassign_draws;
      hipLaunchKernelGGL(mbk_update_kernel, dim3(n_prob, 2), dim3(kUpdThreads), 0, ctx->stream, keys, v.probs, centres, weights, state, step,
                         words, (long long)n_words, bk, dr, (const int32_t*)v.pidx, v.part_off, (const int32_t*)v.lab[step & 1]);
      hipLaunchKernelGGL(mbk_reassign_apply_kernel, dim3(n_prob, 2), dim3(kUpdThreads), 0, ctx->stream, keys, v.probs, centres, weights, state,
                         step, words, (long long)n_words, bk, dr);
      hipLaunchKernelGGL(mbk_inertia_kernel, dim3(n_prob), dim3(64), 0, ctx->stream, v.probs, state, step, (const double*)v.pper[step & 1]);
      have_spec = false;
      continue;
    }
    // the speculative E-step of step + 1 needs its batch before this launch starts
    const bool spec_next = drawn >= step + 1;
    const long long target = R[s + 1] ? step + 1 : step + 2;
    if (target > drawn) { dr.draw_count = (int)(target - drawn); drawn = target; }
    hipLaunchKernelGGL(mbk_pipe_kernel, dim3(kPipeRoles + (spec_next ? spec_tiles * kSpecChunks : 0), n_prob), dim3(kPipeThreads), 0, ctx->stream,
                       keys, v.probs, centres, weights, state, step, words, (long long)n_words, bk, dr, (const int32_t*)v.lab[step & 1],
                       (const double*)v.pper[step & 1], (const uint32_t*)v.bkeys[(step + 1) & 3], v.pdist, v.pidx, v.part_off);
    have_spec = spec_next;
  }
  *carry = (drawn > step0 + n_steps ? 1 : 0) | (have_spec ? 2 : 0);
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
  hipLaunchKernelGGL(grid_build_kernel, dim3(n_prob), dim3(1024), 0, ctx->stream, v.probs, centres, v.cell_start, v.order,
                     (const double*)nullptr, 0ll);
  hipLaunchKernelGGL(mbk_assign_grid_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, keys, v.probs, centres, v.cell_start, v.order,
                     labels_out, v.blk_off);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

}  // extern "C"
