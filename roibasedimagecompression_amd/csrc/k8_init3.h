// k-means++ chain, third generation (included by k8_minibatch.hip; init samples <= 98 304: the 4K regime).
// Same arithmetic (exact integers) and the same picks as the first two generations; what changes is the pruning index
// and how a pick is laid out on the CU.  Why, measured on the second generation (MI355X, k = 38 392, 115 176 samples):
//   * SQ counters per pick: 6 800 VALU + 3 700 SALU + 720 LDS + 160 VMEM wave-instructions, waves parked 59 % of their
//     cycles (barriers, memory waits), issue stalls 16 %: a latency / synchronisation bound chain, so both the number of
//     instructions and the number of dependent round trips per phase count;
//   * a CPU simulation of the chain on the bench's 4K palettes (the same exact pruning rule, other block sizes): the
//     number of blocks a pick's candidates may improve hardly depends on the block size -- 223 / 261 / 277 / 244
//     (candidate, block) hits per pick for blocks of 64 / 16 / 8 / 4 samples at pick 5 000 -- because most hits are blocks
//     that really hold an improving sample (118 samples improve).  Blocks of 64 therefore evaluate 14 000 samples per
//     pick where blocks of 16 evaluate 4 200.
// Hence:
//   * LEAVES of 16 Morton-consecutive samples; a QUAD of four lanes holds a leaf (4 samples per lane), so one wave
//     instruction stream evaluates 16 (candidate, leaf) items;
//   * two box levels -- leaf (16 samples), super (16 leaves = 256 samples): the wave that found candidate t tests ALL
//     supers (<= 384: six independent rounds of 64 lanes, one LDS round trip for all of them), keeps the hits in a list of
//     its own and goes on to their leaves, four supers per round and four rounds per batch -- no shared hit list, no
//     barrier between the two levels, one atomic per batch.  (Measured on the way: a third level of 4 096-sample boxes
//     and a shared (candidate, super) list expanded behind a barrier took 9 000 cycles per pick for these two steps: every
//     dependent LDS round trip costs ~130 cycles and every dependent VALU instruction ~10 in this chain, so the number of
//     DEPENDENT steps is what counts, not the number of box tests);
//   * the candidate search walks a 64-ary tree of the draw-order sums (4 096-draw sums, 64-draw sums, samples): three
//     whole-wave scans (the second generation: chunked u64 scans over 1 024-draw sums, a row scan, a wave scan);
//   * a pick whose work list overflows (the first ~100 picks, when almost every leaf can still improve) is evaluated by
//     brute force over all samples, all waves side by side (no per-candidate enumeration).
// Round 4, second half (4.60 -> 3.5 us per pick on one box, tools/chain_ab.py, same picks): WHAT THE PICK IS BOUND BY.  Not the dependent
// round trips the first half of the round chased (merging five of them moved the pick by 1 %): the search phase is bound by the INSTRUCTIONS
// THE FOUR WAVES OF A SIMD ISSUE TOGETHER (three search waves + one idle wave) and by the bytes they pull through the LDS.  Evidence: the idle
// waves re-reading 256 leaf maxima per touched hyper -- bookkeeping for a hyper level that was switched off -- cost the search waves 9 % of the
// pick; a dozen fewer instructions per search wave are worth 1 %; letting the lanes without a leaf read a stale list entry's leaf instead
// of all reading entry 0 (fewer instructions, more LDS bytes) cost 2 %.  Hence: the supers' static boxes in registers, dense zero-padded
// maxima, one ds_read_b128 per leaf entry, the targets computed by the idle last wave, 32-bit scans and arg-max while the potential fits,
// straight-line code chosen by ONE wave-uniform branch (a branch per round between loads serialises them), and no wave-uniform test the
// idle waves have to pay for with more than a few instructions.
constexpr int kG3Threads = 1024;
constexpr int kG3Waves = kG3Threads / 64;
constexpr int kG3MaxLeaves = 6144;                      // leaf table in LDS: 96 KB = 98 304 init samples
constexpr int kG3MaxSamples = kG3MaxLeaves * 16;
constexpr int kG3MaxSup = kG3MaxLeaves / 16;            // 384
constexpr int kG3MaxHyp = kG3MaxSup / 16;               // 24 hypers of 16 supers = 4 096 samples
constexpr int kG3MaxDsum = kG3MaxSamples / 64;          // 64-draw sums: 1 536
constexpr int kG3MaxTop = kG3MaxDsum / 64;              // 4 096-draw sums: 24
static_assert(kG3MaxTop <= 32, "the 32-bit top-level scan covers half a wave");
constexpr int kG3MaxItems = 4096;                       // (candidate, leaf) items per pick
constexpr int kG3WList = kG3MaxSup;                     // hit supers one candidate keeps: room for all of them (u16 entries, 12 KB for 16 candidates)
constexpr int kG3Touch = 512;
// Super stage of the one-candidate-per-wave chain through a hyper level (24 boxes of 16 supers, tested in one round, then only the supers
// of the hit hypers): same picks (tests/test_gpu_kernels.py with this switched on), ~2 rounds of box tests instead of 6 -- and SLOWER where it
// matters: 4.50 vs 4.24 us per pick at k = 30 128, 4.2 vs 4.2 at k = 20 556, 3.10 vs 3.31 at k = 2 314 (round 4, MI355X, inside a frame).  The
// extra dependent LDS round trips (hyper table -> scalar bit walk -> super table) cost more than the ~80 box-test instructions they save:
// each search wave is a chain of dependent latencies, not an issue-bound stream.  Kept as a checked alternative, off.  (That explanation was
// the belief of the time; what the hyper level really cost was the bookkeeping it needs -- the idle waves re-reading 256 leaf maxima per
// touched hyper, 9 % of the pick, see above -- and its data-dependent loop.  kG3HypSkip below is what survived of the idea.)
constexpr bool kG3UseHypers = false;
// round 4: a super's BOX never changes after the tables are built (only its maximum does), so every lane keeps the boxes of "its" supers
// (lane + 64 r, r < 6) in 18 registers for the whole chain: the super stage then fetches only the six maxima (4 B instead of 16 B per
// super and wave: 55 KB less LDS traffic per pick) and computes the six box distances WHILE those are in flight, instead of waiting for
// six 16-byte entries first.  (The earlier attempt to hide this round trip -- the whole entries requested at the start of the pick -- lost
// because maxima read that early are stale; the boxes cannot be.)  One candidate per search wave only.
#ifndef RHCCQ_G3_REGBOX
#define RHCCQ_G3_REGBOX 1
#endif
constexpr bool kG3RegBox = RHCCQ_G3_REGBOX != 0;
// ... and a ROUND of 64 supers (= 4 hypers of 16) is skipped as a whole when none of its four hypers can improve: the <= 24 hyper entries are
// requested with the pick's first reads (a maximum read that early may be stale, i.e. larger: conservative), tested by 24 lanes in one go
// behind the search, and the six rounds' box tests run under wave-uniform guards.  Unlike the hyper LEVEL above this adds no dependent
// round trip: the supers' maxima are still requested unconditionally, only VALU work is skipped (a candidate meets 2-4 hypers).
#ifndef RHCCQ_G3_HYPSKIP
#define RHCCQ_G3_HYPSKIP 1
#endif
constexpr bool kG3HypSkip = RHCCQ_G3_HYPSKIP != 0;
#ifndef RHCCQ_G3_FIRSTQ
#define RHCCQ_G3_FIRSTQ 6                                             // rounds of 4 supers the first, straight-line leaf batch reaches (24 supers; 4 and 8 measured: section 8 of DESIGN.md)
#endif
#ifndef RHCCQ_G3_KEEP
#define RHCCQ_G3_KEEP 2
#endif
constexpr int kG3Keep = RHCCQ_G3_KEEP;                              // evaluation instruction streams (16 items each) a wave keeps in registers

struct G3Shared {
  unsigned long long delta[kTMaxI];
  unsigned long long R[kTMaxI];                         // integer search targets: ceil(u * pot)
  unsigned long long red64[kG3Waves];
  int cand[kTMaxI];
  uint2 ck[kTMaxI];                                     // candidate colour, its squared norm
  int n_items, overflow, n_touch2[2];
  int n_it[kTMaxI];                                     // in-wave variant: items of every candidate's own list
};

// number of set bits of a ballot below this lane: v_mbcnt_lo / _hi (2 instructions; popcount of the masked halves takes 4)
__device__ __forceinline__ int g3_rank_in(unsigned long long m) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// inclusive scan over lanes 0 .. 31 only (what lanes 32 .. 63 end up with is not a scan): the wave total of those lanes must fit 32 bits
__device__ __forceinline__ unsigned half_incscan_u32(unsigned v) {
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);   // row_shr:1
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);   // row_shr:2
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);   // row_shr:4
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);   // row_shr:8  -> scan inside each row of 16
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
  return v;
}
// (measured and dropped: the items counter's fetch-and-add as a hand-written ds_add_rtn_u32 instead of atomicAdd() under `if (lane == 0)` with the
// compiler's wave-aggregation around it -- 3.60 vs 3.57 us per pick: the inline wait and memory clobber cost more than the dozen instructions saved)
// a whole 16-byte table entry in ONE ds_read_b128: left alone the compiler fetches box and maximum separately (b96 + b32), and 64 lanes reading
// 4 bytes at a 16-byte stride keep the LDS as busy as the 12-byte part does
#ifndef RHCCQ_G3_B128
#define RHCCQ_G3_B128 1
#endif
__device__ __forceinline__ uint4 g3_lds_b128(const uint4* p) {
  if (!RHCCQ_G3_B128) return *p;
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  const v4u v = *(const volatile __attribute__((address_space(3))) v4u*)p;      // (an LDS pointer: a generic volatile access would become a flat load)
  return make_uint4(v.x, v.y, v.z, v.w);
}
// ... plus a base: v_mbcnt adds its last operand for free
__device__ __forceinline__ int g3_rank_add(unsigned long long m, int base) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, (unsigned)base));
}
__device__ __forceinline__ unsigned dpp_quad_sum(unsigned v) {
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
  return v;
}
__device__ __forceinline__ unsigned dpp_quad_max(unsigned v) {
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));
  return v;
}

// commit of one improved sample: both copies of the sample, the two levels of draw-order sums
__device__ __forceinline__ void g3_store1(int imp, int newcp, unsigned nx, uint32_t x, uint32_t y, int m, uint2* samp, uint2* dsamp, uint32_t* dsum,
                                          uint32_t* dtop) {
  if (imp > 0) {
    const uint32_t dpos = j_dpos(x, y);
    samp[m].y = ((uint32_t)newcp & 0x7ffffu) | (y & 0xfff80000u);
    dsamp[dpos].y = (uint32_t)(newcp + (int)nx);
    atomicSub(&dsum[dpos >> 6], (unsigned)imp);
    atomicSub(&dtop[dpos >> 12], (unsigned)imp);
  }
}

// one quad = one leaf (lane j of the quad holds samples 4j .. 4j+3 in a, bb): lower closest[] against the new centre,
// refresh the leaf's max, note its super as touched.  Straight-line code but for the stores.
// (Measured and dropped: the quad refreshing the super's and the hyper's maximum itself, right here, instead of the touch list the idle
// waves work off during the next search -- 3.79 vs 3.52 us per pick.  The leaves a winner improves are Morton neighbours, i.e. leaves
// of the SAME super held by different waves: each refresh misses the others' new maxima, writes a value that is too large, and the
// search pays for the stale maxima with leaf tests.)
__device__ __forceinline__ void g3_commit_quad(bool on, int b, uint32_t ck, int na, const uint4& a, const uint4& bb, uint2* samp, uint2* dsamp,
                                               uint4* blk, uint32_t* dsum, uint32_t* dtop, int* touch, int* n_touch) {
  const int j = threadIdx.x & 3;
  const int e0 = 2 * (int)__builtin_amdgcn_udot4(ck, a.x, 0u, false), e1 = 2 * (int)__builtin_amdgcn_udot4(ck, a.z, 0u, false);
  const int e2 = 2 * (int)__builtin_amdgcn_udot4(ck, bb.x, 0u, false), e3 = 2 * (int)__builtin_amdgcn_udot4(ck, bb.z, 0u, false);
  const int c0 = j_cprime(a.y), c1 = j_cprime(a.w), c2 = j_cprime(bb.y), c3 = j_cprime(bb.w);
  const int i0 = c0 - na + e0, i1 = c1 - na + e1, i2 = c2 - na + e2, i3 = c3 - na + e3;       // improvements (> 0: the sample moves)
  const unsigned n0 = norm2_key(a.x & 0xffffffu), n1 = norm2_key(a.z & 0xffffffu), n2 = norm2_key(bb.x & 0xffffffu), n3 = norm2_key(bb.z & 0xffffffu);
  const int w0 = i0 > 0 ? na - e0 : c0, w1 = i1 > 0 ? na - e1 : c1, w2 = i2 > 0 ? na - e2 : c2, w3 = i3 > 0 ? na - e3 : c3;   // new c' = d - |x|^2
  if (on && max(max(i0, i1), max(i2, i3)) > 0) {
    const int m0 = (b << 4) + 4 * j;
    g3_store1(i0, w0, n0, a.x, a.y, m0, samp, dsamp, dsum, dtop);
    g3_store1(i1, w1, n1, a.z, a.w, m0 + 1, samp, dsamp, dsum, dtop);
    g3_store1(i2, w2, n2, bb.x, bb.y, m0 + 2, samp, dsamp, dsum, dtop);
    g3_store1(i3, w3, n3, bb.z, bb.w, m0 + 3, samp, dsamp, dsum, dtop);
  }
  unsigned mx = max(max((unsigned)(w0 + (int)n0), (unsigned)(w1 + (int)n1)), max((unsigned)(w2 + (int)n2), (unsigned)(w3 + (int)n3)));
  mx = dpp_quad_max(on ? mx : 0u);
  if (on && j == 0) {
    blk[b].w = mx;
    const int slot = atomicAdd(n_touch, 1);
    if (slot < kG3Touch) touch[slot] = b >> 4;
  }
}

// kLdsS: the samples (both orders, 16 bytes each) live in the unused tail of the leaf table instead of global memory -- init samples
// up to kG3LdsSamples (every problem with k <= 1 900: the level-2 palettes of a frame, the segments of a many-segment frame).  A pick
// then makes no L2 round trip at all: 3.5 us alone on the chip either way, but INSIDE a frame, beside another problem's step
// kernels, the global-memory version of a level-2 chain ran at 6.9 us per pick (its two dependent L2 reads waited behind their traffic).
constexpr int kG3LdsSamples = 5760;   // 17 / 16 x padded samples <= kG3MaxLeaves table entries
// kIW (round 4, kCW == 1 only): EVALUATION INSIDE THE CANDIDATE'S OWN WAVE.  The wave that found candidate t and its leaves keeps them in a list of
// its own (no shared list, no atomic on a shared counter), loads their samples and sums its candidate's improvement itself -- no barrier between
// search and evaluation, no atomics on the improvement sums: two barriers per pick instead of three.  The winner's wave commits from its registers.
// Same picks (tests) and SLOWER: 4.82 vs 4.60 us per pick at k = 30 128 on one box (tools/chain_ab.py opt:3 opt:5): the pick now waits for the
// candidate with the most leaves to search AND evaluate them alone, where the shared list spreads every candidate's leaves over all 16 waves; the
// barrier it saves is cheaper than that imbalance.  Opt-in (RHCCQ_OPT_INIT_KERNEL = 5), kept as a checked alternative.
template <int kCW, bool kLdsS = false, bool kIW = false>          // candidates per search wave: 1 (one wave each), 2 or 3
__global__ __launch_bounds__(kG3Threads) void mbk_init3_kernel(const uint32_t* __restrict__ keys, const MbkP* __restrict__ probs,
                                                               const int32_t* __restrict__ init_idx, const int32_t* __restrict__ perm,
                                                               const double* __restrict__ rand, double* __restrict__ centres,
                                                               int32_t* __restrict__ chosen, uint32_t* scratch,
                                                               const long long* __restrict__ scratch_off, int max_items) {
  __shared__ G3Shared sh;
  __shared__ uint4 blk[kG3MaxLeaves];                    // per leaf: box (3 pairs), max closest
  __shared__ uint4 sup[kG3MaxSup];                       // per super: box, max of the leaves' max (may lag high)
  __shared__ uint32_t supw[kG3MaxSup];                   // the maxima again, DENSE: 64 lanes reading sup[].w walk LDS at a 16-byte stride (bank conflicts)
  __shared__ uint4 hyp[kG3MaxHyp];                       // per hyper (16 supers): box, max (may lag high) -- round 4: see the super stage
  __shared__ uint32_t dsum[kG3MaxDsum];                  // per 64 consecutive draws: sum of closest
  __shared__ uint32_t dtop[kG3MaxTop];                   // per 4 096 consecutive draws
  __shared__ uint32_t items[kG3MaxItems];
  __shared__ uint16_t wlist[kTMaxI * kG3WList];          // per candidate: the supers it may improve
  __shared__ int s_touch[2 * kG3Touch];
  const MbkP P = probs[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform BY CONSTRUCTION: loops and branches on it stay scalar
  const int n = (int)P.init_n, k = (int)P.k, T = P.T;
  const int NW = (T + kCW - 1) / kCW;                             // search waves
  const int nd = (n + 63) >> 6, np = nd << 6;                     // 64-draw blocks; padded sample count
  const int nb = np >> 4, nsb = (nb + 15) >> 4, ntop = (nd + 63) >> 6, nhyp = (nsb + 15) >> 4;
  uint2* samp = kLdsS ? reinterpret_cast<uint2*>(&blk[nb]) : reinterpret_cast<uint2*>(scratch + scratch_off[blockIdx.x]);
  uint2* dsamp = samp + np;
  int32_t* cho = chosen + P.koff;
  const int rq = lane >> 4, rj = lane & 15;
  const int quad = lane >> 2, qj = lane & 3;
  static_assert(!kIW || kCW == 1, "in-wave evaluation: one candidate per search wave");
  const int cap_c = kIW ? min(kG3MaxItems / kTMaxI, max_items / kTMaxI) : 0;      // in-wave variant: room of one candidate's own list
  // ---- gather the sample (both orders), first centre, tables ---------------------------------------------
  const uint32_t kf = keys[P.off + init_idx[P.init_off + P.first]];
  {
    const int last_d = perm[P.init_off + n - 1];
    for (int i = tid; i < np; i += kG3Threads) {
      const int d = i < n ? perm[P.init_off + i] : i;     // padding: the last Morton sample again, closest = 0, unused draw slots
      const uint32_t kk = keys[P.off + init_idx[P.init_off + (i < n ? d : last_d)]];
      const unsigned cl = i < n ? (unsigned)dist2_keys(kk, kf) : 0u;
      samp[i] = jpack(kk, (int)cl - (int)norm2_key(kk), (uint32_t)d);
      dsamp[d] = make_uint2(kk, cl);
    }
  }
  __syncthreads();
  for (int b0 = wave * 4; b0 < nb; b0 += kG3Waves * 4) {   // one row of 16 lanes per leaf (nb is a multiple of 4)
    const int b = b0 + rq;
    const uint2 sv = samp[(b << 4) + rj];
    const uint32_t kk = sv.x & 0xffffffu;
    const unsigned r1 = dpp_row_max(key_r(kk)), r0 = 255u - dpp_row_max(255u - key_r(kk));
    const unsigned g1 = dpp_row_max(key_g(kk)), g0 = 255u - dpp_row_max(255u - key_g(kk));
    const unsigned b1 = dpp_row_max(key_b(kk)), bl0 = 255u - dpp_row_max(255u - key_b(kk));
    const unsigned dm = dpp_row_max((unsigned)(j_cprime(sv.y) + (int)norm2_key(kk)));
    if (rj == 0) blk[b] = make_uint4(box_pair(r0, r1), box_pair(g0, g1), box_pair(bl0, b1), dm);
  }
  for (int d = wave; d < nd; d += kG3Waves) {
    const unsigned ds = wave_sum_u32(dsamp[(d << 6) + lane].y);
    if (lane == 0) dsum[d] = ds;
  }
  for (int d = nd + tid; d < kG3MaxDsum; d += kG3Threads) dsum[d] = 0u;        // (zero sums beyond the last block: the search reads whole rows of 64 unclamped)
  for (int i = tid; i < kTMaxI * kG3WList / 2; i += kG3Threads) reinterpret_cast<uint32_t*>(wlist)[i] = 0u;   // (a stale list entry always names a super)
  __syncthreads();
  for (int sb = tid; sb < nsb; sb += kG3Threads) {
    int r0 = 255, g0 = 255, b0 = 255, r1 = 0, g1 = 0, b1 = 0;
    unsigned m = 0;
    for (int b = sb * 16; b < min(sb * 16 + 16, nb); ++b) {
      const uint4 be = blk[b];
      r0 = min(r0, pair_lo(be.x)); g0 = min(g0, pair_lo(be.y)); b0 = min(b0, pair_lo(be.z));
      r1 = max(r1, pair_hi(be.x)); g1 = max(g1, pair_hi(be.y)); b1 = max(b1, pair_hi(be.z));
      m = max(m, be.w);
    }
    sup[sb] = make_uint4(box_pair((unsigned)r0, (unsigned)r1), box_pair((unsigned)g0, (unsigned)g1), box_pair((unsigned)b0, (unsigned)b1), m);
    supw[sb] = m;
  }
  for (int sb = nsb + tid; sb < kG3MaxSup; sb += kG3Threads) supw[sb] = 0u;      // (no super there: no distance is below 0, the rounds need no lane mask)
  unsigned long long psum = 0;
  for (int tt = tid; tt < ntop; tt += kG3Threads) {
    unsigned sum = 0;                                     // 4 096 x 195 075 fits 32 bits
    for (int d = tt * 64; d < min(tt * 64 + 64, nd); ++d) sum += dsum[d];
    dtop[tt] = sum;
    psum += sum;
  }
  psum = block_sum<unsigned long long>(psum, sh.red64);
  for (int h = tid; h < nhyp; h += kG3Threads) {          // (sup is visible: the sum above went through two barriers)
    int r0 = 255, g0 = 255, b0 = 255, r1 = 0, g1 = 0, b1 = 0;
    unsigned m = 0;
    for (int sb = h * 16; sb < min(h * 16 + 16, nsb); ++sb) {
      const uint4 se = sup[sb];
      r0 = min(r0, pair_lo(se.x)); g0 = min(g0, pair_lo(se.y)); b0 = min(b0, pair_lo(se.z));
      r1 = max(r1, pair_hi(se.x)); g1 = max(g1, pair_hi(se.y)); b1 = max(b1, pair_hi(se.z));
      m = max(m, se.w);
    }
    hyp[h] = make_uint4(box_pair((unsigned)r0, (unsigned)r1), box_pair((unsigned)g0, (unsigned)g1), box_pair((unsigned)b0, (unsigned)b1), m);
  }
  if (tid == 0) { cho[0] = P.first; sh.n_touch2[0] = 0; sh.n_touch2[1] = 0; sh.n_items = 0; sh.overflow = 0; }
  if (tid < kTMaxI) sh.delta[tid] = 0;
  // Every thread tracks the potential in a register (the block sums hand all of them the same totals).  The LAST wave -- never a search
  // wave: T <= 12 for init samples <= 98 304 -- fetches the uniforms of the coming pick (a cold line in HBM: a wave's memory operations
  // return in order, so a search wave that asked for it would wait for it before its own sample read) and turns them into the integer
  // targets at the end of the pick; it also records the winner.
  unsigned long long pot = psum;
  const bool r_wave = wave == kG3Waves - 1;
  if (r_wave && lane < T && k > 1) sh.R[lane] = (unsigned long long)ceil(rand[P.rand_off + lane] * (double)psum);
  __syncthreads();
  uint32_t sbx[kG3MaxSup / 64], sby[kG3MaxSup / 64], sbz[kG3MaxSup / 64];      // (kG3RegBox) the static boxes of supers lane + 64 r
#pragma unroll
  for (int r = 0; r < kG3MaxSup / 64; ++r) {
    const uint4 e = sup[min(r * 64 + lane, nsb - 1)];
    sbx[r] = e.x; sby[r] = e.y; sbz[r] = e.z;
  }
#ifdef RHCCQ_STAMPS
  unsigned long long _acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long _last = clock64();
  unsigned long long _wacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, _wt = 0;
#define WSPLIT(ph) do { const unsigned long long _t2 = clock64(); _wacc[ph] += _t2 - _wt; _wt = _t2; } while (0)
#else
#define WSPLIT(ph) do {} while (0)
#endif
  for (int c = 1; c < k; ++c) {
    WBEGIN();
    // the next pick's uniforms are a cold line in HBM: fetch them now, use them at the end of the pick
    double u_next = 0.0;
    if (r_wave && lane < T && c + 1 < k) u_next = rand[P.rand_off + (size_t)c * T + lane];
    if (!kIW && tid < T) sh.delta[tid] = 0;              // (read for the arg-max before the previous pick's closing barrier; the in-wave variant
                                                         //  stores every candidate's sum instead of adding to it)
    const int* touch_r = s_touch + (((c - 1) & 1) ? kG3Touch : 0);
    const int n_touched = min(sh.n_touch2[(c - 1) & 1], kG3Touch);
    uint4 ka[kG3Keep], kb[kG3Keep];
    uint32_t kw[kG3Keep];
    int n_it = 0;                                        // (in-wave variant: items in this wave's own list)
    // ================= phase 1: waves t < T -- candidate t, the supers and the leaves it may improve ================
    if (wave < NW) {
      // Search wave w finds candidates w, w + NW, ... (kCW of them, the last wave possibly fewer): kCW independent dependency chains
      // in ONE instruction stream.  A chain of this phase is latency bound (~11 cycles per instruction with three such waves per
      // SIMD, the youngest of them starved by the issue arbiter until the older ones stall); interleaved in one wave the chains fill
      // each other's bubbles, and whatever does not depend on the candidate (the top-level scan, every super's box) is loaded once.
      // np.searchsorted(cumsum(closest), r, 'left') in DRAW order; cum and the target R = ceil(r) are exact integers
      int tc[kCW];
      bool on[kCW];
      unsigned long long R[kCW], rv[kCW];
      const unsigned v_top = dtop[min(lane, ntop - 1)];       // (the targets and the top level of the sums in ONE round trip)
      uint4 hyp_e = make_uint4(0, 0, 0, 0);
      if constexpr (kCW == 1 && kG3RegBox && kG3HypSkip) hyp_e = g3_lds_b128(&hyp[min(lane, nhyp - 1)]);
#pragma unroll
      for (int c2 = 0; c2 < kCW; ++c2) {
        tc[c2] = wave + c2 * NW;
        on[c2] = tc[c2] < T;
        rv[c2] = sh.R[min(tc[c2], T - 1)];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c2 = 0; c2 < kCW; ++c2)
        R[c2] = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(rv[c2] >> 32)) << 32) |
                (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)rv[c2]);
      int cand[kCW], l1[kCW], bh[kCW];
      uint32_t ck[kCW];
      bool found[kCW];
      unsigned rr[kCW], rr3[kCW];
      if (__builtin_amdgcn_readfirstlane((int)(pot >> 32)) == 0) {
        // the whole potential fits 32 bits (every pick of a photo's chain: 90 000 samples x ~10^4): ONE scan of the <= 24 top-level
        // sums over half a wave (5 shifted adds) instead of two 16-bit-limb scans over the whole wave and 64-bit compares
        const unsigned v = lane < ntop ? v_top : 0u;
        const unsigned inc = half_incscan_u32(v);
        const unsigned exc = inc - v;
#pragma unroll
        for (int c2 = 0; c2 < kCW; ++c2) {
          const unsigned r32 = (unsigned)R[c2];
          const unsigned long long m1 = __ballot(v > 0 && exc < r32 && r32 <= inc);
          found[c2] = m1 != 0ull;
          l1[c2] = found[c2] ? __ffsll((long long)m1) - 1 : 0;
          rr[c2] = r32 - (unsigned)__builtin_amdgcn_readlane((int)exc, l1[c2]);             // <= the 4 096-draw sum
        }
      } else {
        const unsigned v = lane < ntop ? v_top : 0u;
        const unsigned long long inc = wave_incscan_limbs(v);
        const unsigned long long exc = inc - v;
#pragma unroll
        for (int c2 = 0; c2 < kCW; ++c2) {
          const unsigned long long m1 = __ballot(v > 0 && exc < R[c2] && R[c2] <= inc);
          found[c2] = m1 != 0ull;
          l1[c2] = found[c2] ? __ffsll((long long)m1) - 1 : 0;
          rr[c2] = (unsigned)(R[c2] - readlane64(exc, l1[c2]));             // <= the 4 096-draw sum
        }
      }
      {
        unsigned v2[kCW];
#pragma unroll
        for (int c2 = 0; c2 < kCW; ++c2) v2[c2] = dsum[l1[c2] * 64 + lane];      // (l1 < 24: inside the zero-padded table)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c2 = 0; c2 < kCW; ++c2) {
          const unsigned w2 = v2[c2];
          const unsigned inc2 = wave_incscan_u32(w2);
          const unsigned long long m2 = __ballot(w2 > 0 && (inc2 - w2) < rr[c2] && rr[c2] <= inc2);
          found[c2] = found[c2] && m2 != 0ull;
          const int l2 = m2 ? __ffsll((long long)m2) - 1 : 0;
          bh[c2] = l1[c2] * 64 + l2;
          rr3[c2] = rr[c2] - (unsigned)__builtin_amdgcn_readlane((int)(inc2 - w2), l2);   // <= the block's sum
        }
      }
      {
        uint2 sv[kCW];
#pragma unroll
        for (int c2 = 0; c2 < kCW; ++c2) sv[c2] = dsamp[(bh[c2] << 6) + lane];      // (bh names a real block; the slots behind the last sample hold closest = 0)
        // (measured, no gain: requesting the supers' maxima here, so that they arrive during this L2 round trip -- 3.53 vs 3.54 us per pick)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c2 = 0; c2 < kCW; ++c2) {
          const int i = (bh[c2] << 6) + lane;
          const unsigned inc3 = wave_incscan_u32(sv[c2].y);                       // 64 x 195075 fits 32 bits
          const unsigned long long m3 = __ballot(inc3 >= rr3[c2]);               // (a padding slot adds nothing: never the first lane to reach the target)
          const int l3 = m3 ? __ffsll((long long)m3) - 1 : min(63, n - 1 - (bh[c2] << 6));
          // R = 0: position 0; a target beyond the total (cannot happen): the last sample
          cand[c2] = found[c2] ? (bh[c2] << 6) + l3 : (R[c2] == 0 ? 0 : n - 1);
          ck[c2] = (uint32_t)__builtin_amdgcn_readlane((int)sv[c2].x, l3);
        }
      }
#pragma unroll
      for (int c2 = 0; c2 < kCW; ++c2)
        if (!found[c2]) ck[c2] = dsamp[cand[c2]].x;          // (wave-uniform, rare)
      WSPLIT(4);
      CandP cp[kCW];
#pragma unroll
      for (int c2 = 0; c2 < kCW; ++c2) {
        cp[c2] = cand_pairs(ck[c2]);
        if (lane == 0 && on[c2]) {
          sh.cand[tc[c2]] = cand[c2];
          sh.ck[tc[c2]] = make_uint2(ck[c2], norm2_key(ck[c2]));
        }
      }
      // (round 4, measured and dropped: requesting the super entries at the START of the pick, so that they arrive in the shadow of the search's
      // three dependent round trips -- same picks, 4.75 vs 4.62 us per pick on one box, tools/chain_ab.py: read that early the maxima of the supers
      // the previous winner touched have not been refreshed by the idle waves yet, and a stale larger maximum means more leaves to test.)
      // every super, 64 per round, against the wave's candidates; the rounds are independent of each other (one LDS round trip for
      // all: loads unconditional, indices clamped, all of them issued before the first test -- sched_barrier: left alone the
      // compiler waits for each entry before it asks for the next).  The maxima may be mid-refresh by the idle waves: a stale,
      // larger maximum is conservative.  Hits go to the candidate's own list.
      int n_sup[kCW];
#pragma unroll
      for (int c2 = 0; c2 < kCW; ++c2) n_sup[c2] = 0;
      if constexpr (kCW == 1 && kG3UseHypers) {
        // round 4: the <= 24 hypers (16 supers each) in ONE round, then only the supers of the hypers that may improve, four hypers per
        // round (a row of 16 lanes each).  A candidate meets 2-4 hypers of the 22 of a 90 000-sample problem: ~2 rounds of box tests
        // instead of 6 (the chain is bound by the instructions its 12 search waves issue on 4 SIMDs), for one more dependent LDS
        // round trip.  Hit supers land in the list in ascending order, as before.
        const uint4 he = hyp[min(lane, nhyp - 1)];
        const bool hh = (bool)((int)(box_dist2(cp[0], he.x, he.y, he.z) < he.w) & (int)(lane < nhyp) & (int)on[0]);
        unsigned long long mh = __ballot(hh);
        while (mh) {
          int h4[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            h4[i] = mh ? __ffsll((long long)mh) - 1 : -1;
            mh &= mh - 1ull;                               // (0 stays 0)
          }
          const int my_h = sel4(rq, h4);
          const int sbi = my_h * 16 + rj;
          const bool valid = (bool)((int)(my_h >= 0) & (int)(sbi < nsb));
          const uint4 se = sup[valid ? sbi : 0];
          const bool h = (bool)((int)valid & (int)(box_dist2(cp[0], se.x, se.y, se.z) < se.w));
          const unsigned long long m = __ballot(h);
          if (h) wlist[tc[0] * kG3WList + n_sup[0] + g3_rank_in(m)] = (uint16_t)sbi;
          n_sup[0] += __popcll(m);
        }
      } else if constexpr (kCW == 1 && kG3RegBox) {
        // two straight-line versions (all six rounds / the two a small problem has) behind ONE wave-uniform branch: a branch per round
        // between the loads serialises them
        unsigned hyp_m = 0xffffffffu;
        if constexpr (kG3HypSkip)
          hyp_m = (unsigned)__ballot((bool)((int)(box_dist2(cp[0], hyp_e.x, hyp_e.y, hyp_e.z) < hyp_e.w) & (int)(lane < nhyp)));
        auto rounds = [&](auto nr_c) {
          constexpr int kNr = decltype(nr_c)::value;
          uint32_t sw[kNr];
          unsigned d2[kNr];
#pragma unroll
          for (int r = 0; r < kNr; ++r) sw[r] = supw[r * 64 + lane];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int r = 0; r < kNr; ++r) d2[r] = box_dist2(cp[0], sbx[r], sby[r], sbz[r]);      // in the shadow of the loads
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int r = 0; r < kNr; ++r) {
            if (kG3HypSkip && ((hyp_m >> (4 * r)) & 15u) == 0u) continue;      // (wave-uniform)
            const bool h = d2[r] < sw[r];                   // (kCW == 1: the wave's candidate exists; lanes beyond the last super read a zero maximum)
            const unsigned long long m = __ballot(h);
            if (h) wlist[g3_rank_add(m, tc[0] * kG3WList + n_sup[0])] = (uint16_t)(r * 64 + lane);
            n_sup[0] += __popcll(m);
          }
        };
        if (nsb > 128) rounds(std::integral_constant<int, kG3MaxSup / 64>{});
        else rounds(std::integral_constant<int, 2>{});
      } else {
        uint4 se[kG3MaxSup / 64];
        const int nr = nsb > 128 ? kG3MaxSup / 64 : 2;      // (wave-uniform; the unrolled rounds beyond it cost a scalar branch)
#pragma unroll
        for (int r = 0; r < kG3MaxSup / 64; ++r) {
          se[r] = make_uint4(0, 0, 0, 0);
          if (r < nr) se[r] = sup[min(r * 64 + lane, nsb - 1)];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < kG3MaxSup / 64; ++r) {
#pragma unroll
          for (int c2 = 0; c2 < kCW; ++c2) {
            const unsigned d2 = box_dist2(cp[c2], se[r].x, se[r].y, se[r].z);
            const bool h = (bool)((int)(d2 < se[r].w) & (int)(r * 64 + lane < nsb) & (int)on[c2]);      // (no short circuit)
            const unsigned long long m = __ballot(h);
            if (h) wlist[tc[c2] * kG3WList + n_sup[c2] + g3_rank_in(m)] = (uint16_t)(r * 64 + lane);
            n_sup[c2] += __popcll(m);
          }
        }
      }
      WSPLIT(5);
#ifdef RHCCQ_STAMPS
      if (wave == 0) { _acc[13] += (unsigned long long)n_sup[0]; _acc[14] += (unsigned long long)((n_sup[0] + 15) >> 4); }
#endif
      // the leaves of the hit supers: four supers per round (one row of lanes each), four independent rounds per batch; the FIRST
      // batch of every candidate of the wave in one straight line (one atomic for all of them), further batches (one candidate in
      // six has more than 16 hit supers) candidate by candidate
      // (straight-line versions for 1 .. 4 rounds behind ONE wave-uniform branch -- most candidates have 5 .. 16 hit supers; a branch per
      //  round between the loads serialises them: measured, 4.65 vs 4.39 us per pick)
      // (a candidate with more than 16 hit supers -- one in six, i.e. most picks have one -- used to take a second, dependent batch: list read,
      //  leaf read, tests, atomic; the pick waits for its slowest search wave, so the first straight line now reaches RHCCQ_G3_FIRSTQ rounds)
      constexpr int kFirstQ = kCW == 1 ? RHCCQ_G3_FIRSTQ : 4;
      auto first_batch = [&](auto nq_c) {
        constexpr int kNq = decltype(nq_c)::value;
        int bq[kCW][kNq];
        bool hb[kCW][kNq];
        unsigned long long mb[kCW][kNq];
        uint32_t sq[kCW][kNq];
        uint4 be[kCW][kNq];
#pragma unroll
        for (int c2 = 0; c2 < kCW; ++c2)
#pragma unroll
          for (int q = 0; q < kNq; ++q) sq[c2][q] = wlist[min(tc[c2], kTMaxI - 1) * kG3WList + 4 * q + rq];      // (entries beyond n_sup: stale, masked below)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c2 = 0; c2 < kCW; ++c2)
#pragma unroll
          for (int q = 0; q < kNq; ++q) {
            const int b = (int)sq[c2][q] * 16 + rj;
            bq[c2][q] = ((int)(4 * q + rq < n_sup[c2]) & (int)(b < nb)) ? b : -1;      // (the lanes without a leaf all read entry 0: one broadcast.
            be[c2][q] = g3_lds_b128(&blk[max(bq[c2][q], 0)]);                          //  Measured: letting them read whatever a stale list entry names costs 2 %)
          }
        __builtin_amdgcn_sched_barrier(0);
        int total = 0;
#pragma unroll
        for (int c2 = 0; c2 < kCW; ++c2)
#pragma unroll
          for (int q = 0; q < kNq; ++q) {
            const unsigned d2 = box_dist2(cp[c2], be[c2][q].x, be[c2][q].y, be[c2][q].z);
            hb[c2][q] = (bool)((int)(d2 < be[c2][q].w) & (int)(bq[c2][q] >= 0));
            mb[c2][q] = __ballot(hb[c2][q]);
            total += __popcll(mb[c2][q]);
          }
        if (total) {
          int base = 0;
          if constexpr (kIW) {
            // the wave's own list: no shared counter
            if (n_it + total > cap_c) { if (lane == 0) sh.overflow = 1; n_it = cap_c + 1; }
            else {
              base = tc[0] * cap_c + n_it;
#pragma unroll
              for (int q = 0; q < kNq; ++q) {
                if (hb[0][q]) items[base + g3_rank_in(mb[0][q])] = ((uint32_t)tc[0] << 24) | (uint32_t)bq[0][q];
                base += __popcll(mb[0][q]);
              }
              n_it += total;
            }
          } else {
          if (lane == 0) base = atomicAdd(&sh.n_items, total);
          base = __builtin_amdgcn_readfirstlane(base);
          if (base + total > max_items) { if (lane == 0) sh.overflow = 1; }
          else {
#pragma unroll
            for (int c2 = 0; c2 < kCW; ++c2)
#pragma unroll
              for (int q = 0; q < kNq; ++q) {
                if (hb[c2][q]) items[g3_rank_add(mb[c2][q], base)] = ((uint32_t)tc[c2] << 24) | (uint32_t)bq[c2][q];
                base += __popcll(mb[c2][q]);
              }
          }
          }
        }
      };
      {
        int ns_max = n_sup[0];
#pragma unroll
        for (int c2 = 1; c2 < kCW; ++c2) ns_max = max(ns_max, n_sup[c2]);
        if constexpr (kFirstQ >= 8) {
          if (ns_max > 28) { first_batch(std::integral_constant<int, 8>{}); ns_max = 0; }
          else if (ns_max > 24) { first_batch(std::integral_constant<int, 7>{}); ns_max = 0; }
        }
        if constexpr (kFirstQ >= 6) {
          if (ns_max > 20) { first_batch(std::integral_constant<int, 6>{}); ns_max = 0; }
          else if (ns_max > 16) { first_batch(std::integral_constant<int, 5>{}); ns_max = 0; }
        }
        if (ns_max > 12) first_batch(std::integral_constant<int, 4>{});
        else if (ns_max > 8) first_batch(std::integral_constant<int, 3>{});
        else if (ns_max > 4) first_batch(std::integral_constant<int, 2>{});
        else if (ns_max > 0) first_batch(std::integral_constant<int, 1>{});
      }
#pragma unroll
      for (int c2 = 0; c2 < kCW; ++c2) {
        const uint16_t* wl = wlist + min(tc[c2], kTMaxI - 1) * kG3WList;
        for (int h0 = 4 * kFirstQ; h0 < n_sup[c2]; h0 += 16) {
          int bq[4];
          bool hb[4];
          unsigned long long mb[4];
          uint32_t sq[4];
          uint4 be[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) sq[q] = wl[min(h0 + 4 * q + rq, kG3WList - 1)];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int b = (int)sq[q] * 16 + rj;
            bq[q] = ((int)(h0 + 4 * q + rq < n_sup[c2]) & (int)(b < nb)) ? b : -1;
            be[q] = g3_lds_b128(&blk[max(bq[q], 0)]);
          }
          __builtin_amdgcn_sched_barrier(0);
          int total = 0;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const unsigned d2 = box_dist2(cp[c2], be[q].x, be[q].y, be[q].z);
            hb[q] = (bool)((int)(d2 < be[q].w) & (int)(bq[q] >= 0));
            mb[q] = __ballot(hb[q]);
            total += __popcll(mb[q]);
          }
          if (total) {
            int base = 0;
            if constexpr (kIW) {
              if (n_it + total > cap_c) { if (lane == 0) sh.overflow = 1; n_it = cap_c + 1; }
              else {
                base = tc[c2] * cap_c + n_it;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  if (hb[q]) items[base + g3_rank_in(mb[q])] = ((uint32_t)tc[c2] << 24) | (uint32_t)bq[q];
                  base += __popcll(mb[q]);
                }
                n_it += total;
              }
            } else {
            if (lane == 0) base = atomicAdd(&sh.n_items, total);
            base = __builtin_amdgcn_readfirstlane(base);
            if (base + total > max_items) { if (lane == 0) sh.overflow = 1; }
            else {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                if (hb[q]) items[base + g3_rank_in(mb[q])] = ((uint32_t)tc[c2] << 24) | (uint32_t)bq[q];
                base += __popcll(mb[q]);
              }
            }
            }
          }
        }
      }
      if constexpr (kIW) {
        // ================= the candidate's improvement, summed by its own wave: 16 leaves per instruction stream (a quad of lanes holds a leaf),
        // the first kG3Keep streams stay in registers for the commit.  The list was written by this wave: LDS keeps a wave's accesses in order.
        const bool ok = n_it <= cap_c;
        unsigned long long acc = 0;
        if (ok && on[0]) {
          const uint32_t* mine = items + tc[0] * cap_c;
#pragma unroll
          for (int s2 = 0; s2 < kG3Keep; ++s2) {
            kw[s2] = 0xffffffffu;
            ka[s2] = make_uint4(0, 0, 0, 0);
            kb[s2] = ka[s2];
            if (16 * s2 < n_it) {
              const int ii = 16 * s2 + quad;
              const uint32_t w = mine[min(ii, n_it - 1)];
              const uint4* p4 = reinterpret_cast<const uint4*>(samp + ((w & 0xffffffu) << 4) + 4 * qj);
              ka[s2] = p4[0];
              kb[s2] = p4[1];
              kw[s2] = ii < n_it ? w : (w | 0xf0000000u);  // the high nibble marks a padding quad
            }
          }
          const uint32_t ckx = ck[0];
          const int na = (int)norm2_key(ckx);
#pragma unroll
          for (int s2 = 0; s2 < kG3Keep; ++s2)
            if (16 * s2 < n_it) acc += (kw[s2] >> 28) ? 0u : j_eval4(ckx, na, ka[s2], kb[s2]);
          for (int o = kG3Keep; 16 * o < n_it; ++o) {        // more leaves than the registers hold
            const int ii = 16 * o + quad;
            if (ii < n_it) {
              const uint32_t w = mine[ii];
              const uint4* p4 = reinterpret_cast<const uint4*>(samp + ((w & 0xffffffu) << 4) + 4 * qj);
              acc += j_eval4(ckx, na, p4[0], p4[1]);
            }
          }
        }
        acc = wave_sum(acc);
        if (lane == 0 && on[0]) { sh.delta[tc[0]] = acc; sh.n_it[tc[0]] = ok ? n_it : 0; }
      }
    } else {
      // the other waves refresh the super maxima the previous winner touched
      for (int i = tid - NW * 64; i < n_touched * 16; i += kG3Threads - NW * 64) {
        const int sb = touch_r[i >> 4], b = sb * 16 + (i & 15);
        unsigned m = b < nb ? blk[b].w : 0u;
        m = dpp_row_max(m);
        if ((i & 15) == 0) { sup[sb].w = m; supw[sb] = m; }
      }
      // ... and the maxima of their hypers, straight from the leaves (256 per hyper, four per lane): independent of the super writes
      // above, and like them allowed to lag high
      if constexpr (kG3RegBox && kG3HypSkip && !kG3UseHypers) {
        // the round guards only need the hyper maxima roughly: from the (dense) super maxima, a row of 16 lanes per touched super.  A super
        // another lane is refreshing right now may still show its old, larger value: conservative, and put right the next time the hyper
        // is touched.  (Re-reading the hyper's 256 leaves, as the hyper LEVEL wants it, cost the search waves 9 % of the pick: the
        // phase is bound by the instructions the four waves of a SIMD issue together.)
        for (int i = tid - NW * 64; i < n_touched * 16; i += kG3Threads - NW * 64) {
          const int h = touch_r[i >> 4] >> 4;
          unsigned m = supw[h * 16 + (i & 15)];
          m = dpp_row_max(m);
          if ((i & 15) == 0) hyp[h].w = m;
        }
      }
      if constexpr (kG3UseHypers)                                   // (exact hyper maxima only while the hyper level reads them)
      for (int e = wave - NW; e < n_touched; e += kG3Waves - NW) {
        const int h = touch_r[e] >> 4;
        unsigned m = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int b = h * 256 + j * 64 + lane;
          m = max(m, b < nb ? blk[b].w : 0u);
        }
        m = wave_max_u32(m);
        if (lane == 0) hyp[h].w = m;
      }
    }
    WEND(0);
    STAMP(0);
    __syncthreads();
    STAMP(1);
    WBEGIN();
    // ================= phase 3: potentials ===================================================================
    // (this wave's slots of the work list are requested TOGETHER with the list's length -- one LDS round trip instead of two; what a slot
    //  beyond the length holds is never used)
    uint32_t wpre[kG3Keep];
#pragma unroll
    for (int s = 0; s < kG3Keep; ++s) wpre[s] = kIW ? 0u : items[16 * (wave + s * kG3Waves) + quad];
    const int ovf_ = sh.overflow, nit_ = sh.n_items;
    const int ntp_ = sh.n_touch2[(c - 1) & 1];             // (the PREVIOUS pick's touch count, stable until this pick's closing barrier: see the rebuild below)
    __builtin_amdgcn_sched_barrier(0);
    const bool use_list = ovf_ == 0;
    const int n_items = use_list ? nit_ : 0;
    const int n_ops = (n_items + 15) >> 4;                 // one evaluation instruction stream = 16 items, one per quad
    const bool kept = n_ops <= kG3Waves * kG3Keep;         // every item's samples stay in the registers of its wave
    int* touch_w = s_touch + ((c & 1) ? kG3Touch : 0);
    int* n_touch_w = &sh.n_touch2[c & 1];
    if constexpr (kIW) {
      if (!use_list) {
        // a candidate's list overflowed (the first picks): the in-wave sums of the others are void -- start over, brute force (below)
        if (tid < T) sh.delta[tid] = 0;
        __syncthreads();
      }
    }
    if (kIW && use_list) {
      // (evaluated in phase 1 by the candidates' own waves)
    } else if (use_list) {
      // (wave-uniform guards; a quad beyond the list reads leaf 0 and contributes nothing.  The candidate's colour is requested BEFORE the
      //  samples: left to the compiler that LDS read waits behind the L2 round trip)
      uint2 kc[kG3Keep];
#pragma unroll
      for (int s = 0; s < kG3Keep; ++s) {
        kw[s] = 0xffffffffu;
        ka[s] = make_uint4(0, 0, 0, 0);
        kb[s] = ka[s];
        kc[s] = make_uint2(0, 0);
        if (16 * (wave + s * kG3Waves) < n_items) {
          const int ii = 16 * (wave + s * kG3Waves) + quad;
          const uint32_t w = ii < n_items ? wpre[s] : 0xf0000000u;      // candidate numbers are < 16: the high nibble marks a padding quad (it reads leaf 0)
          kc[s] = sh.ck[(w >> 24) & (kTMaxI - 1)];
          __builtin_amdgcn_sched_barrier(0);
          const uint4* p4 = reinterpret_cast<const uint4*>(samp + ((w & 0xffffffu) << 4) + 4 * qj);
          ka[s] = p4[0];
          kb[s] = p4[1];
          kw[s] = w;
        }
      }
#pragma unroll
      for (int s = 0; s < kG3Keep; ++s) {
        if (16 * (wave + s * kG3Waves) < n_items) {
          const int t = (int)(kw[s] >> 24) & (kTMaxI - 1);
          const uint2 cc = kc[s];
          unsigned imp = j_eval4(cc.x, (int)cc.y, ka[s], kb[s]);
          imp = dpp_quad_sum((kw[s] >> 28) ? 0u : imp);
          if (qj == 0 && imp) atomicAdd(&sh.delta[t], (unsigned long long)imp);
        }
      }
      for (int o = wave + kG3Keep * kG3Waves; o < n_ops; o += kG3Waves) {   // more items than the registers hold
        const int ii = 16 * o + quad;
        const uint32_t w = ii < n_items ? items[ii] : 0xffffffffu;
        const bool on = w != 0xffffffffu;
        unsigned imp = 0;
        if (on) {
          const uint4* p4 = reinterpret_cast<const uint4*>(samp + ((w & 0xffffffu) << 4) + 4 * qj);
          const uint2 cc = sh.ck[w >> 24];
          imp = j_eval4(cc.x, (int)cc.y, p4[0], p4[1]);
        }
        imp = dpp_quad_sum(imp);
        if (on && qj == 0 && imp) atomicAdd(&sh.delta[w >> 24], (unsigned long long)imp);
      }
    } else {
      // the work list overflowed (the first picks: almost every leaf can still improve): brute force, every wave its
      // slice of the samples against four candidates at a time
      for (int t0 = 0; t0 < T; t0 += 4) {
        const uint2 c0 = sh.ck[t0], c1 = sh.ck[min(t0 + 1, T - 1)], c2 = sh.ck[min(t0 + 2, T - 1)], c3 = sh.ck[min(t0 + 3, T - 1)];
        unsigned long long a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        for (int m0 = wave * 64; m0 < np; m0 += kG3Threads) {
          const uint2 sv = samp[m0 + lane];
          const int cpv = j_cprime(sv.y);
          a0 += (unsigned)max(cpv - (int)c0.y + 2 * (int)__builtin_amdgcn_udot4(c0.x, sv.x, 0u, false), 0);
          a1 += (unsigned)max(cpv - (int)c1.y + 2 * (int)__builtin_amdgcn_udot4(c1.x, sv.x, 0u, false), 0);
          a2 += (unsigned)max(cpv - (int)c2.y + 2 * (int)__builtin_amdgcn_udot4(c2.x, sv.x, 0u, false), 0);
          a3 += (unsigned)max(cpv - (int)c3.y + 2 * (int)__builtin_amdgcn_udot4(c3.x, sv.x, 0u, false), 0);
        }
        a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2); a3 = wave_sum(a3);
        if (lane == 0) {
          if (a0) atomicAdd(&sh.delta[t0], a0);
          if (a1 && t0 + 1 < T) atomicAdd(&sh.delta[t0 + 1], a1);
          if (a2 && t0 + 2 < T) atomicAdd(&sh.delta[t0 + 2], a2);
          if (a3 && t0 + 3 < T) atomicAdd(&sh.delta[t0 + 3], a3);
        }
      }
    }
    WEND(2);
    STAMP(4);
#ifdef RHCCQ_STAMPS
    _acc[10] += (unsigned long long)n_items;
    _acc[11] += use_list ? (kept ? 0 : 1) : 0;
    _acc[12] += use_list ? 0 : 1;
#endif
    if (!(kIW && use_list)) __syncthreads();              // (in-wave variant: the sums were complete behind the first barrier)
    STAMP(5);
    WBEGIN();
    // ================= phase 4: greedy choice + commit =======================================================
    // largest reduction == smallest potential; the first candidate wins ties
    const uint32_t ckl = sh.ck[lane & (kTMaxI - 1)].x;      // (requested with the sums: one round trip)
    const unsigned long long dv = sh.delta[lane & (kTMaxI - 1)];
    __builtin_amdgcn_sched_barrier(0);
    // (the <= 16 sums sit in the first row of lanes: a row maximum and one readlane; and while the potential is below 2^32 -- it bounds
    //  every sum -- their high words are zero)
    static_assert(kTMaxI == 16, "the candidates' sums fill one row of lanes");
    const unsigned dhi = lane < T ? (unsigned)(dv >> 32) : 0u, dlo = lane < T ? (unsigned)dv : 0u;
    unsigned long long bd;
    if (__builtin_amdgcn_readfirstlane((int)(pot >> 32)) == 0) {
      bd = (unsigned)__builtin_amdgcn_readlane((int)dpp_row_max(dlo), 0);
    } else {
      const unsigned mhi = (unsigned)__builtin_amdgcn_readlane((int)dpp_row_max(dhi), 0);
      const unsigned mlo = (unsigned)__builtin_amdgcn_readlane((int)dpp_row_max(dhi == mhi ? dlo : 0u), 0);
      bd = ((unsigned long long)mhi << 32) | mlo;
    }
    const int best = __ffsll((long long)__ballot(lane < T && dv == bd)) - 1;
    const uint32_t kbest = (uint32_t)__builtin_amdgcn_readlane((int)ckl, best);
    const int nabest = (int)norm2_key(kbest);
    if (kIW && use_list) {
      // the winner's wave commits the leaves it still holds; leaves beyond those are shared out over all waves (their samples re-read)
      const int n_best = sh.n_it[best];
      if (wave == best) {
#pragma unroll
        for (int s = 0; s < kG3Keep; ++s)
          if (16 * s < n_best) {
            const bool mine = (kw[s] >> 28) == 0u;
            g3_commit_quad(mine, (int)(kw[s] & 0xffffffu), kbest, nabest, ka[s], kb[s], samp, dsamp, blk, dsum, dtop, touch_w, n_touch_w);
          }
      }
      for (int o = kG3Keep + wave; 16 * o < n_best; o += kG3Waves) {
        const int ii = 16 * o + quad;
        const bool mine = ii < n_best;
        const uint32_t w = items[best * cap_c + min(ii, n_best - 1)];
        uint4 a = make_uint4(0, 0, 0, 0), bb = a;
        if (mine) {
          const uint4* p4 = reinterpret_cast<const uint4*>(samp + ((w & 0xffffffu) << 4) + 4 * qj);
          a = p4[0];
          bb = p4[1];
        }
        g3_commit_quad(mine, (int)(w & 0xffffffu), kbest, nabest, a, bb, samp, dsamp, blk, dsum, dtop, touch_w, n_touch_w);
      }
    } else if (use_list && kept) {
#pragma unroll
      for (int s = 0; s < kG3Keep; ++s) {
        if (16 * (wave + s * kG3Waves) < n_items) {
          const bool mine = (int)(kw[s] >> 24) == best;     // (a padding quad's high nibble never matches)
          if (__ballot(mine))
            g3_commit_quad(mine, (int)(kw[s] & 0xffffffu), kbest, nabest, ka[s], kb[s], samp, dsamp, blk, dsum, dtop, touch_w, n_touch_w);
        }
      }
    } else if (use_list) {
      for (int o = wave; o < n_ops; o += kG3Waves) {
        const int ii = 16 * o + quad;
        const uint32_t w = ii < n_items ? items[ii] : 0xffffffffu;
        const bool mine = w != 0xffffffffu && (int)(w >> 24) == best;
        if (!__ballot(mine)) continue;
        uint4 a = make_uint4(0, 0, 0, 0), bb = a;
        if (mine) {
          const uint4* p4 = reinterpret_cast<const uint4*>(samp + ((w & 0xffffffu) << 4) + 4 * qj);
          a = p4[0];
          bb = p4[1];
        }
        g3_commit_quad(mine, (int)(w & 0xffffffu), kbest, nabest, a, bb, samp, dsamp, blk, dsum, dtop, touch_w, n_touch_w);
      }
    } else {
      // brute-force commit: one sample per lane, one row of lanes per leaf; every leaf maximum is rewritten, the levels
      // above are rebuilt behind the closing barrier (the touch counter is pushed over its capacity below)
      for (int m0 = wave * 64; m0 < np; m0 += kG3Threads) {
        const int m = m0 + lane;
        const uint2 sv = samp[m];
        const int e = 2 * (int)__builtin_amdgcn_udot4(kbest, sv.x, 0u, false);
        const int c0 = j_cprime(sv.y);
        const int i0 = c0 - nabest + e;
        const int w0 = i0 > 0 ? nabest - e : c0;
        const unsigned n0 = norm2_key(sv.x & 0xffffffu);
        g3_store1(i0, w0, n0, sv.x, sv.y, m, samp, dsamp, dsum, dtop);
        const unsigned mx = dpp_row_max((unsigned)(w0 + (int)n0));
        if (rj == 0) blk[m >> 4].w = mx;
      }
    }
    if (tid == 0) {
      sh.n_items = 0; sh.overflow = 0;
      sh.n_touch2[(c + 1) & 1] = 0;
    }
    pot -= bd;
    if (r_wave) {
      if (lane == 0) cho[c] = sh.cand[best];
      // (a potential below 2^32 -- the usual case -- converts in one instruction each way; the generic 64-bit conversions are ~30)
      if (lane < T)
        sh.R[lane] = (pot >> 32) == 0 ? (unsigned long long)(unsigned)ceil(u_next * (double)(unsigned)pot) : (unsigned long long)ceil(u_next * (double)pot);
    }
    WEND(3);
    STAMP(6);
    __syncthreads();
    STAMP(7);
    // a brute-force pick, or more touched leaves than the list held in the PREVIOUS pick: rebuild every super maximum.  (The overflow is
    // noticed one pick late, with a read that rides on phase 3's: asking for this pick's count here was a round trip of its own for every
    // wave between two picks.  Until then the supers the list could not name keep their old, larger maxima: conservative.)
    if (!use_list || ntp_ > kG3Touch) {
      for (int sb = tid; sb < nsb; sb += kG3Threads) {
        unsigned m = 0;
        for (int b = sb * 16; b < min(sb * 16 + 16, nb); ++b) m = max(m, blk[b].w);
        sup[sb].w = m;
        supw[sb] = m;
      }
      if (tid == 0) sh.n_touch2[c & 1] = 0;
      __syncthreads();
      for (int h = tid; h < nhyp; h += kG3Threads) {
        unsigned m = 0;
        for (int sb = h * 16; sb < min(h * 16 + 16, nsb); ++sb) m = max(m, sup[sb].w);
        hyp[h].w = m;
      }
      __syncthreads();
    }
  }
#ifdef RHCCQ_STAMPS
  if (tid == 0 && blockIdx.x == gridDim.x - 1)
    for (int i = 0; i < 16; ++i) g_init_stamps[i] += _acc[i];
  if (lane == 0 && blockIdx.x == gridDim.x - 1)
    for (int i = 0; i < 8; ++i) atomicAdd(&g_wave_stamps[i][wave], _wacc[i]);
#endif
  for (int j = tid; j < k; j += kG3Threads) {
    const uint32_t kk = dsamp[cho[j]].x;
    const double c0 = (double)key_r(kk), c1 = (double)key_g(kk), c2 = (double)key_b(kk);
    double* C = centres + (P.koff + j) * 4;
    C[0] = c0; C[1] = c1; C[2] = c2; C[3] = km64_csq(c0, c1, c2);
  }
}
