// Overlapped mini-batch steps for ONE straggler problem (included by k8_minibatch.hip behind the update kernel).
//
// A 4K frame's critical path ends in one MiniBatchKMeans problem that runs its full 100 * n / 1000 steps (~2 000) while every
// other problem has converged: three dependent launches per step (E-step over all k centres, fold, update), ~30 us, of which the
// E-step's 1 000 x k float64 distance evaluations are ~14 us.  But a step that does not reassign changes ONLY the centres its own
// batch touched (<= 1 000 of k, sklearn _minibatch_update_dense: a centre without members keeps its value), so almost all of the
// next E-step does not depend on this step's update:
//
//   launch A(t)   roles 0-3  update of step t, a quarter of the touched centres each (lean_update: the non-reassigning part of
//                            mbk_update_body)
//                 role 4  draws batch t + 2                     (one batch ahead of the classic sequence)
//                 role 5  batch inertia + EWA rule of step t    (mbk_inertia_block, unchanged)
//                 rest    SPECULATIVE E-step of step t + 1: batch t + 1 against every centre that step t does NOT touch
//                         (estep_tile with U_t = the labels of batch t, known before the launch: touched centres get +inf)
//   launch B(t+1) for every row of batch t + 1: first arg-min over the speculative tile minima and the <= 1 000 centres of U_t at
//                 their NEW values (read through the labels of batch t), the row's inertia term; -> labels of step t + 1
//
// Every distance is the same expression on the same operands as in mbk_batch_estep_kernel (cs + fma(x2, -2 c2, fma(x1, -2 c1,
// x0 * -2 c0))), and the arg-min over a union is the (distance, index)-lexicographic minimum of the parts' arg-mins, which is
// what "first index of the minimum" means: results are bit-identical to the classic sequence, the E-step just leaves the chain.
// Steps that reassign (every 10 k / 1000 steps once no centre has zero weight; the host knows them in advance and the device
// checks, UpdDraws::expect_reassign) and the step behind them run the classic E-step.

constexpr int kPipeThreads = 256;     // small workgroups with little LDS: they share the CUs with the other problems' kernels
constexpr int kUpdParts = 4;         // workgroups sharing the centre update: each owns the centres whose label hash names it
constexpr int kPipeRoles = kUpdParts + 2;
constexpr int kLeanCap = 8;           // batch rows listed per touched centre; beyond, the centre's thread walks the batch
constexpr int kLeanSlots = 1024;      // hash slots of one update workgroup: it owns a quarter of the <= 1 000 touched centres (all of them at worst)

struct LeanShared {
  int lab[kBatch];                    // labels of the batch (update role); the drawn row indices (draw role)
  uint32_t bkey[kBatch];
  int hkey[kLeanSlots];
  int hcnt[kLeanSlots];
  unsigned short hmem[kLeanSlots][kLeanCap];
  int ired[kPipeThreads / 64 + 1];
  long long cursor;
};
constexpr int kSpecPts = 2 * 64;      // batch rows per workgroup of the speculative E-step: two per lane; its four waves share the tile
constexpr int kSpecChunks = (1000 + kSpecPts - 1) / kSpecPts;
struct SpecShared {
  double bd[4][kSpecPts];
  int bj[4][kSpecPts];
  unsigned excl[kTileS / 32];
};
constexpr size_t kPipeLds = sizeof(LeanShared) > sizeof(SpecShared) ? sizeof(LeanShared) : sizeof(SpecShared);

__device__ __forceinline__ bool argmin_better(double d, int j, double bd, int bj) { return d < bd || (d == bd && j < bj); }

// update_center_dense of a step that does not reassign (the only kind the overlapped sequence hands to this role): the same
// operations in the same order as role 0 of mbk_update_body -- c * w, += x for the members in batch order, w += count,
// c *= 1 / w -- with 256 threads and 40 KB of LDS instead of 1 024 threads and 150 KB
__device__ __forceinline__ void lean_update(LeanShared& sh, const int p, const MbkP* __restrict__ probs, double* __restrict__ centres,
                                            double* __restrict__ weights, double* __restrict__ state, long long step,
                                            const uint32_t* __restrict__ bkeys_cur, const int32_t* __restrict__ labels_p, const int part) {
  const int tid = threadIdx.x;
  double* st = state + p * kStateStride;
  const double st_since = st[st_slot(kStSince, step)], st_nzero = st[st_slot(kStNzero, step)];
  const MbkP P = probs[p];
  if (mbk_stopped(st, step, P.n)) return;
  const int k = (int)P.k;
  const int bs = (int)min((long long)1000, P.n);
  const double since = st_since + (double)bs;
  if (st_nzero > 0.0 || since >= 10.0 * (double)k) {       // a reassigning step: the host's schedule and the device state disagree
    if (tid == 0) st[4] = 5.0;
    return;
  }
  double* C = centres + P.koff * 4;
  double* W = weights + P.koff;
  constexpr int kRows = kBatch / kPipeThreads, kSlotsPer = kLeanSlots / kPipeThreads;
  int lj[kRows];
  uint32_t lk[kRows];
#pragma unroll
  for (int q = 0; q < kRows; ++q) {
    const int b = tid + q * kPipeThreads;
    lj[q] = b < bs ? labels_p[b] : -1;
    lk[q] = b < bs ? bkeys_cur[(size_t)p * kBatch + b] : 0u;
  }
#pragma unroll
  for (int q = 0; q < kSlotsPer; ++q) { sh.hkey[tid + q * kPipeThreads] = -1; sh.hcnt[tid + q * kPipeThreads] = 0; }
#pragma unroll
  for (int q = 0; q < kRows; ++q) { sh.lab[tid + q * kPipeThreads] = lj[q]; sh.bkey[tid + q * kPipeThreads] = lk[q]; }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < kRows; ++q) {
    const int j = lj[q];
    if (j < 0) continue;
    const unsigned hh = (unsigned)j * 2654435761u;
    if (((hh >> 19) & (kUpdParts - 1)) != (unsigned)part) continue;      // another workgroup's centre
    unsigned h = hh >> 22;                              // 10 bits
    while (true) {
      int cur = sh.hkey[h];
      if (cur == -1) {
        const int old = atomicCAS(&sh.hkey[h], -1, j);
        cur = old == -1 ? j : old;
      }
      if (cur == j) break;
      h = (h + 1) & (kLeanSlots - 1);
    }
    const int pos = atomicAdd(&sh.hcnt[h], 1);
    if (pos < kLeanCap) sh.hmem[h][pos] = (unsigned short)(tid + q * kPipeThreads);
  }
  __syncthreads();
  // every touched centre: requested together (they were read for the E-step moments ago: L2), then updated one after the other
  int sj[kSlotsPer];
  double c0[kSlotsPer], c1[kSlotsPer], c2[kSlotsPer], cw[kSlotsPer];
#pragma unroll
  for (int q = 0; q < kSlotsPer; ++q) {
    sj[q] = sh.hkey[tid + q * kPipeThreads];
    c0[q] = c1[q] = c2[q] = cw[q] = 0.0;
    if (sj[q] >= 0) {
      const int j = sj[q];
      const double2 a = *reinterpret_cast<const double2*>(C + (size_t)j * 4);
      c0[q] = a.x; c1[q] = a.y; c2[q] = C[(size_t)j * 4 + 2]; cw[q] = W[j];
    }
  }
#pragma unroll
  for (int q = 0; q < kSlotsPer; ++q) {
    const int j = sj[q];
    if (j < 0) continue;
    const int h = tid + q * kPipeThreads, cnt = sh.hcnt[h];
    const double w = cw[q];
    double a0 = c0[q] * w, a1 = c1[q] * w, a2 = c2[q] * w;
    if (cnt <= kLeanCap) {
      int prev = -1;
      for (int i = 0; i < cnt; ++i) {                    // next member in ascending batch row
        int best = 0x7fffffff;
#pragma unroll
        for (int m = 0; m < kLeanCap; ++m) {
          const int r = m < cnt ? (int)sh.hmem[h][m] : 0x7fffffff;
          if (r > prev && r < best) best = r;
        }
        prev = best;
        const uint32_t kk = sh.bkey[best];
        a0 = a0 + (double)key_r(kk); a1 = a1 + (double)key_g(kk); a2 = a2 + (double)key_b(kk);
      }
    } else {
      walk_members(sh.lab, sh.bkey, bs, j, a0, a1, a2);
    }
    const double wn = w + (double)cnt;
    const double alpha = 1.0 / wn;
    const double n0 = a0 * alpha, n1 = a1 * alpha, n2 = a2 * alpha;
    C[(size_t)j * 4] = n0; C[(size_t)j * 4 + 1] = n1; C[(size_t)j * 4 + 2] = n2;
    C[(size_t)j * 4 + 3] = km64_csq(n0, n1, n2);
    W[j] = wn;
  }
  if (tid == 0 && part == 0) {
    st[st_slot(kStSince, step + 1)] = since;
    st[st_slot(kStNzero, step + 1)] = st_nzero;            // (0: it stays 0)
  }
}

// Speculative E-step of one (tile of 256 centres, 128 batch rows) unit: wave w walks centres 64 w .. 64 w + 63 of the tile for two
// rows per lane.  The centre of an iteration is the same for the whole wave, so it comes through the SCALAR cache (constant address
// space, wave-uniform index) straight into SGPR operands: no tile in LDS, no load phase in front of the loop, and the loop is float64
// VALU only (the LDS version's two 16-byte broadcast reads per centre and wave co-limited it).  The -2 of the distance expression
// moves from the centre to the row (x' = -2 x, exact: cs + fma(x2', c2, fma(x1', c1, x0' c0)) rounds like the tiled kernel's
// cs + fma(x2, -2 c2, fma(x1, -2 c1, x0 * -2 c0)), a power of two commutes with every rounding); a centre the labels of the running
// step name (its value is changing under this launch) gets the distance +inf through a 256-bit mask built in LDS.
__device__ __forceinline__ void spec_tile(const int unit, const int p, const MbkP& P, const long long po, const double* __restrict__ centres,
                                          const uint32_t* __restrict__ bkeys, double* __restrict__ pdist, int32_t* __restrict__ pidx,
                                          const int32_t* __restrict__ excl_labels, SpecShared& sp) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = unit / kSpecChunks, chunk = unit % kSpecChunks;
  const int j0 = tile * kTileS;
  if ((long long)j0 >= P.k) return;
  const int nj = (int)min((long long)kTileS, P.k - j0);
  const int bs = (int)min((long long)1000, P.n);
  const int b0 = chunk * kSpecPts + lane, b1 = b0 + 64;
  int ex[kBatch / kPipeThreads];
#pragma unroll
  for (int q = 0; q < kBatch / kPipeThreads; ++q) ex[q] = tid + q * kPipeThreads < bs ? excl_labels[tid + q * kPipeThreads] - j0 : -1;
  const uint32_t k0 = bkeys[(size_t)p * kBatch + min(b0, bs - 1)], k1 = bkeys[(size_t)p * kBatch + min(b1, bs - 1)];
  if (tid < kTileS / 32) sp.excl[tid] = 0u;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < kBatch / kPipeThreads; ++q)
    if (ex[q] >= 0 && ex[q] < nj) atomicOr(&sp.excl[ex[q] >> 5], 1u << (ex[q] & 31));
  __syncthreads();
  const unsigned long long mask = ((unsigned long long)sp.excl[2 * wave + 1] << 32) | sp.excl[2 * wave];
  const unsigned long long um = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(mask >> 32)) << 32) |
                                (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)mask);
  const double x0 = -2.0 * (double)key_r(k0), x1 = -2.0 * (double)key_g(k0), x2 = -2.0 * (double)key_b(k0);
  const double y0 = -2.0 * (double)key_r(k1), y1 = -2.0 * (double)key_g(k1), y2 = -2.0 * (double)key_b(k1);
  const int ja = __builtin_amdgcn_readfirstlane(wave * 64), jb = min(ja + 64, nj);
  typedef double double4v __attribute__((ext_vector_type(4)));
  typedef const double4v __attribute__((address_space(4))) * cptr;
  cptr C = (cptr)(centres + (P.koff + j0) * 4);
  double bd0 = INFINITY, bd1 = INFINITY;
  int bj0 = 0, bj1 = 0;
#pragma unroll 8
  for (int j = ja; j < jb; ++j) {
    const double4v cc = C[j];                                  // one 32-byte scalar load: (c0, c1, c2, csq)
    const double c0 = cc.x, c1 = cc.y, c2 = cc.z;
    const long long cb = ((um >> (j - ja)) & 1ull) ? 0x7ff0000000000000ll : __double_as_longlong(cc.w);
    const double cs = __longlong_as_double(cb);
    const double d0 = cs + km64_dot(x0, x1, x2, c0, c1, c2);
    const double d1 = cs + km64_dot(y0, y1, y2, c0, c1, c2);
    if (d0 < bd0) { bd0 = d0; bj0 = j; }
    if (d1 < bd1) { bd1 = d1; bj1 = j; }
  }
  sp.bd[wave][lane] = bd0; sp.bd[wave][64 + lane] = bd1;
  sp.bj[wave][lane] = bj0; sp.bj[wave][64 + lane] = bj1;
  __syncthreads();
  if (tid < kSpecPts) {
    const int b = chunk * kSpecPts + tid;
    if (b < bs) {
      double bd = sp.bd[0][tid];
      int bj = sp.bj[0][tid];
#pragma unroll
      for (int w = 1; w < 4; ++w) {                           // ascending waves = ascending centre index: strict '<'
        const double od = sp.bd[w][tid];
        if (od < bd) { bd = od; bj = sp.bj[w][tid]; }
      }
      pdist[po + (size_t)tile * kBatch + b] = bd;              // (+inf: every centre of the tile is touched)
      pidx[po + (size_t)tile * kBatch + b] = j0 + bj;
    }
  }
}

#ifdef RHCCQ_STAMPS
__device__ unsigned long long g_pipe_stamps[8];          // cycles of the update / draw / inertia role, [3] launches (problem 0)
#define PIPE_ROLE_END(r) do { if (threadIdx.x == 0 && blockIdx.y == 0) { atomicAdd(&g_pipe_stamps[r], clock64() - _t_pipe); if ((r) == 0) atomicAdd(&g_pipe_stamps[3], 1ull); } } while (0)
#else
#define PIPE_ROLE_END(r) do {} while (0)
#endif

__global__ __launch_bounds__(kPipeThreads) void mbk_pipe_kernel(const uint32_t* __restrict__ keys, const MbkP* __restrict__ probs,
                                                                double* __restrict__ centres, double* __restrict__ weights,
                                                                double* __restrict__ state, long long step,
                                                                const uint32_t* __restrict__ words, long long n_words,
                                                                const uint32_t* __restrict__ bkeys_cur, UpdDraws dr,
                                                                const int32_t* __restrict__ lab_cur, const double* __restrict__ pper_cur,
                                                                const uint32_t* __restrict__ bkeys_spec, double* __restrict__ pdist,
                                                                int32_t* __restrict__ pidx, const long long* __restrict__ part_off) {
  __shared__ __align__(16) unsigned char smem[kPipeLds];
  const int p = blockIdx.y, tid = threadIdx.x;
#ifdef RHCCQ_STAMPS
  const unsigned long long _t_pipe = clock64();
#endif
  if (blockIdx.x < kUpdParts) {
    lean_update(*reinterpret_cast<LeanShared*>(smem), p, probs, centres, weights, state, step, bkeys_cur, lab_cur + (size_t)p * kBatch,
                (int)blockIdx.x);
    if (blockIdx.x == 0) PIPE_ROLE_END(0);
    return;
  }
  const MbkP P = probs[p];
  double* st = state + p * kStateStride;
  if (mbk_stopped(st, step, P.n)) return;
  if (blockIdx.x == kUpdParts) {
    // ---- the batch ahead: minibatch_indices = random_state.randint(0, n_samples, batch_size) ---------------------------------
    LeanShared& sh = *reinterpret_cast<LeanShared*>(smem);
    long long cursor = (long long)st[st_slot(kStCursor, dr.draw_first - 1)];
    for (int q = 0; q < dr.draw_count; ++q) {
      const long long b = dr.draw_first + q;
      // (256 threads x 8 words: one round of the rejection replay covers a batch unless half the candidates are rejected)
      cursor = draw_batch<8>(keys, P, words, n_words, cursor, dr.ring[b & 3] + (size_t)p * kBatch, sh.lab, sh.ired, &sh.cursor);
      if (cursor < 0) break;
      if (tid == 0) st[st_slot(kStCursor, b)] = (double)cursor;
      __syncthreads();
    }
    if (tid == 0 && cursor < 0) st[4] = 3.0;               // word table exhausted (the host sizes it so that this cannot happen)
    PIPE_ROLE_END(1);
    return;
  }
  if (blockIdx.x == kUpdParts + 1) {
    mbk_inertia_block(P, st, step, pper_cur + (size_t)p * kBatch, nullptr);
    PIPE_ROLE_END(2);
    return;
  }
  // ---- speculative E-step of step + 1: every centre this step's batch does not touch -----------------------------------------
  spec_tile((int)blockIdx.x - kPipeRoles, p, P, part_off[p], centres, bkeys_spec, pdist, pidx, lab_cur + (size_t)p * kBatch,
            *reinterpret_cast<SpecShared*>(smem));
}

// launch B: labels of step `step` from the tile minima (classic E-step: all centres; speculative: the untouched ones) and, behind
// a speculative E-step, the centres touched by step - 1 at their new values; the rows' inertia terms.  4 rows per workgroup,
// 64 threads per row; two rounds of loads (labels / tile minima, then the centres they name), everything else in registers.
constexpr int kFixPts = 4, kFixGroups = 256 / kFixPts, kFixU = (1000 + kFixGroups - 1) / kFixGroups;
__global__ __launch_bounds__(256) void mbk_fix_kernel(const MbkP* __restrict__ probs, const double* __restrict__ state, long long step,
                                                      const double* __restrict__ centres, const uint32_t* __restrict__ bkeys,
                                                      const double* __restrict__ pdist, const int32_t* __restrict__ pidx,
                                                      const long long* __restrict__ part_off, double* __restrict__ pper,
                                                      const int32_t* __restrict__ lab_prev, int32_t* __restrict__ lab_out, int tile_c) {
  __shared__ double s_d[4][kFixPts], s_p[4][kFixPts];
  __shared__ int s_j[4][kFixPts];
  const int p = blockIdx.y;
  const MbkP P = probs[p];
  const long long po = part_off[p];
  if (mbk_stopped(state + p * 16, step, P.n)) return;
  const int n_tiles = (int)((P.k + tile_c - 1) / tile_c);
  const int bs = (int)min((long long)1000, P.n);
  const int tid = threadIdx.x, pt = tid & (kFixPts - 1), g = tid / kFixPts;
  const int b = min((int)blockIdx.x * kFixPts + pt, bs - 1);   // clamped lanes redo the last row
  const double* C = centres + P.koff * 4;
  // round 1: the row, this thread's tile minima, its share of the previous step's labels
  const uint32_t kk = bkeys[(size_t)p * kBatch + b];
  int jj[kFixU];
  if (lab_prev != nullptr) {
    const int32_t* lp = lab_prev + (size_t)p * kBatch;
#pragma unroll
    for (int q = 0; q < kFixU; ++q) jj[q] = lp[min(g + q * kFixGroups, bs - 1)];      // (a repeated entry changes nothing)
  }
  double bd = INFINITY;
  int bj = 0x7fffffff;
  for (int t = g; t < n_tiles; t += 4 * kFixGroups) {           // four tiles in flight (k = 20 000: 81 tiles of 256 centres, two per thread)
    double d[4];
    int j[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int tq = min(t + q * kFixGroups, n_tiles - 1);       // (a repeated tile changes nothing)
      d[q] = pdist[po + (size_t)tq * kBatch + b];
      j[q] = pidx[po + (size_t)tq * kBatch + b];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (argmin_better(d[q], j[q], bd, bj)) { bd = d[q]; bj = j[q]; }
  }
  const double x0 = (double)key_r(kk), x1 = (double)key_g(kk), x2 = (double)key_b(kk);
  // round 2: the centres those name
  double b0 = 0.0, b1 = 0.0, b2 = 0.0;                     // coordinates of this thread's best centre
  {
    const int j = bj == 0x7fffffff ? 0 : bj;
    const double2 a = *reinterpret_cast<const double2*>(C + (size_t)j * 4);
    b0 = a.x; b1 = a.y; b2 = C[(size_t)j * 4 + 2];
  }
  if (lab_prev != nullptr) {
    double2 ca[kFixU], cb[kFixU];
#pragma unroll
    for (int q = 0; q < kFixU; ++q) {
      ca[q] = *reinterpret_cast<const double2*>(C + (size_t)jj[q] * 4);
      cb[q] = *reinterpret_cast<const double2*>(C + (size_t)jj[q] * 4 + 2);
    }
#pragma unroll
    for (int q = 0; q < kFixU; ++q) {
      const double d = cb[q].y + km64_dot(x0, x1, x2, -2.0 * ca[q].x, -2.0 * ca[q].y, -2.0 * cb[q].x);
      if (argmin_better(d, jj[q], bd, bj)) { bd = d; bj = jj[q]; b0 = ca[q].x; b1 = ca[q].y; b2 = cb[q].x; }
    }
  }
  // the row's term of the batch inertia against the centres BEFORE this step's update (see mbk_fold_tiles_kernel), for this
  // thread's candidate; the winner's travels with it
  const double e0 = x0 - b0, e1 = x1 - b1, e2 = x2 - b2;
  double per = (e0 * e0 + e1 * e1) + e2 * e2;
  // the 64 threads of a row: 16 per wave (lane bits 2..5), then the four waves
#pragma unroll
  for (int o = kFixPts; o < 64; o <<= 1) {
    const double od = __shfl_xor(bd, o, 64), op = __shfl_xor(per, o, 64);
    const int oj = __shfl_xor(bj, o, 64);
    if (argmin_better(od, oj, bd, bj)) { bd = od; bj = oj; per = op; }
  }
  if ((tid & 63) < kFixPts) { s_d[tid >> 6][pt] = bd; s_j[tid >> 6][pt] = bj; s_p[tid >> 6][pt] = per; }
  __syncthreads();
  if (tid < kFixPts) {
#pragma unroll
    for (int w = 1; w < 4; ++w)
      if (argmin_better(s_d[w][pt], s_j[w][pt], bd, bj)) { bd = s_d[w][pt]; bj = s_j[w][pt]; per = s_p[w][pt]; }
    lab_out[(size_t)p * kBatch + b] = bj;
    pper[(size_t)p * kBatch + b] = per;
  }
}
