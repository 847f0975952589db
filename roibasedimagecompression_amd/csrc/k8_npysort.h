// k8_npysort.h -- included by k8_minibatch.hip (namespace rhccq, after UpdShared).
//
// WHICH low-count centres a capped reassignment takes.  sklearn's _mini_batch_step keeps np.argsort(weight_sums)[:batch / 2]
// of the candidates (reference call site encoder/compression/clustering.py:211-218); np.argsort is numpy's UNSTABLE quicksort and
// the weights are massively tied, so the answer is whatever numpy's sort kernel happens to leave in the first batch / 2 slots.
// The host setting of record is numpy's scalar kernel, numpy/_core/src/npysort/quicksort.cpp
// aquicksort_<double> (+ heapsort.cpp aheapsort_ behind the depth limit), restated on the CPU in oracle/npy_argsort.c and
// emulated here EXACTLY, in parallel:
//
//   * only the set of the first `cap` slots matters (the reassigned centres are written through a boolean mask), so of
//     quicksort's recursion only the ONE chain of partitions that still straddles slot `cap` is followed (a quickselect
//     that performs numpy's partitions and nothing else): a part that lies wholly below `cap` keeps its members however it
//     is sorted, a part wholly above never enters;
//   * one partition = numpy's median of three (positions pl, pm, pr), pivot parked at pr - 1, then the Hoare scan.  The scan
//     is sequential in numpy but its outcome is not: pi stops at the positions > pl whose weight is >= the pivot's in
//     ascending order (L-stops), pj at the positions < pr - 1 whose weight is <= the pivot's in descending order (R-stops),
//     both taken from the array AS IT WAS (a position is visited by at most one of the two before they cross), the m-th
//     swap exchanges L[m] and R[m] while L[m] < R[m], and with M swaps done pi ends at min(L[M], R[M - 1]) (L[0] if M = 0).
//     So: two ballot/popcount rank passes over the part (a wave owns a contiguous stretch), one scatter of the stop lists,
//     one pass of independent swaps;
//   * elements are (weight << 32 | centre) words: the weights are sample counts, integers below 100 n <= 2^32 (n <= 2^24
//     colours exist), so one 64-bit word carries what numpy's index array and value array carry together;
//   * parts of at most 16 elements are insertion-sorted and parts behind the depth limit (2 floor(log2 k), counted as numpy
//     counts it: only a part that was PUSHED on its stack is checked) heap-sorted by one lane, operation for operation.
//
// The arrays start in global memory (L2-resident: 16 B per centre, any k); once the part that straddles slot `cap` has shrunk to what the
// caller's spare LDS holds (the update's member tables: 7 680 elements) it moves there -- every level is a handful of dependent round
// trips and barriers, ~1 us each through the L2, a tenth of that in LDS: ~114 us per call at k = 30 128 in global memory alone.  Checked against oracle/npy_argsort.c on its own (rhccq_npysort_head, with the depth limit lowered
// to drive the heapsort branch) and through whole fits against scikit-learn's untouched fit_predict (G11 scalar records).

struct QsScratch {
  unsigned long long* e;   // [k] (weight << 32 | centre), permuted in place
  int* lpos;               // [k] L-stops of the current partition, ascending
  int* rpos;               // [k] R-stops, descending
  unsigned* mask;          // [(k + 31) / 32] out: bit j set <=> centre j sits in the first `cap` slots
};

__device__ __forceinline__ unsigned qs_w(unsigned long long e) { return (unsigned)(e >> 32); }

// numpy's insertion sort of a small part (quicksort.cpp, tail of aquicksort_), one lane
__device__ inline void qs_insertion(unsigned long long* e, int pl, int pr) {
  for (int pi = pl + 1; pi <= pr; ++pi) {
    const unsigned long long vi = e[pi];
    const unsigned vp = qs_w(vi);
    int pj = pi;
    while (pj > pl && vp < qs_w(e[pj - 1])) { e[pj] = e[pj - 1]; --pj; }
    e[pj] = vi;
  }
}

// numpy's aheapsort_ (heapsort.cpp) on e0[0 .. n), one lane
__device__ inline void qs_heapsort(unsigned long long* e0, int n) {
  unsigned long long* a = e0 - 1;   // 1-based
  int i, j, l;
  unsigned long long tmp;
  for (l = n >> 1; l > 0; --l) {
    tmp = a[l];
    for (i = l, j = l << 1; j <= n;) {
      if (j < n && qs_w(a[j]) < qs_w(a[j + 1])) j += 1;
      if (qs_w(tmp) < qs_w(a[j])) { a[i] = a[j]; i = j; j += j; }
      else break;
    }
    a[i] = tmp;
  }
  for (; n > 1;) {
    tmp = a[n];
    a[n] = a[1];
    n -= 1;
    for (i = 1, j = 2; j <= n;) {
      if (j < n && qs_w(a[j]) < qs_w(a[j + 1])) j++;
      if (qs_w(tmp) < qs_w(a[j])) { a[i] = a[j]; i = j; j += j; }
      else break;
    }
    a[i] = tmp;
  }
}

// All kUpdThreads threads of the workgroup.  W[0 .. k): the weights; cap < k.  depth0 < 0: numpy's limit 2 floor(log2 k).
// On return q.mask names the centres np.argsort(W)[:cap] holds (as a set).  Uses sh.weq / sh.wsel / sh.ired.
// lds / lds_elems: spare LDS of 16 * lds_elems bytes (nullptr: none).
__device__ __forceinline__ void npysort_head(UpdShared& sh, const double* __restrict__ W, const int k, const int cap, const QsScratch& q0,
                                             const int depth0, void* lds = nullptr, const int lds_elems = 0) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long below = (1ull << lane) - 1ull;
  QsScratch q = q0;                                      // (the arrays move to LDS on the way down)
  int off = 0;                                           // position x of the part lives at index x - off of the current arrays (an explicit
                                                         //  offset: a pointer re-based BELOW the start of an LDS array leaves the LDS aperture)
  int moved_at = 0x7fffffff;                             // positions >= moved_at live in LDS
  for (int j = tid; j < k; j += kUpdThreads) q.e[j] = ((unsigned long long)(unsigned)W[j] << 32) | (unsigned)j;
  for (int j = tid; j < (k + 31) / 32; j += kUpdThreads) q.mask[j] = 0u;
  __syncthreads();
  int pl = 0, pr = k - 1;
  int cdepth = depth0 >= 0 ? depth0 : 2 * (31 - __clz(k));
  bool pushed = true;                                    // (numpy tests the depth on entry and whenever it pops a part)
  while (pr >= cap && pl < cap) {
    if (lds != nullptr && moved_at == 0x7fffffff && pr - pl + 1 <= lds_elems) {
      // the part fits the spare LDS: copy it once; from here on position x lives at index x - off
      unsigned long long* le = reinterpret_cast<unsigned long long*>(lds);
      int* ll = reinterpret_cast<int*>(le + lds_elems);
      for (int x = pl + tid; x <= pr; x += kUpdThreads) le[x - pl] = q.e[x];
      q.e = le;
      q.lpos = ll;
      q.rpos = ll + lds_elems;
      off = pl;
      moved_at = pl;
      __syncthreads();
    }
    if (pushed && cdepth < 0) {
      if (tid == 0) qs_heapsort(q.e + (pl - off), pr - pl + 1);
      break;
    }
    if (pr - pl <= 15) {
      if (tid == 0) qs_insertion(q.e, pl - off, pr - off);
      break;
    }
    // ---- median of three, pivot to pr - 1 (every thread evaluates it, thread 0 stores)
    const int pm = pl + ((pr - pl) >> 1);
    unsigned long long a = q.e[(pl) - off], b = q.e[(pm) - off], c = q.e[(pr) - off];
    const unsigned long long d = q.e[(pr - 1) - off];
    unsigned long long t;
    if (qs_w(b) < qs_w(a)) { t = a; a = b; b = t; }
    if (qs_w(c) < qs_w(b)) { t = c; c = b; b = t; }
    if (qs_w(b) < qs_w(a)) { t = a; a = b; b = t; }
    const unsigned vp = qs_w(b);
    __syncthreads();
    if (tid == 0) { q.e[(pl) - off] = a; q.e[(pr) - off] = c; q.e[(pm) - off] = d; q.e[(pr - 1) - off] = b; }
    __syncthreads();
    // ---- stop lists: wave w owns positions [x0, x1)
    const int m = pr - pl + 1;
    const int S = (((m + kUpdWaves - 1) / kUpdWaves) + 63) & ~63;
    const int x0 = pl + wave * S, x1 = min(x0 + S, pr + 1);
    int cL = 0, cR = 0;
    for (int xb = x0; xb < x1; xb += 64) {
      const int x = xb + lane;
      const bool valid = x < x1;
      const unsigned w = valid ? qs_w(q.e[(x) - off]) : 0u;
      cL += __popcll(__ballot(valid && x > pl && x < pr && w >= vp));
      cR += __popcll(__ballot(valid && x < pr - 1 && w <= vp));
    }
    if (lane == 0) { sh.weq[wave] = cL; sh.wsel[wave] = cR; }
    __syncthreads();
    int runL = 0, runR = 0, nL = 0, nR = 0;
    for (int w = 0; w < kUpdWaves; ++w) {
      if (w < wave) { runL += sh.weq[w]; runR += sh.wsel[w]; }
      nL += sh.weq[w];
      nR += sh.wsel[w];
    }
    for (int xb = x0; xb < x1; xb += 64) {
      const int x = xb + lane;
      const bool valid = x < x1;
      const unsigned w = valid ? qs_w(q.e[(x) - off]) : 0u;
      const bool fl = valid && x > pl && x < pr && w >= vp, fr = valid && x < pr - 1 && w <= vp;
      const unsigned long long bL = __ballot(fl), bR = __ballot(fr);
      if (fl) q.lpos[(pl + runL + __popcll(bL & below)) - off] = x;
      if (fr) q.rpos[(pl + nR - 1 - (runR + __popcll(bR & below))) - off] = x;
      runL += __popcll(bL);
      runR += __popcll(bR);
    }
    __syncthreads();
    // ---- the swaps: pair m exchanges while L[m] < R[m] (a prefix of the pairs)
    const int np = min(nL, nR);
    int cnt = 0;
    for (int mm = tid; mm < np; mm += kUpdThreads) {
      const int l = q.lpos[(pl + mm) - off], r = q.rpos[(pl + mm) - off];
      if (l < r) {
        const unsigned long long el = q.e[(l) - off], er = q.e[(r) - off];
        q.e[(l) - off] = er;
        q.e[(r) - off] = el;
        ++cnt;
      }
    }
    const int M = block_sum<int>(cnt, sh.ired);
    const int pi = M >= 1 ? min(q.lpos[(pl + M) - off], q.rpos[(pl + M - 1) - off]) : q.lpos[(pl) - off];
    __syncthreads();
    if (tid == 0) {
      const unsigned long long ei = q.e[(pi) - off];
      q.e[(pi) - off] = q.e[(pr - 1) - off];
      q.e[(pr - 1) - off] = ei;
    }
    __syncthreads();
    --cdepth;
    const bool left_smaller = (pi - pl) < (pr - pi);      // numpy pushes the LARGER part and goes on with the other
    if (pi >= cap) { pushed = !left_smaller; pr = pi - 1; }
    else { pushed = left_smaller; pl = pi + 1; }
  }
  __syncthreads();
  for (int mm = tid; mm < cap; mm += kUpdThreads) {
    const unsigned j = (unsigned)(mm >= moved_at ? q.e[(mm) - off] : q0.e[mm]);      // (slots below the move were final when it happened)
    atomicOr(&q.mask[j >> 5], 1u << (j & 31));
  }
  __syncthreads();
}
