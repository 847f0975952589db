// K3 + K4: DBSCAN(eps, min_samples = 1) labels of a palette == connected components of the graph
// { (i,j) : |c_i - c_j|^2 <= eps^2 } numbered by smallest member index.
//
// Reference behaviour replaced: sklearn DBSCAN(eps/255, 1).fit_predict(palette/255) at
// encoder/compression/clustering.py:233-235 (KD-tree radius_neighbors materialising ~N^2 neighbour
// lists + Cython DFS).
//
// MI355X design (one workgroup per palette).  Two kernels share the algorithm:
//   eps_components_lds_kernel  palettes of <= RHCCQ_EPS_LDS_MAX points (every DBSCAN-branch palette of the
//       reference: N < 10 000): keys, parents, the cell permutation and a grid of up to 30^3 cells (packed u16
//       counters) all live in LDS; sparse palettes (n < 3 x occupied cells) switch to one thread per point.
//   eps_components_kernel      larger inputs of the generic entry point: same passes, arrays in global
//       memory, grid <= 16^3.
//   K3  fixed-radius neighbour pass over the RGB grid.  Cells have side s with 3 (s-1)^2 <= eps^2 whenever that
//       keeps the grid within its cap: then all points of a cell are mutually within eps (a clique) and only
//       ONE witness pair per pair of nearby cells is needed; a wave searches a cell pair 64 candidate pairs at
//       a time and stops at the first hit (wave ballot).  For small eps (cell side at its minimum) every pair
//       of nearby cells is tested exhaustively.
//   K4  lock-free union-find in LDS (root = smallest index, atomicCAS linking, path halving), then a
//       flag + block prefix-sum turns roots into sklearn's label order.
// The eps predicate is the exact integer d2 <= thr; pairs with d2 == boundary (eps^2 integral) are
// decided by the same float64 expression the KD-tree leaf test uses, with FMA contraction disabled.
// Not a dense contraction: no MFMA; the work is integer VALU + LDS.
#include "rhccq_common.h"

namespace rhccq {

constexpr int kEpsThreads = 1024;   // (round 4: was 512 -- a lone level-3 palette of ~5 000 colours keeps one CU busy for 1.5 ms: four waves per SIMD instead of two)
constexpr int kMaxCells = 4096;  // 16^3 (global-memory variant)

struct EpsArrays {
  uint32_t* keys;     // [n]
  uint32_t* parent;   // [n]
  uint32_t* perm;     // [n]  points ordered by cell, later reused for flags / ranks
};

__device__ __forceinline__ uint32_t uf_find(uint32_t* parent, uint32_t i) {
  while (true) {
    uint32_t p = parent[i];
    if (p == i) return i;
    uint32_t gp = parent[p];
    if (gp != p) parent[i] = gp;  // path halving; racy writes only ever move towards an ancestor
    i = p;
  }
}

// read-only root lookup (labelling phase: no thread may write parent[] while others still walk it)
__device__ __forceinline__ uint32_t uf_root(const uint32_t* parent, uint32_t i) {
  while (true) {
    const uint32_t p = parent[i];
    if (p == i) return i;
    i = p;
  }
}

__device__ __forceinline__ void uf_union(uint32_t* parent, uint32_t a, uint32_t b) {
  while (true) {
    a = uf_find(parent, a);
    b = uf_find(parent, b);
    if (a == b) return;
    if (a < b) { uint32_t t = a; a = b; b = t; }       // a = larger root goes under b
    if (atomicCAS(&parent[a], a, b) == a) return;
  }
}

// float64 boundary test: sum_k (a_k/255 - b_k/255)^2 <= r2, sequential, no FMA
__device__ __forceinline__ bool boundary_neighbor(uint32_t ka, uint32_t kb, double r2) {
  double d = 0.0;
  {
    double t = __dsub_rn(__ddiv_rn((double)key_r(ka), 255.0), __ddiv_rn((double)key_r(kb), 255.0));
    d = __dadd_rn(d, __dmul_rn(t, t));
  }
  {
    double t = __dsub_rn(__ddiv_rn((double)key_g(ka), 255.0), __ddiv_rn((double)key_g(kb), 255.0));
    d = __dadd_rn(d, __dmul_rn(t, t));
  }
  {
    double t = __dsub_rn(__ddiv_rn((double)key_b(ka), 255.0), __ddiv_rn((double)key_b(kb), 255.0));
    d = __dadd_rn(d, __dmul_rn(t, t));
  }
  return d <= r2;
}

__device__ __forceinline__ bool is_neighbor(uint32_t ka, uint32_t kb, int thr, int boundary, double r2) {
  const int d2 = dist2_keys(ka, kb);
  if (d2 <= thr) return true;
  if (d2 == boundary) return boundary_neighbor(ka, kb, r2);
  return false;
}

template <bool kLds>
__device__ void eps_components_body(const uint32_t* __restrict__ gkeys, int n, int thr, int boundary, double r2,
                                    int32_t* __restrict__ labels_out, int32_t* __restrict__ ncomp_out, EpsArrays A,
                                    unsigned* cell_start /* [kMaxCells+1] */, unsigned* cell_fill /* [kMaxCells] */,
                                    unsigned* red /* [16] */) {
  const int tid = threadIdx.x;
  // effective squared radius for geometric pruning
  const int reach2 = boundary >= 0 ? boundary : thr;
  // clique cell side: largest s with 3 (s-1)^2 <= reach2 (conservative for the boundary case: use thr)
  int s = 1;
  while (3 * s * s <= thr) ++s;                         // now 3 (s-1)^2 <= thr < 3 s^2
  const bool clique = s >= 16;
  if (!clique) s = 16;
  const int G = (255 / s) + 1;                          // cells per axis (<= 16)
  const int n_cells = G * G * G;

  for (int i = tid; i < n; i += blockDim.x) {
    A.keys[i] = gkeys[i];
    A.parent[i] = i;
  }
  for (int c = tid; c <= n_cells; c += blockDim.x) cell_start[c] = 0;
  for (int c = tid; c < n_cells; c += blockDim.x) cell_fill[c] = 0;
  __syncthreads();
  // histogram
  for (int i = tid; i < n; i += blockDim.x) {
    const uint32_t k = A.keys[i];
    const int cell = ((int)key_r(k) / s * G + (int)key_g(k) / s) * G + (int)key_b(k) / s;
    atomicAdd(&cell_start[cell + 1], 1u);
  }
  __syncthreads();
  // inclusive scan over n_cells+1 entries (<= 4097): each thread owns 9 consecutive entries
  {
    constexpr int per = (kMaxCells + 1 + kEpsThreads - 1) / kEpsThreads;  // 9
    unsigned loc[per];
    unsigned sum = 0;
    for (int j = 0; j < per; ++j) {
      int c = tid * per + j;
      loc[j] = c <= n_cells ? cell_start[c] : 0u;
      sum += loc[j];
    }
    unsigned tot;
    unsigned base = block_exscan<unsigned>(sum, red, &tot);
    for (int j = 0; j < per; ++j) {
      int c = tid * per + j;
      base += loc[j];
      if (c <= n_cells) cell_start[c] = base;        // cell_start[c] = #points in cells < c
    }
  }
  __syncthreads();
  for (int i = tid; i < n; i += blockDim.x) {
    const uint32_t k = A.keys[i];
    const int cell = ((int)key_r(k) / s * G + (int)key_g(k) / s) * G + (int)key_b(k) / s;
    const unsigned pos = cell_start[cell] + atomicAdd(&cell_fill[cell], 1u);
    A.perm[pos] = i;
  }
  __syncthreads();
  if (clique) {
    // every point joins the first-listed point of its cell
    for (int p = tid; p < n; p += blockDim.x) {
      const uint32_t i = A.perm[p];
      const uint32_t k = A.keys[i];
      const int cell = ((int)key_r(k) / s * G + (int)key_g(k) / s) * G + (int)key_b(k) / s;
      const uint32_t rep = A.perm[cell_start[cell]];
      if (rep != i) uf_union(A.parent, i, rep);
    }
    __syncthreads();
  }
  // neighbour pass: one wave per (cell A, forward neighbour offset) task
  int R = 0;
  while ((long long)(R * s) * (R * s) <= reach2 + 0ll && R < G) ++R;  // cells farther than R apart cannot hold a pair
  // a pair of cells dx apart on an axis has min coordinate gap max(0, (dx-1)*s + 1)
  const int span = 2 * R + 1;
  const int n_off = span * span * span;
  const int lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
  const long long n_tasks = (long long)n_cells * n_off;
  for (long long t = wave; t < n_tasks; t += n_waves) {
    const int ca = (int)(t / n_off);
    const unsigned a0 = cell_start[ca], na = cell_start[ca + 1] - a0;
    if (na == 0) { continue; }
    const int o = (int)(t % n_off);
    const int dz = o % span - R, dy = (o / span) % span - R, dx = o / (span * span) - R;
    // forward half only (each unordered cell pair once); same cell handled when !clique
    if (dx < 0 || (dx == 0 && (dy < 0 || (dy == 0 && dz < 0)))) continue;
    const bool same = dx == 0 && dy == 0 && dz == 0;
    if (same && clique) continue;
    const int ax = ca / (G * G), ay = (ca / G) % G, az = ca % G;
    const int bx = ax + dx, by = ay + dy, bz = az + dz;
    if (bx < 0 || bx >= G || by < 0 || by >= G || bz < 0 || bz >= G) continue;
    const int cb = (bx * G + by) * G + bz;
    const unsigned b0 = cell_start[cb], nb = cell_start[cb + 1] - b0;
    if (nb == 0) continue;
    {
      const int gx = dx ? (abs(dx) - 1) * s + 1 : 0, gy = dy ? (abs(dy) - 1) * s + 1 : 0, gz = dz ? (abs(dz) - 1) * s + 1 : 0;
      if (gx * gx + gy * gy + gz * gz > reach2) continue;
    }
    if (clique) {
      const uint32_t repa = A.perm[a0], repb = A.perm[b0];
      if (uf_find(A.parent, repa) == uf_find(A.parent, repb)) continue;  // already connected
      const unsigned total = na * nb;
      for (unsigned base = 0; base < total; base += 64) {
        const unsigned p = base + lane;
        bool hit = false;
        if (p < total) {
          const uint32_t ka = A.keys[A.perm[a0 + p / nb]], kb = A.keys[A.perm[b0 + p % nb]];
          hit = is_neighbor(ka, kb, thr, boundary, r2);
        }
        if (__any(hit)) {
          if (lane == 0) uf_union(A.parent, repa, repb);
          break;
        }
      }
    } else {
      const unsigned total = na * nb;
      for (unsigned base = 0; base < total; base += 64) {
        const unsigned p = base + lane;
        if (p < total) {
          const unsigned ia = p / nb, ib = p % nb;
          if (!same || ia < ib) {
            const uint32_t i = A.perm[a0 + ia], j = A.perm[b0 + ib];
            if (is_neighbor(A.keys[i], A.keys[j], thr, boundary, r2)) uf_union(A.parent, i, j);
          }
        }
      }
    }
  }
  __syncthreads();
  // K4 labelling: root = smallest member index; label = rank of the root among roots
  // roots first into perm[] (read-only walk), then parent[] = root, perm[] = is_root
  for (int i = tid; i < n; i += blockDim.x) A.perm[i] = uf_root(A.parent, i);
  __syncthreads();
  for (int i = tid; i < n; i += blockDim.x) {
    const uint32_t r = A.perm[i];
    A.parent[i] = r;
    A.perm[i] = r == (uint32_t)i ? 1u : 0u;
  }
  __syncthreads();
  {
    const int per = (n + blockDim.x - 1) / blockDim.x;
    const int lo = tid * per, hi = min(lo + per, n);
    unsigned sum = 0;
    for (int i = lo; i < hi; ++i) sum += A.perm[i];
    unsigned tot;
    unsigned base = block_exscan<unsigned>(sum, red, &tot);
    for (int i = lo; i < hi; ++i) {
      const unsigned f = A.perm[i];
      A.perm[i] = base;                                 // exclusive rank of index i among roots
      base += f;
    }
    if (tid == 0) *ncomp_out = (int32_t)tot;
  }
  __syncthreads();
  for (int i = tid; i < n; i += blockDim.x) labels_out[i] = (int32_t)A.perm[A.parent[i]];
}

// ---- fast path: the whole problem in LDS, up to 30^3 cells held as packed u16 pairs ------------------
constexpr int kFastG = 30;
constexpr int kFastCells = kFastG * kFastG * kFastG;      // 27000
constexpr int kFastWords = (kFastCells + 2 + 1) / 2;      // u16 cell table (cells + sentinel) as u32 words

__device__ __forceinline__ unsigned cell16(const uint32_t* tab, int c) { return (tab[c >> 1] >> ((c & 1) * 16)) & 0xffffu; }

__global__ __launch_bounds__(kEpsThreads) void eps_components_lds_kernel(const uint32_t* __restrict__ keys, const int32_t* __restrict__ desc,
                                                                          const double* __restrict__ r2v, int32_t* __restrict__ labels_out,
                                                                          int32_t* __restrict__ ncomp_out) {
  __shared__ uint32_t s_keys[RHCCQ_EPS_LDS_MAX];
  __shared__ uint32_t s_parent[RHCCQ_EPS_LDS_MAX];
  __shared__ unsigned short s_perm[RHCCQ_EPS_LDS_MAX];
  __shared__ uint32_t s_cell[kFastWords];
  __shared__ unsigned s_red[20];
  const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = kEpsThreads / 64;
  const int off = desc[p * 4 + 0], n = desc[p * 4 + 1], thr = desc[p * 4 + 2], boundary = desc[p * 4 + 3];
  if (n <= 0) {
    if (tid == 0) ncomp_out[p] = 0;
    return;
  }
  if (n > RHCCQ_EPS_LDS_MAX) return;                    // handled by eps_components_kernel
  const double r2 = r2v[p];
  const int reach2 = boundary >= 0 ? boundary : thr;
  // cell side: the clique side (3 (s-1)^2 <= thr) when the grid then fits, else the smallest side that fits
  int s = 1;
  while (3 * s * s <= thr) ++s;
  const int s_min = (255 / kFastG) + 1;                 // 9
  const bool clique = s >= s_min;
  if (!clique) s = s_min;
  const int G = (255 / s) + 1, n_cells = G * G * G;
  auto cell_of = [&](uint32_t k) { return ((int)key_r(k) / s * G + (int)key_g(k) / s) * G + (int)key_b(k) / s; };
  for (int i = tid; i < n; i += kEpsThreads) { s_keys[i] = keys[off + i]; s_parent[i] = i; }
  for (int w = tid; w < kFastWords; w += kEpsThreads) s_cell[w] = 0;
  __syncthreads();
  // histogram into entry cell+0 (packed u16), inclusive scan -> ends, scatter by decrementing ends -> starts
  for (int i = tid; i < n; i += kEpsThreads) {
    const int c = cell_of(s_keys[i]);
    atomicAdd(&s_cell[c >> 1], 1u << ((c & 1) * 16));
  }
  __syncthreads();
  {
    const int per = (n_cells + kEpsThreads - 1) / kEpsThreads;
    const int lo = min(tid * per, n_cells), hi = min(lo + per, n_cells);
    unsigned sum = 0;
    for (int c = lo; c < hi; ++c) sum += cell16(s_cell, c);
    unsigned tot;
    unsigned base = block_exscan<unsigned>(sum, s_red, &tot);
    __syncthreads();
    // two neighbouring threads may share a u32 word: write through 16-bit stores
    unsigned short* c16 = reinterpret_cast<unsigned short*>(s_cell);
    for (int c = lo; c < hi; ++c) {
      base += c16[c];
      c16[c] = (unsigned short)base;                    // end of cell c
    }
    if (tid == 0) c16[n_cells] = (unsigned short)n;    // sentinel
  }
  __syncthreads();
  for (int i = tid; i < n; i += kEpsThreads) {
    const int c = cell_of(s_keys[i]);
    const unsigned old = atomicSub(&s_cell[c >> 1], 1u << ((c & 1) * 16));
    const unsigned pos = ((old >> ((c & 1) * 16)) & 0xffffu) - 1u;
    s_perm[pos] = (unsigned short)i;
  }
  __syncthreads();
  // now cell16(c) = start of cell c, cell16(c+1) = its end (the next start / the sentinel)
  if (clique) {
    for (int q = tid; q < n; q += kEpsThreads) {
      const uint32_t i = s_perm[q];
      const uint32_t rep = s_perm[cell16(s_cell, cell_of(s_keys[i]))];
      if (rep != i) uf_union(s_parent, i, rep);
    }
    __syncthreads();
  }
  int R = 0;
  while ((long long)(R * s) * (R * s) <= reach2 + 0ll && R < G) ++R;
  const int span = 2 * R + 1, n_off = span * span * span;
  // sparse palettes (about one point per occupied cell, e.g. already-quantised level-2/3 palettes): one
  // THREAD per point walks the neighbouring cells and unions with every later-listed neighbour in range;
  // dense palettes: one WAVE per occupied cell pair (below)
  int n_heads = 0;
  for (int q = tid; q < n; q += kEpsThreads) n_heads += (int)cell16(s_cell, cell_of(s_keys[s_perm[q]])) == q;
  n_heads = block_sum<int>(n_heads, reinterpret_cast<int*>(s_red));
  const bool point_mode = n < 3 * n_heads;
  if (point_mode) {
    for (int q = tid; q < n; q += kEpsThreads) {
      const uint32_t i = s_perm[q];
      const uint32_t ki = s_keys[i];
      const int A = cell_of(ki);
      const int ax = A / (G * G), ay = (A / G) % G, az = A % G;
      for (int bx = max(ax - R, 0); bx <= min(ax + R, G - 1); ++bx)
        for (int by = max(ay - R, 0); by <= min(ay + R, G - 1); ++by)
          for (int bz = max(az - R, 0); bz <= min(az + R, G - 1); ++bz) {
            const int cb = (bx * G + by) * G + bz;
            const unsigned b0 = cell16(s_cell, cb), b1 = cell16(s_cell, cb + 1);
            for (unsigned e = b0; e < b1; ++e) {
              if ((int)e <= q) continue;                 // each unordered pair once (by position in the cell order)
              const uint32_t j = s_perm[e];
              if (is_neighbor(ki, s_keys[j], thr, boundary, r2)) uf_union(s_parent, i, j);
            }
          }
    }
  }
  // one wave per non-empty cell A (found as the heads of the cell-sorted point list); its lanes test the
  // neighbour offsets 64 at a time, then the wave searches each surviving cell pair
  for (int q0 = wave * 64; q0 < n && !point_mode; q0 += n_waves * 64) {
    const int q = q0 + lane;
    int ca = -1;
    if (q < n) {
      const int c = cell_of(s_keys[s_perm[q]]);
      if ((int)cell16(s_cell, c) == q) ca = c;          // q is the first point of its cell
    }
    unsigned long long heads = __ballot(ca >= 0);
    while (heads) {
      const int hl = __ffsll((long long)heads) - 1;
      heads &= heads - 1;
      const int A = __shfl(ca, hl, 64);
      const unsigned a0 = cell16(s_cell, A), na = cell16(s_cell, A + 1) - a0;
      const int ax = A / (G * G), ay = (A / G) % G, az = A % G;
      const uint32_t repa = s_perm[a0];
      for (int ob = 0; ob < n_off; ob += 64) {
        const int o = ob + lane;
        int B = -1;
        if (o < n_off) {
          const int dz = o % span - R, dy = (o / span) % span - R, dx = o / (span * span) - R;
          const bool fwd = dx > 0 || (dx == 0 && (dy > 0 || (dy == 0 && dz >= 0)));
          const bool same = dx == 0 && dy == 0 && dz == 0;
          const int bx = ax + dx, by = ay + dy, bz = az + dz;
          if (fwd && !(same && clique) && bx >= 0 && bx < G && by >= 0 && by < G && bz >= 0 && bz < G) {
            const int gx = dx ? (abs(dx) - 1) * s + 1 : 0, gy = dy ? (abs(dy) - 1) * s + 1 : 0, gz = dz ? (abs(dz) - 1) * s + 1 : 0;
            const int cb = (bx * G + by) * G + bz;
            if (gx * gx + gy * gy + gz * gz <= reach2 && cell16(s_cell, cb + 1) > cell16(s_cell, cb)) B = cb;
          }
        }
        unsigned long long mb = __ballot(B >= 0);
        while (mb) {
          const int bl = __ffsll((long long)mb) - 1;
          mb &= mb - 1;
          const int cb = __shfl(B, bl, 64);
          const unsigned b0 = cell16(s_cell, cb), nbp = cell16(s_cell, cb + 1) - b0;
          const unsigned total = na * nbp;
          if (clique) {
            const uint32_t repb = s_perm[b0];
            if (uf_find(s_parent, repa) == uf_find(s_parent, repb)) continue;
            for (unsigned base = 0; base < total; base += 64) {
              const unsigned pp = base + lane;
              bool hit = false;
              if (pp < total) hit = is_neighbor(s_keys[s_perm[a0 + pp / nbp]], s_keys[s_perm[b0 + pp % nbp]], thr, boundary, r2);
              if (__any(hit)) {
                if (lane == 0) uf_union(s_parent, repa, repb);
                break;
              }
            }
          } else {
            const bool same = cb == A;
            for (unsigned base = 0; base < total; base += 64) {
              const unsigned pp = base + lane;
              if (pp < total) {
                const unsigned ia = pp / nbp, ib = pp % nbp;
                if (!same || ia < ib) {
                  const uint32_t i = s_perm[a0 + ia], j = s_perm[b0 + ib];
                  if (is_neighbor(s_keys[i], s_keys[j], thr, boundary, r2)) uf_union(s_parent, i, j);
                }
              }
            }
          }
        }
      }
    }
  }
  __syncthreads();
  // K4 labelling: root = smallest member index; label = rank of the root among roots
  {
    constexpr int kPer = (RHCCQ_EPS_LDS_MAX + kEpsThreads - 1) / kEpsThreads;   // 20 points per thread
    uint32_t root[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const int i = tid + j * kEpsThreads;
      root[j] = i < n ? uf_root(s_parent, i) : 0u;     // nobody writes parent[] here
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const int i = tid + j * kEpsThreads;
      if (i < n) { s_parent[i] = root[j]; s_perm[i] = root[j] == (uint32_t)i ? 1 : 0; }
    }
  }
  __syncthreads();
  {
    const int per = (n + kEpsThreads - 1) / kEpsThreads;
    const int lo = min(tid * per, n), hi = min(lo + per, n);
    unsigned sum = 0;
    for (int i = lo; i < hi; ++i) sum += s_perm[i];
    unsigned tot;
    unsigned base = block_exscan<unsigned>(sum, s_red, &tot);
    for (int i = lo; i < hi; ++i) {
      const unsigned f = s_perm[i];
      s_perm[i] = (unsigned short)base;
      base += f;
    }
    if (tid == 0) ncomp_out[p] = (int32_t)tot;
  }
  __syncthreads();
  for (int i = tid; i < n; i += kEpsThreads) labels_out[off + i] = (int32_t)s_perm[s_parent[i]];
}

__global__ __launch_bounds__(kEpsThreads) void eps_components_kernel(const uint32_t* __restrict__ keys, const int32_t* __restrict__ desc,
                                                                      const double* __restrict__ r2, int32_t* __restrict__ labels_out,
                                                                      int32_t* __restrict__ ncomp_out, uint32_t* __restrict__ gwork,
                                                                      int work_stride) {
  __shared__ unsigned s_cell_start[kMaxCells + 1];
  __shared__ unsigned s_cell_fill[kMaxCells];
  __shared__ unsigned s_red[20];
  const int p = blockIdx.x;
  const int off = desc[p * 4 + 0], n = desc[p * 4 + 1], thr = desc[p * 4 + 2], boundary = desc[p * 4 + 3];
  if (n <= 0) {
    if (threadIdx.x == 0) ncomp_out[p] = 0;
    return;
  }
  EpsArrays A;
  if (n <= RHCCQ_EPS_LDS_MAX) {
    return;                                             // handled by eps_components_lds_kernel
  } else {
    uint32_t* w = gwork + (size_t)p * 3 * work_stride;
    A.keys = w; A.parent = w + work_stride; A.perm = w + 2 * (size_t)work_stride;
    eps_components_body<false>(keys + off, n, thr, boundary, r2[p], labels_out + off, ncomp_out + p, A, s_cell_start, s_cell_fill, s_red);
  }
}

// ---- DBSCAN with min_samples > 1 (clustering.py:233-271; no call site of the pipeline uses it, the function's own default is 2):
// neighbour counts (self included) by brute force over LDS tiles -- a palette on this branch has < 10 000 colours -- then, with the
// core points' component labels known (the kernel above on the core subset), every other point takes the smallest label among its
// core neighbours (sklearn's dbscan_inner reaches a border point first from the cluster with the lowest seed), -1 = noise.
__global__ __launch_bounds__(256) void eps_count_kernel(const uint32_t* __restrict__ keys, int n, int thr, int boundary, double r2, int32_t* __restrict__ counts) {
  __shared__ uint32_t tile[1024];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const uint32_t ka = i < n ? keys[i] : 0u;
  int cnt = 0;
  for (int t0 = 0; t0 < n; t0 += 1024) {
    for (int j = threadIdx.x; j < 1024; j += 256) tile[j] = t0 + j < n ? keys[t0 + j] : 0u;
    __syncthreads();
    const int m = min(1024, n - t0);
    if (i < n)
      for (int j = 0; j < m; ++j) cnt += is_neighbor(ka, tile[j], thr, boundary, r2) ? 1 : 0;
    __syncthreads();
  }
  if (i < n) counts[i] = cnt;
}

__global__ __launch_bounds__(256) void eps_border_kernel(const uint32_t* __restrict__ keys, int n, int thr, int boundary, double r2,
                                                         const int32_t* __restrict__ core_label /* -1: not a core point */, int32_t* __restrict__ out) {
  __shared__ uint32_t tile[1024];
  __shared__ int32_t tlab[1024];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const uint32_t ka = i < n ? keys[i] : 0u;
  const int mine = i < n ? core_label[i] : 0;
  int best = 0x7fffffff;
  for (int t0 = 0; t0 < n; t0 += 1024) {
    for (int j = threadIdx.x; j < 1024; j += 256) {
      tile[j] = t0 + j < n ? keys[t0 + j] : 0u;
      tlab[j] = t0 + j < n ? core_label[t0 + j] : -1;
    }
    __syncthreads();
    const int m = min(1024, n - t0);
    if (i < n && mine < 0)
      for (int j = 0; j < m; ++j)
        if (tlab[j] >= 0 && tlab[j] < best && is_neighbor(ka, tile[j], thr, boundary, r2)) best = tlab[j];
    __syncthreads();
  }
  if (i < n) out[i] = mine >= 0 ? mine : (best == 0x7fffffff ? -1 : best);
}

}  // namespace rhccq

using namespace rhccq;

extern "C" int rhccq_eps_components(rhccq_ctx* ctx, const uint32_t* keys, const int32_t* desc, const double* r2, int32_t n_prob,
                                    int32_t max_n, int32_t* labels_out, int32_t* ncomp_out) {
  if (!ctx || !keys || !desc || !r2 || !labels_out || !ncomp_out || n_prob <= 0 || max_n < 0)
    return rhccq_fail(ctx, RHCCQ_E_ARG, "eps_components: bad argument");
  uint32_t* gwork = nullptr;
  int stride = 0;
  if (max_n > RHCCQ_EPS_LDS_MAX) {
    // oversize problems run the same code from global memory (slow path, API completeness)
    stride = (max_n + 63) & ~63;
    const size_t bytes = (size_t)n_prob * 3 * stride * sizeof(uint32_t);
    if (ctx->scratch_bytes < bytes) {
      if (ctx->scratch) RHCCQ_HIP(ctx, hipFree(ctx->scratch));
      ctx->scratch = nullptr;
      ctx->scratch_bytes = 0;
      RHCCQ_HIP(ctx, hipMalloc(&ctx->scratch, bytes));
      ctx->scratch_bytes = bytes;
    }
    gwork = (uint32_t*)ctx->scratch;
  }
  hipLaunchKernelGGL(eps_components_lds_kernel, dim3(n_prob), dim3(kEpsThreads), 0, ctx->stream, keys, desc, r2, labels_out, ncomp_out);
  if (max_n > RHCCQ_EPS_LDS_MAX)
    hipLaunchKernelGGL(eps_components_kernel, dim3(n_prob), dim3(kEpsThreads), 0, ctx->stream, keys, desc, r2, labels_out, ncomp_out, gwork, stride);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

extern "C" int rhccq_eps_counts(rhccq_ctx* ctx, const uint32_t* keys, int32_t n, int32_t thr, int32_t boundary, double r2, int32_t* counts) {
  if (!ctx || !keys || !counts || n <= 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "eps_counts: bad argument");
  hipLaunchKernelGGL(eps_count_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, keys, n, thr, boundary, r2, counts);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

extern "C" int rhccq_eps_border(rhccq_ctx* ctx, const uint32_t* keys, int32_t n, int32_t thr, int32_t boundary, double r2, const int32_t* core_label,
                                int32_t* labels_out) {
  if (!ctx || !keys || !core_label || !labels_out || n <= 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "eps_border: bad argument");
  hipLaunchKernelGGL(eps_border_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, keys, n, thr, boundary, r2, core_label, labels_out);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}
