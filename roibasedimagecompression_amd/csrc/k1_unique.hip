// K0 / K0b / K1 / K6: per-job unique colours through a 2^24-bit bitmap per job.
//
// Reference behaviour replaced: np.unique(pixels, axis=0) + the per-pixel dict loop of
// get_all_unique_colors (encoder/compression/clustering.py:21-48), the per-segment mask / bbox /
// crop of subregion_quantization (encoder/compression/subregions.py:317-383), its black-in-segment
// fix (:393-421) and the index remap gather (clustering.py:373-377).
//
// MI355X design: a colour is a 24-bit key, so "sorted unique + rank" needs no sort: one streaming
// pass sets bits (test-before-atomicOr; the bitmaps of the jobs of a frame live in L2 / Infinity
// Cache), a popcount scan turns the bitmap into ranks (== np.unique order because the key order is
// the lexicographic R,G,B order), and every later pass recomputes a pixel's palette index as
//   (word, prefix) = word_prefix[key >> 5];  prefix + popc(word & lower_mask)        (one 8-byte gather)
// instead of storing a 4 B/pixel index map.  All per-pixel passes are HBM-streaming kernels:
// 4 pixels per thread, 12 B of RGB as three dwords + one int4 of labels per class.
#include <hipcub/hipcub.hpp>

#include "rhccq_common.h"

namespace rhccq {

constexpr int kMaxClass = 4;

struct ClassArgs {
  const int32_t* labels[kMaxClass];
  int32_t job_base[kMaxClass];
  int32_t n_class;
};

// ---- tiny per-block hash table of job statistics (LDS), flushed with global atomics ------------
constexpr int kStatSlots = 64;
struct StatTable {
  int job[kStatSlots];
  int minr[kStatSlots], maxr[kStatSlots], minc[kStatSlots], maxc[kStatSlots];
  unsigned cnt[kStatSlots], nblack[kStatSlots];
};

// a thread's running statistics of the job it is currently inside (per class); flushed to the block's
// LDS table only when the job changes -- with large segments that is once or twice per thread
struct RunStat {
  int job, minr, maxr, minc, maxc;
  unsigned cnt, nblack;
};

__device__ __forceinline__ void stat_flush(StatTable& t, int32_t* gstats, const RunStat& rs) {
  if (rs.job < 0) return;
  unsigned h = ((unsigned)rs.job * 2654435761u) >> 26;  // 6 bits
  for (int probe = 0; probe < kStatSlots; ++probe) {
    int s = (h + probe) & (kStatSlots - 1);
    int cur = t.job[s];
    if (cur == -1) {
      int old = atomicCAS(&t.job[s], -1, rs.job);
      cur = (old == -1) ? rs.job : old;
    }
    if (cur == rs.job) {
      atomicMin(&t.minr[s], rs.minr);
      atomicMax(&t.maxr[s], rs.maxr);
      atomicMin(&t.minc[s], rs.minc);
      atomicMax(&t.maxc[s], rs.maxc);
      atomicAdd(&t.cnt[s], rs.cnt);
      if (rs.nblack) atomicAdd(&t.nblack[s], rs.nblack);
      return;
    }
  }
  // table full: straight to global memory
  int32_t* g = gstats + (size_t)rs.job * 6;
  atomicMin(&g[0], rs.minr);
  atomicMax(&g[1], rs.maxr);
  atomicMin(&g[2], rs.minc);
  atomicMax(&g[3], rs.maxc);
  atomicAdd((unsigned*)&g[4], rs.cnt);
  if (rs.nblack) atomicAdd((unsigned*)&g[5], rs.nblack);
}

__device__ __forceinline__ void load4px(const uint8_t* rgb, int64_t p0, int64_t n_px, uint32_t key[4]) {
  // 4 pixels = 12 bytes starting at a 4-byte aligned address when p0 % 4 == 0 and rgb is 4-B aligned
  if (p0 + 4 <= n_px) {
    const uint32_t* q = reinterpret_cast<const uint32_t*>(rgb + p0 * 3);
    uint32_t a = q[0], b = q[1], c = q[2];
    // bytes: a = r0 g0 b0 r1 | b = g1 b1 r2 g2 | c = b2 r3 g3 b3   (little endian)
    key[0] = ((a & 255u) << 16) | (((a >> 8) & 255u) << 8) | ((a >> 16) & 255u);
    key[1] = ((a >> 24) << 16) | ((b & 255u) << 8) | ((b >> 8) & 255u);
    key[2] = (((b >> 16) & 255u) << 16) | ((b >> 24) << 8) | (c & 255u);
    key[3] = (((c >> 8) & 255u) << 16) | (((c >> 16) & 255u) << 8) | (c >> 24);
  } else {
    for (int i = 0; i < 4; ++i) {
      int64_t p = p0 + i;
      key[i] = p < n_px ? ((uint32_t)rgb[p * 3] << 16) | ((uint32_t)rgb[p * 3 + 1] << 8) | rgb[p * 3 + 2] : 0u;
    }
  }
}

__device__ __forceinline__ void load4lab(const int32_t* lab, int64_t p0, int64_t n_px, int32_t out[4]) {
  if (lab == nullptr) {
    out[0] = out[1] = out[2] = out[3] = 1;
  } else if (p0 + 4 <= n_px) {
    int4 v = *reinterpret_cast<const int4*>(lab + p0);
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
  } else {
    for (int i = 0; i < 4; ++i) out[i] = (p0 + i < n_px) ? lab[p0 + i] : 0;
  }
}

// kBytes = true: colour flags are BYTES (one per colour, 16 MiB per job) written with plain stores --
// idempotent, so no atomics are needed and the per-XCD L2s merge them by byte mask; bytemap_pack_kernel
// then folds them into the 2 MiB bitmaps.  kBytes = false: bits set with test-before-atomicOr (scattered
// device-scope atomics execute at the memory side on MI355X: use only when 16 MiB per job is too much).
template <bool kBytes>
__global__ __launch_bounds__(256) void job_scan_kernel(const uint8_t* __restrict__ rgb, int H, int W, ClassArgs ca,
                                                       int black_is_colour, uint32_t* bitmaps, uint8_t* bytemaps,
                                                       int32_t* stats) {
  __shared__ StatTable tab;
  for (int i = threadIdx.x; i < kStatSlots; i += blockDim.x) {
    tab.job[i] = -1;
    tab.minr[i] = INT_MAX; tab.maxr[i] = -1; tab.minc[i] = INT_MAX; tab.maxc[i] = -1;
    tab.cnt[i] = 0; tab.nblack[i] = 0;
  }
  __syncthreads();
  const int64_t n_px = (int64_t)H * W;
  const int64_t n_quads = (n_px + 3) >> 2;
  // each block owns a contiguous run of quads so that its pixels touch few jobs
  const int64_t per_block = (n_quads + gridDim.x - 1) / gridDim.x;
  const int64_t q_begin = (int64_t)blockIdx.x * per_block;
  const int64_t q_end = min(q_begin + per_block, n_quads);
  RunStat rs[kMaxClass];
#pragma unroll
  for (int c = 0; c < kMaxClass; ++c) rs[c].job = -1;
  for (int64_t q = q_begin + threadIdx.x; q < q_end; q += blockDim.x) {
    const int64_t p0 = q << 2;
    uint32_t key[4];
    load4px(rgb, p0, n_px, key);
    const int r0 = (int)(p0 / W), c0 = (int)(p0 - (int64_t)r0 * W);
#pragma unroll
    for (int c = 0; c < kMaxClass; ++c) {
      if (c >= ca.n_class) break;
      int32_t lab[4];
      load4lab(ca.labels[c], p0, n_px, lab);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (lab[i] <= 0 || p0 + i >= n_px) continue;
        const int job = ca.job_base[c] + lab[i] - 1;
        int r = r0, col = c0 + i;
        while (col >= W) { col -= W; ++r; }                // a quad spans at most two rows when W >= 4
        const bool black = key[i] == 0u;
        if (job != rs[c].job) {
          stat_flush(tab, stats, rs[c]);
          rs[c].job = job; rs[c].minr = r; rs[c].maxr = r; rs[c].minc = col; rs[c].maxc = col; rs[c].cnt = 0; rs[c].nblack = 0;
        }
        rs[c].minr = min(rs[c].minr, r); rs[c].maxr = max(rs[c].maxr, r);
        rs[c].minc = min(rs[c].minc, col); rs[c].maxc = max(rs[c].maxc, col);
        rs[c].cnt += 1; rs[c].nblack += black;
        if (!black || black_is_colour) {
          if (kBytes) {
            bytemaps[((size_t)job << 24) + key[i]] = 1;
          } else if (bitmaps != nullptr) {                    // (NULL: statistics only, rhccq_job_stats)
            uint32_t* wptr = bitmaps + (size_t)job * RHCCQ_BITMAP_WORDS + (key[i] >> 5);
            const uint32_t bit = 1u << (key[i] & 31u);
            if ((*wptr & bit) == 0u) atomicOr(wptr, bit);   // bits are only ever set: a stale read costs one redundant atomic
          }
        }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < kMaxClass; ++c) stat_flush(tab, stats, rs[c]);
  __syncthreads();
  for (int s = threadIdx.x; s < kStatSlots; s += blockDim.x) {
    if (tab.job[s] >= 0) {
      int32_t* g = stats + (size_t)tab.job[s] * 6;
      atomicMin(&g[0], tab.minr[s]);
      atomicMax(&g[1], tab.maxr[s]);
      atomicMin(&g[2], tab.minc[s]);
      atomicMax(&g[3], tab.maxc[s]);
      atomicAdd((unsigned*)&g[4], tab.cnt[s]);
      if (tab.nblack[s]) atomicAdd((unsigned*)&g[5], tab.nblack[s]);
    }
  }
}

// 32 colour bytes -> one bitmap word (ORed into the bitmap so that job_set_black / earlier tiles survive)
__global__ __launch_bounds__(256) void bytemap_pack_kernel(const uint8_t* __restrict__ bytemaps, uint32_t* __restrict__ bitmaps, size_t n_words) {
  for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (size_t)gridDim.x * blockDim.x) {
    const uint4 a = reinterpret_cast<const uint4*>(bytemaps)[w * 2], b = reinterpret_cast<const uint4*>(bytemaps)[w * 2 + 1];
    const uint32_t v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t word = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      // bytes are 0 or 1: gather bit 0 of each of the 4 bytes into 4 adjacent bits
      const uint32_t x = v[i];
      word |= ((x & 1u) | ((x >> 7) & 2u) | ((x >> 14) & 4u) | ((x >> 21) & 8u)) << (4 * i);
    }
    if (word) bitmaps[w] |= word;
  }
}

__global__ void job_set_black_kernel(uint32_t* bitmaps, const int32_t* jobs, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicOr(bitmaps + (size_t)jobs[i] * RHCCQ_BITMAP_WORDS, 1u);
}

// ---- bitmap -> ranks ---------------------------------------------------------------------------
constexpr int kChunkWords = 1024;                      // 4 KiB of bitmap per block
constexpr int kChunks = RHCCQ_BITMAP_WORDS / kChunkWords;  // 512 per job

__global__ __launch_bounds__(256) void bitmap_chunk_count_kernel(const uint32_t* __restrict__ bitmaps,
                                                                 uint32_t* __restrict__ chunk_sums) {
  __shared__ unsigned red[4];
  const size_t job = blockIdx.y;
  const uint4 v = reinterpret_cast<const uint4*>(bitmaps + job * RHCCQ_BITMAP_WORDS + (size_t)blockIdx.x * kChunkWords)[threadIdx.x];
  unsigned c = __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
  unsigned t = block_sum<unsigned>(c, red);
  if (threadIdx.x == 0) chunk_sums[job * kChunks + blockIdx.x] = t;
}

// in-place exclusive scan of the 512 chunk sums of each job; counts[job] = total
__global__ __launch_bounds__(512) void bitmap_chunk_scan_kernel(uint32_t* __restrict__ chunk_sums, int32_t* __restrict__ counts) {
  __shared__ unsigned red[9];
  const size_t job = blockIdx.x;
  unsigned v = chunk_sums[job * kChunks + threadIdx.x];
  unsigned tot;
  unsigned ex = block_exscan<unsigned>(v, red, &tot);
  chunk_sums[job * kChunks + threadIdx.x] = ex;
  if (threadIdx.x == 0) counts[job] = (int32_t)tot;
}

__global__ __launch_bounds__(256) void bitmap_emit_kernel(const uint32_t* __restrict__ bitmaps, const uint32_t* __restrict__ chunk_base,
                                                          const int64_t* __restrict__ pal_off, uint32_t* __restrict__ word_prefix,
                                                          uint32_t* __restrict__ keys_out) {
  __shared__ unsigned red[5];
  const size_t job = blockIdx.y;
  const size_t w0 = (size_t)blockIdx.x * kChunkWords + (size_t)threadIdx.x * 4;
  const uint4 v = reinterpret_cast<const uint4*>(bitmaps + job * RHCCQ_BITMAP_WORDS)[w0 >> 2];
  const unsigned c0 = __popc(v.x), c1 = __popc(v.y), c2 = __popc(v.z), c3 = __popc(v.w);
  unsigned tot;
  unsigned ex = block_exscan<unsigned>(c0 + c1 + c2 + c3, red, &tot);
  if (tot == 0) return;                                 // empty chunk: its prefixes are never read
  const unsigned base = chunk_base[job * kChunks + blockIdx.x] + ex;
  // (bitmap word, exclusive prefix) pairs: the per-pixel rank lookup of K1d / K6 is ONE 8-byte gather
  uint4* wp = reinterpret_cast<uint4*>(word_prefix + 2 * (job * RHCCQ_BITMAP_WORDS + w0));
  wp[0] = make_uint4(v.x, base, v.y, base + c0);
  wp[1] = make_uint4(v.z, base + c0 + c1, v.w, base + c0 + c1 + c2);
  if (keys_out) {
    uint32_t* out = keys_out + pal_off[job];
    uint32_t words[4] = {v.x, v.y, v.z, v.w};
    unsigned rank = base;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t m = words[j];
      while (m) {
        int b = __ffs(m) - 1;
        m &= m - 1;
        out[rank++] = (uint32_t)((w0 + j) << 5) | (uint32_t)b;
      }
    }
  }
}

// ---- K0b: black-in-segment fix -----------------------------------------------------------------
__global__ __launch_bounds__(256) void job_blackfix_kernel(const uint8_t* __restrict__ rgb, int H, int W, ClassArgs ca,
                                                           const uint8_t* __restrict__ needs_fix, unsigned long long* __restrict__ best) {
  const int64_t n_px = (int64_t)H * W;
  const int64_t n_quads = (n_px + 3) >> 2;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_quads; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p0 = q << 2;
    uint32_t key[4];
    load4px(rgb, p0, n_px, key);
    for (int c = 0; c < ca.n_class; ++c) {
      int32_t lab[4];
      load4lab(ca.labels[c], p0, n_px, lab);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (lab[i] <= 0 || p0 + i >= n_px || key[i] == 0u) continue;
        const int job = ca.job_base[c] + lab[i] - 1;
        if (!needs_fix[job]) continue;
        const unsigned r = key_r(key[i]), g = key_g(key[i]), b = key_b(key[i]);
        const unsigned long long v = ((unsigned long long)(r * r + g * g + b * b) << 40) | (unsigned long long)(p0 + i);
        if (v < best[job]) atomicMin(&best[job], v);       // monotone: a stale read costs one redundant atomic
      }
    }
  }
}

// ---- many-segment frames: unique colours by ONE device sort ------------------------------------------------------------------
// The bitmap path keeps 6 MiB of tables per job (2 MiB bitmap + 4 MiB of (word, prefix) pairs): fine for the handful of segments a
// 4K frame is cut into, 12 GB at 2 048 jobs, impossible for a fine grid.  Here every masked pixel of every class becomes one
// 64-bit key (job << 24 | colour), one radix sort (rocPRIM through hipCUB) puts equal (job, colour) pairs next to each other in
// (job, R, G, B) order = np.unique order per job, a head flag + exclusive scan numbers the distinct pairs, and every pixel's rank
// inside its job's palette is STORED (int32 per pixel and class) instead of being recomputed from a bitmap.  Memory, whatever the
// number of jobs: ~40 B of scratch per (pixel, class) entry (keys in / out 2 x 8 B, values in / out 2 x 4 B, head flags + their scan
// 2 x 4 B, the radix sort's own scratch ~8 B: rhccq_job_sort_unique_bytes -- 0.66 GB for a 4K frame with two classes) + 4 B of rank.
__global__ __launch_bounds__(256) void sort_keys_kernel(const uint8_t* __restrict__ rgb, int H, int W, ClassArgs ca, const uint32_t* __restrict__ fix_key,
                                                        const int32_t* __restrict__ black_jobs, int n_black, int invalid_shift,
                                                        unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals) {
  const int64_t n_px = (int64_t)H * W;
  const int64_t total = (int64_t)ca.n_class * n_px + n_black;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    unsigned long long kv = 1ull << invalid_shift;         // sorts behind every real key
    if (i >= (int64_t)ca.n_class * n_px) {
      kv = (unsigned long long)black_jobs[i - (int64_t)ca.n_class * n_px] << 24;      // colour 0 of a job whose crop shows background
    } else {
      const int c = (int)(i / n_px);
      const int64_t p = i - (int64_t)c * n_px;
      const int32_t lab = ca.labels[c] ? ca.labels[c][p] : 1;
      if (lab > 0) {
        const int job = ca.job_base[c] + lab - 1;
        uint32_t k = ((uint32_t)rgb[p * 3] << 16) | ((uint32_t)rgb[p * 3 + 1] << 8) | rgb[p * 3 + 2];
        if (k == 0u && fix_key) k = fix_key[job];
        kv = ((unsigned long long)job << 24) | k;
      }
    }
    keys[i] = kv;
    vals[i] = (uint32_t)i;
  }
}

__global__ __launch_bounds__(256) void sort_heads_kernel(const unsigned long long* __restrict__ keys, int64_t total, int invalid_shift,
                                                         uint32_t* __restrict__ heads) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const unsigned long long k = keys[i];
  heads[i] = ((k >> invalid_shift) == 0ull && (i == 0 || keys[i - 1] != k)) ? 1u : 0u;
}

// first element of every job: where its palette starts in the list of distinct (job, colour) pairs; the palette keys themselves
__global__ __launch_bounds__(256) void sort_emit_kernel(const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ heads,
                                                        const uint32_t* __restrict__ uidx, int64_t total, int invalid_shift,
                                                        int32_t* __restrict__ job_start, uint32_t* __restrict__ keys_out, int32_t* __restrict__ n_unique) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const unsigned long long k = keys[i];
  const bool valid = (k >> invalid_shift) == 0ull;
  if (valid && heads[i]) {
    keys_out[uidx[i]] = (uint32_t)(k & 0xffffffull);
    if (i == 0 || (keys[i - 1] >> 24) != (k >> 24)) job_start[(int)(k >> 24)] = (int32_t)uidx[i];
  }
  if (i == total - 1) *n_unique = (int32_t)(uidx[i] + heads[i]);
}

__global__ __launch_bounds__(256) void sort_ranks_kernel(const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                         const uint32_t* __restrict__ heads, const uint32_t* __restrict__ uidx, int64_t total,
                                                         int64_t n_pixel_entries, int invalid_shift, const int32_t* __restrict__ job_start,
                                                         int32_t* __restrict__ rankmap) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const uint32_t v = vals[i];
  if ((int64_t)v >= n_pixel_entries) return;               // a synthetic black entry: no pixel behind it
  const unsigned long long k = keys[i];
  if ((k >> invalid_shift) != 0ull) { rankmap[v] = -1; return; }
  rankmap[v] = (int32_t)(uidx[i] + heads[i] - 1u) - job_start[(int)(k >> 24)];
}

// ---- rank lookup shared by K1d and K6 ----------------------------------------------------------
__device__ __forceinline__ uint32_t rank_of(const uint32_t* __restrict__ word_prefix, int job, uint32_t key) {
  const uint2 wp = reinterpret_cast<const uint2*>(word_prefix)[(size_t)job * RHCCQ_BITMAP_WORDS + (key >> 5)];
  return wp.y + __popc(wp.x & ((1u << (key & 31u)) - 1u));
}

__global__ __launch_bounds__(256) void job_index_kernel(const uint8_t* __restrict__ rgb, int H, int W, ClassArgs ca,
                                                        const uint32_t* __restrict__ bitmaps, const uint32_t* __restrict__ word_prefix,
                                                        const int64_t* __restrict__ pal_off, const uint32_t* __restrict__ fix_key,
                                                        int32_t* __restrict__ idx_out, int32_t* first_pos,
                                                        const int32_t* __restrict__ fp_lut, const int32_t* __restrict__ rankmap,
                                                        int32_t* __restrict__ entry_out) {
  const int64_t n_px = (int64_t)H * W;
  const int64_t n_quads = (n_px + 3) >> 2;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_quads; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p0 = q << 2;
    uint32_t key[4] = {0u, 0u, 0u, 0u};
    if (rgb) load4px(rgb, p0, n_px, key);                  // (ranked path: no pixel colour is read at all -- rgb == nullptr, uniform branch)
    for (int c = 0; c < ca.n_class; ++c) {
      int32_t lab[4];
      load4lab(ca.labels[c], p0, n_px, lab);
      int32_t res[4], ent[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        res[i] = -1;
        ent[i] = -1;
        if (lab[i] <= 0 || p0 + i >= n_px) continue;
        const int job = ca.job_base[c] + lab[i] - 1;
        uint32_t rk;
        if (rankmap) rk = (uint32_t)rankmap[(size_t)c * n_px + p0 + i];        // many-segment frames: the rank was stored by the sort
        else {
          uint32_t k = key[i];
          if (k == 0u && fix_key) k = fix_key[job];    // recoloured in-mask black (0 = keep black)
          rk = rank_of(word_prefix, job, k);
        }
        res[i] = (int32_t)rk;
        if (first_pos) {
          // fp_lut (optional) maps (job, rank) to an entry of a smaller table, e.g. the level-1 clustered
          // palette: the table then stays cache resident and almost every pixel stops at the plain compare
          const int64_t e = pal_off[job] + rk;
          const int64_t te = fp_lut ? (int64_t)fp_lut[e] : e;
          ent[i] = (int32_t)te;                          // the entry of the (clustered) table this pixel shows: kept for the final remap
          int32_t* fp = first_pos + te;
          const int32_t p = (int32_t)(p0 + i);
          if (p < *fp) atomicMin(fp, p);
        }
      }
      if (entry_out) {
        int32_t* o = entry_out + (size_t)c * n_px + p0;
        if (p0 + 4 <= n_px) {
          *reinterpret_cast<int4*>(o) = make_int4(ent[0], ent[1], ent[2], ent[3]);
        } else {
          for (int i = 0; i < 4 && p0 + i < n_px; ++i) o[i] = ent[i];
        }
      }
      if (idx_out) {
        int32_t* o = idx_out + (size_t)c * n_px + p0;
        if (p0 + 4 <= n_px) {
          *reinterpret_cast<int4*>(o) = make_int4(res[0], res[1], res[2], res[3]);
        } else {
          for (int i = 0; i < 4 && p0 + i < n_px; ++i) o[i] = res[i];
        }
      }
    }
  }
}

template <typename OutT>
__global__ __launch_bounds__(256) void frame_remap_kernel(const uint8_t* __restrict__ rgb, int H, int W, ClassArgs ca,
                                                          const uint32_t* __restrict__ bitmaps, const uint32_t* __restrict__ word_prefix,
                                                          const int64_t* __restrict__ pal_off, const uint32_t* __restrict__ fix_key,
                                                          const int32_t* __restrict__ lut, const int32_t* __restrict__ lut2,
                                                          int32_t default_index, OutT* __restrict__ out, const int32_t* __restrict__ rankmap) {
  const int64_t n_px = (int64_t)H * W;
  const int64_t n_quads = (n_px + 3) >> 2;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_quads; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p0 = q << 2;
    uint32_t key[4] = {0u, 0u, 0u, 0u};
    if (rgb) load4px(rgb, p0, n_px, key);                  // (ranked path: no pixel colour is read at all -- rgb == nullptr, uniform branch)
    int32_t res[4] = {-1, -1, -1, -1};
    for (int c = 0; c < ca.n_class; ++c) {
      if (res[0] >= 0 && res[1] >= 0 && res[2] >= 0 && res[3] >= 0) break;
      int32_t lab[4];
      load4lab(ca.labels[c], p0, n_px, lab);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (res[i] >= 0 || lab[i] <= 0 || p0 + i >= n_px) continue;
        const int job = ca.job_base[c] + lab[i] - 1;
        uint32_t rk;
        if (rankmap) rk = (uint32_t)rankmap[(size_t)c * n_px + p0 + i];
        else {
          uint32_t k = key[i];
          if (k == 0u && fix_key) k = fix_key[job];
          rk = rank_of(word_prefix, job, k);
        }
        int32_t v = lut[pal_off[job] + rk];
        if (lut2) v = lut2[v];                          // level-1 index -> composed levels 2/3 (small, cache resident)
        res[i] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (res[i] < 0) res[i] = default_index;
    if (p0 + 4 <= n_px) {
      if constexpr (sizeof(OutT) == 1) {
        *reinterpret_cast<uint32_t*>(out + p0) = (uint32_t)res[0] | ((uint32_t)res[1] << 8) | ((uint32_t)res[2] << 16) | ((uint32_t)res[3] << 24);
      } else if constexpr (sizeof(OutT) == 2) {
        *reinterpret_cast<uint2*>(out + p0) = make_uint2((uint32_t)res[0] | ((uint32_t)res[1] << 16), (uint32_t)res[2] | ((uint32_t)res[3] << 16));
      } else {
        *reinterpret_cast<int4*>(out + p0) = make_int4(res[0], res[1], res[2], res[3]);
      }
    } else {
      for (int i = 0; i < 4 && p0 + i < n_px; ++i) out[p0 + i] = (OutT)res[i];
    }
  }
}

// K6 from stored table entries: the first-position pass of a class (job_index_kernel with entry_out) already looked every pixel's
// colour up -- rank in its job's palette, then the level-1 clustered entry -- so the final remap needs neither the pixels nor the
// 16 MB rank tables nor the level-1 look-up table again: labels + 4 bytes per (class, pixel) streamed, one gather into the small
// composed table of levels 2/3.  (frame_remap_kernel fetched 681 MB for 91 MB of pixels and labels: two random gathers per pixel.)
template <typename OutT>
__global__ __launch_bounds__(256) void frame_remap_entries_kernel(int H, int W, ClassArgs ca, const int32_t* __restrict__ entries,
                                                                  const int32_t* __restrict__ lut2, int32_t default_index, OutT* __restrict__ out) {
  const int64_t n_px = (int64_t)H * W;
  const int64_t n_quads = (n_px + 3) >> 2;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_quads; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p0 = q << 2;
    int32_t res[4] = {-1, -1, -1, -1};
    for (int c = 0; c < ca.n_class; ++c) {
      if (res[0] >= 0 && res[1] >= 0 && res[2] >= 0 && res[3] >= 0) break;
      int32_t lab[4], ent[4];
      load4lab(ca.labels[c], p0, n_px, lab);
      load4lab(entries + (size_t)c * n_px, p0, n_px, ent);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (res[i] >= 0 || lab[i] <= 0 || p0 + i >= n_px) continue;
        res[i] = lut2 ? lut2[ent[i]] : ent[i];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (res[i] < 0) res[i] = default_index;
    if (p0 + 4 <= n_px) {
      if constexpr (sizeof(OutT) == 1) {
        *reinterpret_cast<uint32_t*>(out + p0) = (uint32_t)res[0] | ((uint32_t)res[1] << 8) | ((uint32_t)res[2] << 16) | ((uint32_t)res[3] << 24);
      } else if constexpr (sizeof(OutT) == 2) {
        *reinterpret_cast<uint2*>(out + p0) = make_uint2((uint32_t)res[0] | ((uint32_t)res[1] << 16), (uint32_t)res[2] | ((uint32_t)res[3] << 16));
      } else {
        *reinterpret_cast<int4*>(out + p0) = make_int4(res[0], res[1], res[2], res[3]);
      }
    } else {
      for (int i = 0; i < 4 && p0 + i < n_px; ++i) out[p0 + i] = (OutT)res[i];
    }
  }
}

__global__ __launch_bounds__(256) void remap_kernel(const int32_t* __restrict__ idx, int64_t n, const int32_t* __restrict__ lut,
                                                    int64_t lut_n, int32_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t v = idx[i];
    out[i] = (v >= 0 && v < lut_n) ? lut[v] : 0;
  }
}

template <typename IdxT>
__global__ __launch_bounds__(256) void decode_kernel(const IdxT* __restrict__ idx, int64_t n, const uint8_t* __restrict__ pal,
                                                     int64_t pal_n, uint8_t* __restrict__ rgb) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t v = (int64_t)idx[i];
    if (v >= pal_n) v = 0;
    rgb[i * 3] = pal[v * 3]; rgb[i * 3 + 1] = pal[v * 3 + 1]; rgb[i * 3 + 2] = pal[v * 3 + 2];
  }
}

static int make_class_args(rhccq_ctx* ctx, int n_class, const int32_t* const* labels, const int32_t* job_base, ClassArgs* ca) {
  if (n_class < 1 || n_class > kMaxClass) return rhccq_fail(ctx, RHCCQ_E_ARG, "n_class must be 1..4");
  ca->n_class = n_class;
  for (int i = 0; i < kMaxClass; ++i) {
    ca->labels[i] = i < n_class ? labels[i] : nullptr;
    ca->job_base[i] = i < n_class ? job_base[i] : 0;
  }
  return 0;
}

static inline int stream_grid(int64_t items, int per_block) {
  int64_t b = (items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > 256 * 8) b = 256 * 8;                         // 8 blocks per CU, grid-stride beyond
  return (int)b;
}

}  // namespace rhccq

using namespace rhccq;

extern "C" {

int rhccq_job_scan(rhccq_ctx* ctx, const uint8_t* rgb, int32_t H, int32_t W, int32_t n_class,
                   const int32_t* const* labels_host, const int32_t* job_base_host, int32_t black_is_colour,
                   uint32_t* bitmaps, int32_t* stats) {
  if (!ctx || !rgb || !bitmaps || !stats || H <= 0 || W <= 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "job_scan: bad argument");
  if (((uintptr_t)rgb & 3u) != 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "job_scan: rgb must be 4-byte aligned");
  ClassArgs ca;
  if (int e = make_class_args(ctx, n_class, labels_host, job_base_host, &ca)) return e;
  const int64_t quads = ((int64_t)H * W + 3) / 4;
  // contiguous runs per block keep the per-block job hash small; cap the grid at 8 blocks per CU
  int64_t g64 = (quads + 1023) / 1024;
  if (g64 > 2048) g64 = 2048;
  if (g64 < 1) g64 = 1;
  const int grid = (int)g64;
  hipLaunchKernelGGL(job_scan_kernel<false>, dim3(grid), dim3(256), 0, ctx->stream, rgb, H, W, ca, black_is_colour, bitmaps, (uint8_t*)nullptr, stats);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_job_scan_bytes(rhccq_ctx* ctx, const uint8_t* rgb, int32_t H, int32_t W, int32_t n_class,
                         const int32_t* const* labels_host, const int32_t* job_base_host, int32_t black_is_colour,
                         uint8_t* bytemaps, int32_t* stats) {
  if (!ctx || !rgb || !bytemaps || !stats || H <= 0 || W <= 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "job_scan_bytes: bad argument");
  if (((uintptr_t)rgb & 3u) != 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "job_scan_bytes: rgb must be 4-byte aligned");
  ClassArgs ca;
  if (int e = make_class_args(ctx, n_class, labels_host, job_base_host, &ca)) return e;
  const int64_t quads = ((int64_t)H * W + 3) / 4;
  int64_t g64 = (quads + 1023) / 1024;
  if (g64 > 2048) g64 = 2048;
  if (g64 < 1) g64 = 1;
  hipLaunchKernelGGL(job_scan_kernel<true>, dim3((int)g64), dim3(256), 0, ctx->stream, rgb, H, W, ca, black_is_colour, (uint32_t*)nullptr, bytemaps, stats);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_bytemap_pack(rhccq_ctx* ctx, const uint8_t* bytemaps, int32_t n_jobs, uint32_t* bitmaps) {
  if (!ctx || !bytemaps || !bitmaps || n_jobs <= 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "bytemap_pack: bad argument");
  if (((uintptr_t)bytemaps & 15u) != 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "bytemap_pack: bytemaps must be 16-byte aligned");
  const size_t n_words = (size_t)n_jobs * RHCCQ_BITMAP_WORDS;
  hipLaunchKernelGGL(bytemap_pack_kernel, dim3(2048), dim3(256), 0, ctx->stream, bytemaps, bitmaps, n_words);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_job_set_black(rhccq_ctx* ctx, uint32_t* bitmaps, const int32_t* jobs, int32_t n_jobs) {
  if (!ctx || !bitmaps || (n_jobs > 0 && !jobs)) return rhccq_fail(ctx, RHCCQ_E_ARG, "job_set_black: bad argument");
  if (n_jobs <= 0) return 0;
  hipLaunchKernelGGL(job_set_black_kernel, dim3((n_jobs + 63) / 64), dim3(64), 0, ctx->stream, bitmaps, jobs, n_jobs);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_bitmap_count(rhccq_ctx* ctx, const uint32_t* bitmaps, int32_t n_jobs, uint32_t* chunk_sums, int32_t* counts) {
  if (!ctx || !bitmaps || !chunk_sums || !counts || n_jobs <= 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "bitmap_count: bad argument");
  if (n_jobs > 65535) return rhccq_fail(ctx, RHCCQ_E_LIMIT, "bitmap_count: more than 65535 jobs per call");
  hipLaunchKernelGGL(bitmap_chunk_count_kernel, dim3(kChunks, n_jobs), dim3(256), 0, ctx->stream, bitmaps, chunk_sums);
  hipLaunchKernelGGL(bitmap_chunk_scan_kernel, dim3(n_jobs), dim3(512), 0, ctx->stream, chunk_sums, counts);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_bitmap_emit(rhccq_ctx* ctx, const uint32_t* bitmaps, int32_t n_jobs, const uint32_t* chunk_sums,
                      const int64_t* pal_off, uint32_t* word_prefix, uint32_t* keys_out) {
  if (!ctx || !bitmaps || !chunk_sums || !word_prefix || n_jobs <= 0 || (keys_out && !pal_off))
    return rhccq_fail(ctx, RHCCQ_E_ARG, "bitmap_emit: bad argument");
  if (n_jobs > 65535) return rhccq_fail(ctx, RHCCQ_E_LIMIT, "bitmap_emit: more than 65535 jobs per call");
  hipLaunchKernelGGL(bitmap_emit_kernel, dim3(kChunks, n_jobs), dim3(256), 0, ctx->stream, bitmaps, chunk_sums, pal_off, word_prefix, keys_out);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_job_blackfix(rhccq_ctx* ctx, const uint8_t* rgb, int32_t H, int32_t W, int32_t n_class,
                       const int32_t* const* labels_host, const int32_t* job_base_host, const uint8_t* job_needs_fix,
                       unsigned long long* best) {
  if (!ctx || !rgb || !job_needs_fix || !best || H <= 0 || W <= 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "job_blackfix: bad argument");
  ClassArgs ca;
  if (int e = make_class_args(ctx, n_class, labels_host, job_base_host, &ca)) return e;
  const int64_t quads = ((int64_t)H * W + 3) / 4;
  hipLaunchKernelGGL(job_blackfix_kernel, dim3(stream_grid(quads, 256)), dim3(256), 0, ctx->stream, rgb, H, W, ca, job_needs_fix, best);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_job_index(rhccq_ctx* ctx, const uint8_t* rgb, int32_t H, int32_t W, int32_t n_class,
                    const int32_t* const* labels_host, const int32_t* job_base_host, const uint32_t* bitmaps,
                    const uint32_t* word_prefix, const int64_t* pal_off, const uint32_t* fix_key, int32_t* idx_out,
                    int32_t* first_pos, const int32_t* fp_lut) {
  if (!ctx || !rgb || !bitmaps || !word_prefix || !pal_off || H <= 0 || W <= 0 || (int64_t)H * W > INT32_MAX)
    return rhccq_fail(ctx, RHCCQ_E_ARG, "job_index: bad argument");
  ClassArgs ca;
  if (int e = make_class_args(ctx, n_class, labels_host, job_base_host, &ca)) return e;
  const int64_t quads = ((int64_t)H * W + 3) / 4;
  hipLaunchKernelGGL(job_index_kernel, dim3(stream_grid(quads, 256)), dim3(256), 0, ctx->stream, rgb, H, W, ca, bitmaps, word_prefix,
                     pal_off, fix_key, idx_out, first_pos, fp_lut, (const int32_t*)nullptr, (int32_t*)nullptr);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_job_index_entries(rhccq_ctx* ctx, const uint8_t* rgb, int32_t H, int32_t W, int32_t n_class, const int32_t* const* labels_host,
                            const int32_t* job_base_host, const uint32_t* bitmaps, const uint32_t* word_prefix, const int64_t* pal_off,
                            const uint32_t* fix_key, int32_t* first_pos, const int32_t* fp_lut, int32_t* entries_out) {
  if (!ctx || !rgb || !bitmaps || !word_prefix || !pal_off || !first_pos || !entries_out || H <= 0 || W <= 0 || (int64_t)H * W > INT32_MAX)
    return rhccq_fail(ctx, RHCCQ_E_ARG, "job_index_entries: bad argument");
  ClassArgs ca;
  if (int e = make_class_args(ctx, n_class, labels_host, job_base_host, &ca)) return e;
  const int64_t quads = ((int64_t)H * W + 3) / 4;
  hipLaunchKernelGGL(job_index_kernel, dim3(stream_grid(quads, 256)), dim3(256), 0, ctx->stream, rgb, H, W, ca, bitmaps, word_prefix,
                     pal_off, fix_key, (int32_t*)nullptr, first_pos, fp_lut, (const int32_t*)nullptr, entries_out);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_frame_remap_entries(rhccq_ctx* ctx, int32_t H, int32_t W, int32_t n_class, const int32_t* const* labels_host,
                              const int32_t* job_base_host, const int32_t* entries, const int32_t* lut2, int32_t default_index, void* out,
                              int32_t out_elem_bytes) {
  if (!ctx || !entries || !out || H <= 0 || W <= 0 || (int64_t)H * W > INT32_MAX)
    return rhccq_fail(ctx, RHCCQ_E_ARG, "frame_remap_entries: bad argument");
  ClassArgs ca;
  if (int e = make_class_args(ctx, n_class, labels_host, job_base_host, &ca)) return e;
  const int64_t quads = ((int64_t)H * W + 3) / 4;
  const dim3 grid(stream_grid(quads, 256));
  if (out_elem_bytes == 1)
    hipLaunchKernelGGL(frame_remap_entries_kernel<uint8_t>, grid, dim3(256), 0, ctx->stream, H, W, ca, entries, lut2, default_index, (uint8_t*)out);
  else if (out_elem_bytes == 2)
    hipLaunchKernelGGL(frame_remap_entries_kernel<uint16_t>, grid, dim3(256), 0, ctx->stream, H, W, ca, entries, lut2, default_index, (uint16_t*)out);
  else if (out_elem_bytes == 4)
    hipLaunchKernelGGL(frame_remap_entries_kernel<int32_t>, grid, dim3(256), 0, ctx->stream, H, W, ca, entries, lut2, default_index, (int32_t*)out);
  else
    return rhccq_fail(ctx, RHCCQ_E_ARG, "frame_remap_entries: out_elem_bytes must be 1, 2 or 4");
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_job_index_ranked(rhccq_ctx* ctx, int32_t H, int32_t W, int32_t n_class, const int32_t* const* labels_host, const int32_t* job_base_host,
                           const int32_t* rankmap, const int64_t* pal_off, int32_t* first_pos, const int32_t* fp_lut) {
  if (!ctx || !rankmap || !pal_off || !first_pos || H <= 0 || W <= 0 || (int64_t)H * W > INT32_MAX)
    return rhccq_fail(ctx, RHCCQ_E_ARG, "job_index_ranked: bad argument");
  ClassArgs ca;
  if (int e = make_class_args(ctx, n_class, labels_host, job_base_host, &ca)) return e;
  const int64_t quads = ((int64_t)H * W + 3) / 4;
  // (the kernel reads no pixel colour on this path: rgb == nullptr skips the pixel load)
  hipLaunchKernelGGL(job_index_kernel, dim3(stream_grid(quads, 256)), dim3(256), 0, ctx->stream, (const uint8_t*)nullptr, H, W, ca, (const uint32_t*)nullptr,
                     (const uint32_t*)nullptr, pal_off, (const uint32_t*)nullptr, (int32_t*)nullptr, first_pos, fp_lut, rankmap, (int32_t*)nullptr);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_job_stats(rhccq_ctx* ctx, const uint8_t* rgb, int32_t H, int32_t W, int32_t n_class, const int32_t* const* labels_host,
                    const int32_t* job_base_host, int32_t* stats) {
  if (!ctx || !rgb || !stats || H <= 0 || W <= 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "job_stats: bad argument");
  if (((uintptr_t)rgb & 3u) != 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "job_stats: rgb must be 4-byte aligned");
  ClassArgs ca;
  if (int e = make_class_args(ctx, n_class, labels_host, job_base_host, &ca)) return e;
  const int64_t quads = ((int64_t)H * W + 3) / 4;
  int64_t g64 = (quads + 1023) / 1024;
  if (g64 > 2048) g64 = 2048;
  if (g64 < 1) g64 = 1;
  hipLaunchKernelGGL(job_scan_kernel<false>, dim3((int)g64), dim3(256), 0, ctx->stream, rgb, H, W, ca, 0, (uint32_t*)nullptr, (uint8_t*)nullptr, stats);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

static inline size_t k1_align(size_t v) { return (v + 255) & ~(size_t)255; }

int64_t rhccq_job_sort_unique_bytes(int64_t n_entries) {
  if (n_entries <= 0) return 0;
  // keys in / out (u64), values in / out (u32), heads + their scan (u32), hipCUB's own scratch (bounded by ~ the key array)
  return (int64_t)(2 * k1_align(8 * (size_t)n_entries) + 4 * k1_align(4 * (size_t)n_entries) + k1_align(8 * (size_t)n_entries) + (16u << 20));
}

int rhccq_job_sort_unique(rhccq_ctx* ctx, const uint8_t* rgb, int32_t H, int32_t W, int32_t n_class, const int32_t* const* labels_host,
                          const int32_t* job_base_host, int32_t n_jobs, const uint32_t* fix_key, const int32_t* black_jobs, int32_t n_black,
                          void* tmp, int64_t tmp_bytes, int32_t* rankmap, uint32_t* keys_out, int32_t* job_start, int32_t* n_unique) {
  if (!ctx || !rgb || !tmp || !rankmap || !keys_out || !job_start || !n_unique || H <= 0 || W <= 0 || n_jobs <= 0 || n_black < 0 || (n_black > 0 && !black_jobs))
    return rhccq_fail(ctx, RHCCQ_E_ARG, "job_sort_unique: bad argument");
  ClassArgs ca;
  if (int e = make_class_args(ctx, n_class, labels_host, job_base_host, &ca)) return e;
  const int64_t n_px = (int64_t)H * W, n_pix_entries = (int64_t)n_class * n_px, total = n_pix_entries + n_black;
  if (total >= (1ll << 31)) return rhccq_fail(ctx, RHCCQ_E_LIMIT, "job_sort_unique: more than 2^31 (pixel, class) entries");
  if (tmp_bytes < rhccq_job_sort_unique_bytes(total)) return rhccq_fail(ctx, RHCCQ_E_ARG, "job_sort_unique: tmp too small");
  int jb = 1;
  while ((1ll << jb) < n_jobs) ++jb;
  const int invalid_shift = 24 + jb;                      // valid keys are < 2^(24 + jb)
  char* base = (char*)tmp;
  unsigned long long* k0 = (unsigned long long*)base; base += k1_align(8 * (size_t)total);
  unsigned long long* k1 = (unsigned long long*)base; base += k1_align(8 * (size_t)total);
  uint32_t* v0 = (uint32_t*)base; base += k1_align(4 * (size_t)total);
  uint32_t* v1 = (uint32_t*)base; base += k1_align(4 * (size_t)total);
  uint32_t* heads = (uint32_t*)base; base += k1_align(4 * (size_t)total);
  uint32_t* uidx = (uint32_t*)base; base += k1_align(4 * (size_t)total);
  void* cub_tmp = (void*)base;
  size_t cub_avail = (size_t)tmp_bytes - (size_t)(base - (char*)tmp);
  const unsigned grid = (unsigned)((total + 255) / 256);
  hipLaunchKernelGGL(sort_keys_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, ctx->stream, rgb, H, W, ca, fix_key, black_jobs, n_black, invalid_shift, k0, v0);
  hipcub::DoubleBuffer<unsigned long long> kb(k0, k1);
  hipcub::DoubleBuffer<uint32_t> vb(v0, v1);
  size_t need = 0;
  RHCCQ_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, need, kb, vb, (int)total, 0, invalid_shift + 1, ctx->stream));
  if (need > cub_avail) return rhccq_fail(ctx, RHCCQ_E_LIMIT, "job_sort_unique: sort scratch exceeds tmp");
  RHCCQ_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(cub_tmp, need, kb, vb, (int)total, 0, invalid_shift + 1, ctx->stream));
  const unsigned long long* ks = kb.Current();
  const uint32_t* vs = vb.Current();
  hipLaunchKernelGGL(sort_heads_kernel, dim3(grid), dim3(256), 0, ctx->stream, ks, total, invalid_shift, heads);
  size_t need2 = 0;
  RHCCQ_HIP(ctx, hipcub::DeviceScan::ExclusiveSum(nullptr, need2, heads, uidx, (int)total, ctx->stream));
  if (need2 > cub_avail) return rhccq_fail(ctx, RHCCQ_E_LIMIT, "job_sort_unique: scan scratch exceeds tmp");
  RHCCQ_HIP(ctx, hipcub::DeviceScan::ExclusiveSum(cub_tmp, need2, heads, uidx, (int)total, ctx->stream));
  RHCCQ_HIP(ctx, hipMemsetAsync(job_start, 0xff, sizeof(int32_t) * (size_t)n_jobs, ctx->stream));       // -1: the job has no colour
  hipLaunchKernelGGL(sort_emit_kernel, dim3(grid), dim3(256), 0, ctx->stream, ks, heads, uidx, total, invalid_shift, job_start, keys_out, n_unique);
  hipLaunchKernelGGL(sort_ranks_kernel, dim3(grid), dim3(256), 0, ctx->stream, ks, vs, heads, uidx, total, n_pix_entries, invalid_shift, job_start, rankmap);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

static int frame_remap_impl(rhccq_ctx* ctx, const uint8_t* rgb, int32_t H, int32_t W, int32_t n_class,
                            const int32_t* const* labels_host, const int32_t* job_base_host, const uint32_t* bitmaps,
                            const uint32_t* word_prefix, const int64_t* pal_off, const uint32_t* fix_key, const int32_t* lut,
                            const int32_t* lut2, int32_t default_index, void* out, int32_t out_elem_bytes, const int32_t* rankmap);

int rhccq_frame_remap(rhccq_ctx* ctx, const uint8_t* rgb, int32_t H, int32_t W, int32_t n_class,
                      const int32_t* const* labels_host, const int32_t* job_base_host, const uint32_t* bitmaps,
                      const uint32_t* word_prefix, const int64_t* pal_off, const uint32_t* fix_key, const int32_t* lut,
                      const int32_t* lut2, int32_t default_index, void* out, int32_t out_elem_bytes) {
  if (!ctx || !rgb || !bitmaps || !word_prefix || !pal_off || !lut || !out || H <= 0 || W <= 0)
    return rhccq_fail(ctx, RHCCQ_E_ARG, "frame_remap: bad argument");
  return frame_remap_impl(ctx, rgb, H, W, n_class, labels_host, job_base_host, bitmaps, word_prefix, pal_off, fix_key, lut, lut2, default_index, out,
                          out_elem_bytes, nullptr);
}

int rhccq_frame_remap_ranked(rhccq_ctx* ctx, int32_t H, int32_t W, int32_t n_class, const int32_t* const* labels_host, const int32_t* job_base_host,
                             const int32_t* rankmap, const int64_t* pal_off, const int32_t* lut, const int32_t* lut2, int32_t default_index, void* out,
                             int32_t out_elem_bytes) {
  if (!ctx || !rankmap || !pal_off || !lut || !out || H <= 0 || W <= 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "frame_remap_ranked: bad argument");
  return frame_remap_impl(ctx, (const uint8_t*)nullptr, H, W, n_class, labels_host, job_base_host, nullptr, nullptr, pal_off, nullptr, lut, lut2,
                          default_index, out, out_elem_bytes, rankmap);
}

static int frame_remap_impl(rhccq_ctx* ctx, const uint8_t* rgb, int32_t H, int32_t W, int32_t n_class,
                            const int32_t* const* labels_host, const int32_t* job_base_host, const uint32_t* bitmaps,
                            const uint32_t* word_prefix, const int64_t* pal_off, const uint32_t* fix_key, const int32_t* lut,
                            const int32_t* lut2, int32_t default_index, void* out, int32_t out_elem_bytes, const int32_t* rankmap) {
  ClassArgs ca;
  if (int e = make_class_args(ctx, n_class, labels_host, job_base_host, &ca)) return e;
  const int64_t quads = ((int64_t)H * W + 3) / 4;
  const dim3 grid(stream_grid(quads, 256)), block(256);
  switch (out_elem_bytes) {
    case 1: hipLaunchKernelGGL(frame_remap_kernel<uint8_t>, grid, block, 0, ctx->stream, rgb, H, W, ca, bitmaps, word_prefix, pal_off, fix_key, lut, lut2, default_index, (uint8_t*)out, rankmap); break;
    case 2: hipLaunchKernelGGL(frame_remap_kernel<uint16_t>, grid, block, 0, ctx->stream, rgb, H, W, ca, bitmaps, word_prefix, pal_off, fix_key, lut, lut2, default_index, (uint16_t*)out, rankmap); break;
    case 4: hipLaunchKernelGGL(frame_remap_kernel<int32_t>, grid, block, 0, ctx->stream, rgb, H, W, ca, bitmaps, word_prefix, pal_off, fix_key, lut, lut2, default_index, (int32_t*)out, rankmap); break;
    default: return rhccq_fail(ctx, RHCCQ_E_ARG, "frame_remap: out_elem_bytes must be 1, 2 or 4");
  }
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_remap(rhccq_ctx* ctx, const int32_t* idx, int64_t n, const int32_t* lut, int64_t lut_n, int32_t* out) {
  if (!ctx || n < 0 || (n > 0 && (!idx || !lut || !out))) return rhccq_fail(ctx, RHCCQ_E_ARG, "remap: bad argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(remap_kernel, dim3(stream_grid(n, 1024)), dim3(256), 0, ctx->stream, idx, n, lut, lut_n, out);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_decode(rhccq_ctx* ctx, const void* idx, int32_t idx_elem_bytes, int64_t n, const uint8_t* palette, int64_t pal_n, uint8_t* rgb_out) {
  if (!ctx || n < 0 || (n > 0 && (!idx || !palette || !rgb_out)) || pal_n <= 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "decode: bad argument");
  if (n == 0) return 0;
  const dim3 grid(stream_grid(n, 1024)), block(256);
  switch (idx_elem_bytes) {
    case 1: hipLaunchKernelGGL(decode_kernel<uint8_t>, grid, block, 0, ctx->stream, (const uint8_t*)idx, n, palette, pal_n, rgb_out); break;
    case 2: hipLaunchKernelGGL(decode_kernel<uint16_t>, grid, block, 0, ctx->stream, (const uint16_t*)idx, n, palette, pal_n, rgb_out); break;
    case 4: hipLaunchKernelGGL(decode_kernel<uint32_t>, grid, block, 0, ctx->stream, (const uint32_t*)idx, n, palette, pal_n, rgb_out); break;
    default: return rhccq_fail(ctx, RHCCQ_E_ARG, "decode: idx_elem_bytes must be 1, 2 or 4");
  }
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

}  // extern "C"
