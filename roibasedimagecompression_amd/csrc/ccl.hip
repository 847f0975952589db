// Binary-mask image operators of the ROI stage (reference encoder/ROI/roi.py): connected components with
// statistics (what the reference gets from cv2.connectedComponentsWithStats, roi.py:285-360 and every clean-up step of
// encoder/ROI/*) and the buffer-zone split of extract_roi_nonroi (roi.py:685-718).
//
// Layout: one wave per (row, 64-pixel segment).  The foreground of a segment is one 64-bit ballot, kept as a bit plane
// (8 294 400 px -> 1 MB) so that every neighbourhood test of the later passes is a few bit operations on at most six words.
//   ccl_init:    ballot -> bit plane; parent[p] = first pixel of p's horizontal run inside the segment (no memory traffic between
//                lanes: the run start is the position above the highest 0 bit below the lane).
//   ccl_merge:   lock-free union-find; a pixel links to the row above / the previous segment only where its left neighbour
//                cannot have done so already (run starts and diagonal-only contacts), so a solid area issues no union at all.
//   ccl_flatten: parent[p] = root (one tree walk per run, shared through the wave); roots take a compact id.
//   ccl_stats:   area / bounding box / first 2x2 block per root; every value comes from the ballot of the lanes that share
//                a root (popcount, ctz, clz), gathered per workgroup in an LDS table, then a handful of global atomics.
//   ccl_mark / scan / rank: the numbering.  Ordering keys are unique small integers, so label = 1 + number of smaller keys
//                = a prefix count over the key bit plane; the statistics land in label order, in cv2's column order.
//   ccl_relabel: labels[p] = rank[id[root]].
// One C call, no host round trip inside.  HBM bytes per pixel (algorithmic): 1 B mask read + 4 B parent written (init),
// 4 + 4 (flatten), 4 (stats), 4 + 4 (relabel) = 25 B/px; merge reads the 1-bit plane only.
#include "rhccq_common.h"

namespace rhccq {

constexpr int kCclStat = 6;   // area, min x, max x, min y, max y, ordering key

// walk to the root; kHalve: path halving on the way -- a node that is not a root is re-pointed at its grandparent (still an
// ancestor; only roots are ever compare-and-swapped, and a node that stopped being a root never becomes one again, so the plain
// re-pointing cannot undo a link).  The flatten pass walks WITHOUT halving: a late re-pointing by another wave could otherwise
// overwrite the final parent[p] = root with an older ancestor.
template <bool kHalve>
__device__ __forceinline__ int ccl_find(int32_t* parent, int x) {
  int p = __hip_atomic_load(parent + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (p != x) {
    const int gp = __hip_atomic_load(parent + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (kHalve && gp != p) __hip_atomic_store(parent + x, gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    x = p;
    p = gp;
  }
  return x;
}
__device__ __forceinline__ void ccl_union(int32_t* parent, int a, int b) {
  while (true) {
    a = ccl_find<true>(parent, a);
    b = ccl_find<true>(parent, b);
    if (a == b) return;
    if (a > b) { const int s = a; a = b; b = s; }
    const int old = atomicCAS(&parent[b], b, a);          // the larger root goes under the smaller: a root is its component's first pixel
    if (old == b) return;
    b = old;
  }
}

// wave -> (row, segment); false when the wave has no work
__device__ __forceinline__ bool ccl_wave_pos(int H, int segs, int& y, int& seg, int& lane) {
  const long long w = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  lane = threadIdx.x & 63;
  if (w >= (long long)H * segs) return false;
  y = (int)(w / segs);
  seg = (int)(w % segs);
  return true;
}

__global__ __launch_bounds__(256) void ccl_init_kernel(const uint8_t* __restrict__ mask, int H, int W, int segs, uint64_t* __restrict__ bits,
                                                       int32_t* __restrict__ parent) {
  int y, seg, lane;
  if (!ccl_wave_pos(H, segs, y, seg, lane)) return;
  const int x = seg * 64 + lane;
  const bool fg = x < W && mask[(long long)y * W + x] != 0;
  const uint64_t b = __ballot(fg);
  if (lane == 0) bits[(long long)y * segs + seg] = b;
  if (x < W) {
    int v = -1;
    if (fg) {
      const uint64_t zeros_below = ~b & ((1ull << lane) - 1ull);
      const int start = zeros_below ? 64 - __clzll((long long)zeros_below) : 0;
      v = y * W + seg * 64 + start;
    }
    parent[(long long)y * W + x] = v;
  }
}

// bit x of row y of the plane (0 outside the image)
struct CclRow {
  uint64_t prev, cur, next;   // segments seg-1, seg, seg+1
  __device__ __forceinline__ bool at(int lane) const {      // lane in [-1, 64]
    if (lane < 0) return (prev >> 63) & 1ull;
    if (lane > 63) return next & 1ull;
    return (cur >> lane) & 1ull;
  }
};
__device__ __forceinline__ CclRow ccl_row(const uint64_t* __restrict__ bits, int y, int seg, int H, int segs) {
  CclRow r{0, 0, 0};
  if (y < 0 || y >= H) return r;
  const uint64_t* p = bits + (long long)y * segs;
  r.cur = p[seg];
  if (seg > 0) r.prev = p[seg - 1];
  if (seg + 1 < segs) r.next = p[seg + 1];
  return r;
}

template <int kConn>
__global__ __launch_bounds__(256) void ccl_merge_kernel(const uint64_t* __restrict__ bits, int H, int W, int segs, int32_t* parent) {
  int y, seg, lane;
  if (!ccl_wave_pos(H, segs, y, seg, lane)) return;
  const CclRow c = ccl_row(bits, y, seg, H, segs), u = ccl_row(bits, y - 1, seg, H, segs);
  if (!c.at(lane)) return;
  const int x = seg * 64 + lane;
  const int p = y * W + x;
  const bool L = c.at(lane - 1), U = u.at(lane), UL = u.at(lane - 1), UR = u.at(lane + 1);
  if (lane == 0 && L) ccl_union(parent, p, p - 1);          // runs are joined inside a segment by ccl_init, across segments here
  if (kConn == 4) {
    if (U && !(L && UL)) ccl_union(parent, p, p - W);       // L && UL: the left neighbour linked to UL, which shares U's run
  } else {
    if (U) {
      if (!L) ccl_union(parent, p, p - W);                  // L: the left neighbour reaches U through its own U or UR link
    } else {
      if (UL && !L) ccl_union(parent, p, p - W - 1);
      if (UR) ccl_union(parent, p, p - W + 1);
    }
  }
}

constexpr int kCclEmpty = -3;     // LDS slot marker (roots are >= 0, the background aggregates under -1)
constexpr int kCclSlots = 128;

__device__ __forceinline__ void ccl_stat_reset(int32_t* s, int key) {
  s[0] = 0; s[1] = 0x7fffffff; s[2] = -1; s[3] = 0x7fffffff; s[4] = -1; s[5] = key;
}

// parent[p] = root.  Only the first pixel of each horizontal run walks the tree (the other pixels of a run still point at it:
// they are never roots, so no union ever touched them); roots take a compact id and reset their statistics row.
// key_is_root (4-connectivity, or raster numbering asked for): the ordering key of a component is its first pixel = the root
__global__ __launch_bounds__(1024) void ccl_flatten_kernel(int32_t* parent, int H, int W, int segs, int32_t* __restrict__ cid, int cap,
                                                           int32_t* __restrict__ stats, int32_t* count, int key_is_root) {
  __shared__ int s_cnt[16], s_base;
  int y = 0, seg = 0, lane = 0;
  const bool valid = ccl_wave_pos(H, segs, y, seg, lane);
  const int wave = threadIdx.x >> 6;
  const int x = seg * 64 + lane;
  const int p = y * W + x;
  if (valid && p == 0 && stats) ccl_stat_reset(stats + (long long)cap * kCclStat, -1);   // row `cap` = background (label 0 of cv2's stats)
  const bool fg = valid && x < W && parent[p] >= 0;
  const uint64_t b = __ballot(fg);
  const bool is_start = fg && (lane == 0 || !((b >> (lane - 1)) & 1ull));
  int r = -1;
  if (is_start) r = ccl_find<false>(parent, p);
  const uint64_t zeros_below = ~b & ((1ull << lane) - 1ull);
  const int start = zeros_below ? 64 - __clzll((long long)zeros_below) : 0;
  const int rs = __shfl(r, start);
  if (fg && !is_start) r = rs;
  if (fg) parent[p] = r;
  const bool root = fg && r == p;
  // compact ids: one counter update per WORKGROUP (a noisy 4K edge map has a root in almost every wave: 130 000 atomics on one
  // word cost 0.65 ms)
  const uint64_t roots = __ballot(root);
  if (lane == 0) s_cnt[wave] = __popcll(roots);
  __syncthreads();
  if (threadIdx.x == 0) {
    int total = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) total += s_cnt[w];
    s_base = total ? atomicAdd(count, total) : 0;
  }
  __syncthreads();
  if (root) {
    int base = s_base;
    for (int w = 0; w < wave; ++w) base += s_cnt[w];
    const int id = base + __popcll(roots & ((1ull << lane) - 1ull));
    cid[p] = id;
    if (stats && id < cap) ccl_stat_reset(stats + (long long)id * kCclStat, key_is_root ? r : 0x7fffffff);
  }
}

__device__ __forceinline__ void ccl_stat_add(int32_t* s, int area, int x0, int x1, int y0, int y1, int key) {
  atomicAdd(&s[0], area);
  atomicMin(&s[1], x0);
  atomicMax(&s[2], x1);
  atomicMin(&s[3], y0);
  atomicMax(&s[4], y1);
  if (key != 0x7fffffff) atomicMin(&s[5], key);
}

// area / bounding box / first 2x2 block per root.  A workgroup walks `units` consecutive (row, segment) units; per wave every
// value comes from the ballot of the lanes that share a root (popcount, ctz, clz); they are gathered in an LDS table keyed by
// root and only its occupied slots reach the global statistics rows (a solid 4K area: 6 atomics per workgroup, not per wave).
__global__ __launch_bounds__(1024) void ccl_stats_kernel(const int32_t* __restrict__ parent, const int32_t* __restrict__ cid, int H, int W, int segs,
                                                         int cap, int32_t* stats, int key_is_root, int units) {
  __shared__ int s_key[kCclSlots];
  __shared__ int32_t s_val[kCclSlots][kCclStat];
  for (int i = threadIdx.x; i < kCclSlots; i += 1024) {
    s_key[i] = kCclEmpty;
    ccl_stat_reset(s_val[i], 0x7fffffff);
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long total = (long long)H * segs;
  const int w2 = (W + 1) >> 1;
  for (int it = wave; it < units; it += 16) {
    const long long u = (long long)blockIdx.x * units + it;
    if (u >= total) break;
    const int y = (int)(u / segs), seg = (int)(u % segs);
    const int x = seg * 64 + lane;
    const int r = x < W ? parent[(long long)y * W + x] : -2;  // -1 background, -2 outside
    uint64_t todo = __ballot(r != -2);
    while (todo) {
      const int first = __builtin_ctzll(todo);
      const int lead = __shfl(r, first);
      const uint64_t same = __ballot(r == lead);
      todo &= ~same;
      if (lane == first) {
        const int x0 = seg * 64 + __builtin_ctzll(same), x1 = seg * 64 + 63 - __clzll((long long)same);
        // first 2x2 block in block-raster order: only pixels of the root's row pair can hold it
        const int key = (!key_is_root && lead >= 0 && (y >> 1) == ((lead / W) >> 1)) ? (y >> 1) * w2 + (x0 >> 1) : 0x7fffffff;
        int slot = (int)(((unsigned)lead * 2654435761u) >> 25);           // 7 bits
        bool done = false;
        for (int probe = 0; probe < 8 && !done; ++probe, slot = (slot + 1) & (kCclSlots - 1)) {
          const int old = atomicCAS(&s_key[slot], kCclEmpty, lead);
          if (old == kCclEmpty || old == lead) {
            ccl_stat_add(s_val[slot], __popcll(same), x0, x1, y, y, key);
            done = true;
          }
        }
        if (!done) {                                                       // table crowded: straight to the global row
          const int id = lead < 0 ? cap : cid[lead];
          if (lead < 0 || id < cap) ccl_stat_add(stats + (long long)id * kCclStat, __popcll(same), x0, x1, y, y, key);
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kCclSlots; i += 1024) {
    const int lead = s_key[i];
    if (lead == kCclEmpty) continue;
    const int id = lead < 0 ? cap : cid[lead];
    if (lead < 0 || id < cap) ccl_stat_add(stats + (long long)id * kCclStat, s_val[i][0], s_val[i][1], s_val[i][2], s_val[i][3], s_val[i][4], s_val[i][5]);
  }
}

// ---- numbering on the device: the ordering keys are unique integers below n_keys, so a component's label is 1 + the number
// of smaller keys = a prefix count over the key bit plane (mark -> per-word popcounts scanned in two levels -> rank)
__global__ __launch_bounds__(256) void ccl_mark_kernel(const int32_t* __restrict__ stats, const int32_t* __restrict__ count, int cap,
                                                       unsigned long long* keybits) {
  const int id = blockIdx.x * 256 + threadIdx.x;
  const int n = min(*count, cap);
  if (id >= n) return;
  const int key = stats[(long long)id * kCclStat + 5];
  atomicOr(&keybits[key >> 6], 1ull << (key & 63));
}

__device__ __forceinline__ int ccl_block_exclusive_scan(int v, int* s_wave /* [17] */, int& block_total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(inc, d);
    if (lane >= d) inc += t;
  }
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    int acc = 0;
    for (int w = 0; w < 16; ++w) { const int t = s_wave[w]; s_wave[w] = acc; acc += t; }
    s_wave[16] = acc;
  }
  __syncthreads();
  block_total = s_wave[16];
  return s_wave[wave] + inc - v;
}

__global__ __launch_bounds__(1024) void ccl_scan_words_kernel(const unsigned long long* __restrict__ keybits, long long n_words, int32_t* __restrict__ wordrank,
                                                              int32_t* __restrict__ blocksum) {
  __shared__ int s_wave[17];
  const long long w = (long long)blockIdx.x * 1024 + threadIdx.x;
  const int v = w < n_words ? __popcll(keybits[w]) : 0;
  int total;
  const int ex = ccl_block_exclusive_scan(v, s_wave, total);
  if (w < n_words) wordrank[w] = ex;
  if (threadIdx.x == 0) blocksum[blockIdx.x] = total;
}

__global__ __launch_bounds__(1024) void ccl_scan_blocks_kernel(int32_t* blocksum, int n_blocks) {
  __shared__ int s_wave[17];
  int carry = 0;
  for (int base = 0; base < n_blocks; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < n_blocks ? blocksum[i] : 0;
    int total;
    const int ex = ccl_block_exclusive_scan(v, s_wave, total);
    if (i < n_blocks) blocksum[i] = carry + ex;
    carry += total;
    __syncthreads();
  }
}

// rank[id] = label; the component's row of the cv2-ordered statistics (LEFT, TOP, WIDTH, HEIGHT, AREA); row 0 = background
__global__ __launch_bounds__(256) void ccl_rank_kernel(const int32_t* __restrict__ stats, const int32_t* __restrict__ count, int cap,
                                                       const unsigned long long* __restrict__ keybits, const int32_t* __restrict__ wordrank,
                                                       const int32_t* __restrict__ blocksum, int32_t* __restrict__ rank, int32_t* __restrict__ out) {
  const int id = blockIdx.x * 256 + threadIdx.x;
  const int n = min(*count, cap);
  if (id > n) return;
  const int32_t* s = stats + (long long)(id == n ? cap : id) * kCclStat;
  int label = 0;
  if (id < n) {
    const int key = s[5], w = key >> 6;
    label = 1 + blocksum[w >> 10] + wordrank[w] + __popcll(keybits[w] & ((1ull << (key & 63)) - 1ull));
    rank[id] = label;
  }
  int32_t* o = out + (long long)label * 5;
  const bool any = s[0] > 0;
  o[0] = any ? s[1] : 0;
  o[1] = any ? s[3] : 0;
  o[2] = any ? s[2] - s[1] + 1 : 0;
  o[3] = any ? s[4] - s[3] + 1 : 0;
  o[4] = s[0];
}

__global__ __launch_bounds__(256) void ccl_relabel_kernel(const int32_t* __restrict__ parent, const int32_t* __restrict__ cid,
                                                          const int32_t* __restrict__ rank, const int32_t* __restrict__ count, int cap, long long n,
                                                          int32_t* __restrict__ labels) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const int r = parent[p];
  int v = 0;
  if (r >= 0) {
    if (!rank) v = cid[r] + 1;                                // unordered ids (numbering 2)
    else if (*count <= cap) v = rank[cid[r]];                 // more components than `cap`: the caller repeats the call; labels stay 0
  }
  labels[p] = v;
}

// out[p] = lut[labels[p]] (u8): keeps / drops whole components (every "remove regions whose statistic ..." step of encoder/ROI/*)
// ---- tile-parallel labelling (parallel.tiled_ccl): the ordering key of every component of a TILE in FRAME coordinates, i.e. the
// smallest key among its pixels -- numbering 0 (8-connectivity): the 2x2 block in block-raster order, (y / 2) * ceil(Wf / 2) + x / 2;
// numbering 1 (and 4-connectivity): the pixel in raster order, y * Wf + x.  Only the first pixel of every horizontal run asks for the
// atomic (one per run and row instead of one per pixel: a solid blob would serialise millions of atomics on one word).
__global__ __launch_bounds__(256) void ccl_keys_kernel(const int32_t* __restrict__ labels, int H, int W, int y0, int x0, int Wf, int block_keys,
                                                       uint32_t* __restrict__ keys) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= (long long)H * W) return;
  int y, x;
  rhccq_row_col(p, W, y, x);
  const int l = labels[p];
  if (l <= 0 || (x > 0 && labels[p - 1] == l)) return;
  const long long gy = y0 + y, gx = x0 + x;
  const unsigned key = block_keys ? (unsigned)((gy >> 1) * ((Wf + 1) >> 1) + (gx >> 1)) : (unsigned)(gy * Wf + gx);
  atomicMin(&keys[l], key);
}

__global__ __launch_bounds__(256) void ccl_select_kernel(const int32_t* __restrict__ labels, const uint8_t* __restrict__ lut, long long n,
                                                         uint8_t* __restrict__ out) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  out[p] = lut[labels[p]];
}

// ---- extract_roi_nonroi (roi.py:685-718): scipy.ndimage.binary_dilation(iterations = R) with the default cross-shaped
// structuring element = every pixel within L1 distance R of the set (border value 0); buffer zone = both dilations;
// the two masks and the two masked copies of the image in one pass.  32x8 tile + R apron of the region map in LDS.
constexpr int kRoiTW = 64, kRoiTH = 16, kRoiMaxR = 8;

__global__ __launch_bounds__(256) void roi_buffer_kernel(const uint8_t* __restrict__ region_map, const uint8_t* __restrict__ rgb, int H, int W, int R,
                                                         uint8_t* __restrict__ roi_mask, uint8_t* __restrict__ non_mask,
                                                         uint8_t* __restrict__ roi_img, uint8_t* __restrict__ non_img) {
  __shared__ uint8_t t[kRoiTH + 2 * kRoiMaxR][kRoiTW + 2 * kRoiMaxR + 4];
  const int tiles_x = (W + kRoiTW - 1) / kRoiTW;
  const int y0 = (blockIdx.x / tiles_x) * kRoiTH, x0 = (blockIdx.x % tiles_x) * kRoiTW;
  const int th = kRoiTH + 2 * R, tw = kRoiTW + 2 * R;
  for (int i = threadIdx.x; i < th * tw; i += 256) {
    const int ly = i / tw, lx = i % tw;
    const int y = y0 + ly - R, x = x0 + lx - R;
    uint8_t v = 2;                                          // outside the image: neither core (border value 0 of both dilations)
    if (y >= 0 && y < H && x >= 0 && x < W) {
      const uint8_t m = region_map[(long long)y * W + x];
      v = m == 1 ? 1 : (m == 0 ? 0 : 2);
    }
    t[ly][lx] = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kRoiTH * kRoiTW; i += 256) {
    const int ly = i / kRoiTW, lx = i % kRoiTW;
    const int y = y0 + ly, x = x0 + lx;
    if (y >= H || x >= W) continue;
    bool near_roi = false, near_non = false;
    for (int dy = -R; dy <= R; ++dy) {
      const int span = R - (dy < 0 ? -dy : dy);
      for (int dx = -span; dx <= span; ++dx) {
        const uint8_t v = t[ly + R + dy][lx + R + dx];
        near_roi |= v == 1;
        near_non |= v == 0;
      }
    }
    const uint8_t c = t[ly + R][lx + R];
    const bool buffer = near_roi && near_non;
    const bool in_roi = c == 1 || buffer, in_non = c == 0 || buffer;
    const long long p = (long long)y * W + x;
    roi_mask[p] = in_roi;
    non_mask[p] = in_non;
    const uint8_t r8 = rgb[3 * p], g8 = rgb[3 * p + 1], b8 = rgb[3 * p + 2];
    roi_img[3 * p] = in_roi ? r8 : 0; roi_img[3 * p + 1] = in_roi ? g8 : 0; roi_img[3 * p + 2] = in_roi ? b8 : 0;
    non_img[3 * p] = in_non ? r8 : 0; non_img[3 * p + 1] = in_non ? g8 : 0; non_img[3 * p + 2] = in_non ? b8 : 0;
  }
}

}  // namespace rhccq

using namespace rhccq;

extern "C" {

namespace {
struct CclWork {
  size_t bits, parent, cid, stats, keybits, wordrank, blocksum, rank, total;
  long long n_keys, n_words;
  int n_blocks;
};
CclWork ccl_layout(int H, int W, int cap, bool key_is_root) {
  CclWork w{};
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const long long segs = (W + 63) / 64, n = (long long)H * W;
  w.n_keys = key_is_root ? n : (long long)((H + 1) / 2) * ((W + 1) / 2);
  w.n_words = (w.n_keys + 63) / 64;
  w.n_blocks = (int)((w.n_words + 1023) / 1024);
  size_t o = 0;
  w.bits = o;     o = up(o + (size_t)H * segs * 8);
  w.parent = o;   o = up(o + (size_t)n * 4);
  w.cid = o;      o = up(o + (size_t)n * 4);
  w.stats = o;    o = up(o + ((size_t)cap + 1) * kCclStat * 4);
  w.keybits = o;  o = up(o + (size_t)w.n_words * 8);
  w.wordrank = o; o = up(o + (size_t)w.n_words * 4);
  w.blocksum = o; o = up(o + (size_t)w.n_blocks * 4);
  w.rank = o;     o = up(o + (size_t)(cap > 0 ? cap : 1) * 4);
  w.total = o;
  return w;
}
}  // namespace

int64_t rhccq_ccl_work_bytes(int32_t H, int32_t W, int32_t cap) {
  if (H <= 0 || W <= 0 || cap < 0) return 0;
  return (int64_t)ccl_layout(H, W, cap, true).total;            // raster keys need the larger planes
}

int rhccq_ccl(rhccq_ctx* ctx, const uint8_t* mask, int32_t H, int32_t W, int32_t connectivity, int32_t numbering, void* work, int64_t work_bytes,
              int32_t cap, int32_t* labels, int32_t* stats, int32_t* count) {
  if (!ctx || !mask || !work || !labels || !count || H <= 0 || W <= 0 || cap < 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "ccl: bad argument");
  if (connectivity != 4 && connectivity != 8) return rhccq_fail(ctx, RHCCQ_E_ARG, "ccl: connectivity must be 4 or 8");
  if (numbering < 0 || numbering > 2) return rhccq_fail(ctx, RHCCQ_E_ARG, "ccl: numbering must be 0 (OpenCV), 1 (raster) or 2 (unordered ids)");
  if (numbering != 2 && !stats) return rhccq_fail(ctx, RHCCQ_E_ARG, "ccl: statistics buffer missing");
  if ((long long)H * W >= 0x7fffffffLL) return rhccq_fail(ctx, RHCCQ_E_LIMIT, "ccl: more than 2^31 pixels");
  if (work_bytes < rhccq_ccl_work_bytes(H, W, cap)) return rhccq_fail(ctx, RHCCQ_E_ARG, "ccl: work buffer too small");
  const int key_is_root = (connectivity == 4 || numbering == 1) ? 1 : 0;
  const CclWork L = ccl_layout(H, W, cap, key_is_root != 0);
  char* base = (char*)work;
  uint64_t* bits = (uint64_t*)(base + L.bits);
  int32_t* parent = (int32_t*)(base + L.parent);
  int32_t* cid = (int32_t*)(base + L.cid);
  int32_t* raw = (int32_t*)(base + L.stats);
  unsigned long long* keybits = (unsigned long long*)(base + L.keybits);
  int32_t* wordrank = (int32_t*)(base + L.wordrank);
  int32_t* blocksum = (int32_t*)(base + L.blocksum);
  int32_t* rank = (int32_t*)(base + L.rank);
  const int segs = (W + 63) / 64;
  const long long n = (long long)H * W, units_total = (long long)H * segs;
  const unsigned wgrid = (unsigned)((units_total + 3) / 4), pgrid = (unsigned)((n + 255) / 256);
  const int units = 128;
  RHCCQ_HIP(ctx, hipMemsetAsync(count, 0, sizeof(int32_t), ctx->stream));
  RHCCQ_HIP(ctx, hipMemsetAsync(keybits, 0, (size_t)L.n_words * 8, ctx->stream));
  hipLaunchKernelGGL(ccl_init_kernel, dim3(wgrid), dim3(256), 0, ctx->stream, mask, H, W, segs, bits, parent);
  if (connectivity == 4)
    hipLaunchKernelGGL(ccl_merge_kernel<4>, dim3(wgrid), dim3(256), 0, ctx->stream, bits, H, W, segs, parent);
  else
    hipLaunchKernelGGL(ccl_merge_kernel<8>, dim3(wgrid), dim3(256), 0, ctx->stream, bits, H, W, segs, parent);
  if (numbering == 2) {
    // unordered ids: labels = compact id + 1; no statistics, no numbering (Canny's hysteresis only needs to tell components apart)
    hipLaunchKernelGGL(ccl_flatten_kernel, dim3((unsigned)((units_total + 15) / 16)), dim3(1024), 0, ctx->stream, parent, H, W, segs, cid, 0,
                       (int32_t*)nullptr, count, 0);
    hipLaunchKernelGGL(ccl_relabel_kernel, dim3(pgrid), dim3(256), 0, ctx->stream, parent, cid, (const int32_t*)nullptr, count, cap, n, labels);
    RHCCQ_LAUNCH_CHECK(ctx);
    return 0;
  }
  hipLaunchKernelGGL(ccl_flatten_kernel, dim3((unsigned)((units_total + 15) / 16)), dim3(1024), 0, ctx->stream, parent, H, W, segs, cid, cap, raw, count,
                     key_is_root);
  hipLaunchKernelGGL(ccl_stats_kernel, dim3((unsigned)((units_total + units - 1) / units)), dim3(1024), 0, ctx->stream, parent, cid, H, W, segs, cap, raw,
                     key_is_root, units);
  const unsigned cgrid = (unsigned)((cap + 1 + 255) / 256);
  hipLaunchKernelGGL(ccl_mark_kernel, dim3(cgrid), dim3(256), 0, ctx->stream, raw, count, cap, keybits);
  hipLaunchKernelGGL(ccl_scan_words_kernel, dim3((unsigned)L.n_blocks), dim3(1024), 0, ctx->stream, keybits, L.n_words, wordrank, blocksum);
  hipLaunchKernelGGL(ccl_scan_blocks_kernel, dim3(1), dim3(1024), 0, ctx->stream, blocksum, L.n_blocks);
  hipLaunchKernelGGL(ccl_rank_kernel, dim3(cgrid), dim3(256), 0, ctx->stream, raw, count, cap, keybits, wordrank, blocksum, rank, stats);
  hipLaunchKernelGGL(ccl_relabel_kernel, dim3(pgrid), dim3(256), 0, ctx->stream, parent, cid, rank, count, cap, n, labels);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_ccl_keys(rhccq_ctx* ctx, const int32_t* labels, int32_t H, int32_t W, int32_t y0, int32_t x0, int32_t frame_w, int32_t numbering,
                   int32_t connectivity, int32_t n_labels, uint32_t* keys) {
  if (!ctx || !labels || !keys || H <= 0 || W <= 0 || y0 < 0 || x0 < 0 || frame_w < x0 + W || n_labels < 0 || (numbering != 0 && numbering != 1) ||
      (connectivity != 4 && connectivity != 8))
    return rhccq_fail(ctx, RHCCQ_E_ARG, "ccl_keys: bad argument");
  const int block_keys = (numbering == 0 && connectivity == 8) ? 1 : 0;
  if (block_keys && ((y0 | x0) & 1)) return rhccq_fail(ctx, RHCCQ_E_ARG, "ccl_keys: block keys need a tile origin with even coordinates");
  RHCCQ_HIP(ctx, hipMemsetAsync(keys, 0xff, sizeof(uint32_t) * ((size_t)n_labels + 1), ctx->stream));
  hipLaunchKernelGGL(ccl_keys_kernel, dim3((unsigned)(((long long)H * W + 255) / 256)), dim3(256), 0, ctx->stream, labels, H, W, y0, x0, frame_w,
                     block_keys, keys);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_ccl_select(rhccq_ctx* ctx, const int32_t* labels, const uint8_t* lut, int64_t n_pixels, uint8_t* out) {
  if (!ctx || !labels || !lut || !out || n_pixels <= 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "ccl_select: bad argument");
  hipLaunchKernelGGL(ccl_select_kernel, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, ctx->stream, labels, lut, (long long)n_pixels, out);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_roi_buffer(rhccq_ctx* ctx, const uint8_t* region_map, const uint8_t* rgb, int32_t H, int32_t W, int32_t buffer_size, uint8_t* roi_mask,
                     uint8_t* nonroi_mask, uint8_t* roi_image, uint8_t* nonroi_image) {
  if (!ctx || !region_map || !rgb || !roi_mask || !nonroi_mask || !roi_image || !nonroi_image || H <= 0 || W <= 0)
    return rhccq_fail(ctx, RHCCQ_E_ARG, "roi_buffer: bad argument");
  if (buffer_size < 0 || buffer_size > kRoiMaxR) return rhccq_fail(ctx, RHCCQ_E_LIMIT, "roi_buffer: buffer_size outside 0..8");
  const unsigned grid = (unsigned)(((W + kRoiTW - 1) / kRoiTW) * (long long)((H + kRoiTH - 1) / kRoiTH));
  hipLaunchKernelGGL(roi_buffer_kernel, dim3(grid), dim3(256), 0, ctx->stream, region_map, rgb, H, W, buffer_size, roi_mask, nonroi_mask, roi_image, nonroi_image);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

}  // extern "C"
