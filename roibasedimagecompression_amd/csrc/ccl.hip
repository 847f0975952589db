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
//   ccl_flatten: parent[p] = root; roots take a compact id and initialise their statistics row.
//   ccl_stats:   area / bounding box / first 2x2 block per root; every value comes from the ballot of the lanes that share
//                a root (popcount, ctz, clz), one lane per (wave, root) issues the atomics.
//   ccl_relabel: labels[p] = rank[id[root]] (the host ranks the roots in OpenCV's numbering order, see api/roi.py).
// HBM bytes per pixel (algorithmic): 1 B mask read + 4 B parent written (init), 4 B read (merge, + the rare union walks),
// 4 + 4 (flatten), 4 (stats), 4 + 4 (relabel): 29 B/px for the whole labelling.
#include "rhccq_common.h"

namespace rhccq {

constexpr int kCclStat = 6;   // area, min x, max x, min y, max y, first block key

__device__ __forceinline__ int ccl_find(const int32_t* parent, int x) {
  int p = __hip_atomic_load(parent + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (p != x) { x = p; p = __hip_atomic_load(parent + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  return x;
}
__device__ __forceinline__ void ccl_union(int32_t* parent, int a, int b) {
  while (true) {
    a = ccl_find(parent, a);
    b = ccl_find(parent, b);
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

// key_is_root (4-connectivity, or raster numbering asked for): the ordering key of a component is its first pixel in raster order = the root itself
__global__ __launch_bounds__(256) void ccl_flatten_kernel(int32_t* parent, long long n, int32_t* __restrict__ cid, int cap, int32_t* __restrict__ stats,
                                                          int32_t* count, int key_is_root) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  if (p == 0) {                                             // statistics row `cap` = background (label 0 of cv2's stats)
    int32_t* s = stats + (long long)cap * kCclStat;
    s[0] = 0; s[1] = 0x7fffffff; s[2] = -1; s[3] = 0x7fffffff; s[4] = -1; s[5] = -1;
  }
  if (parent[p] < 0) return;
  const int r = ccl_find(parent, (int)p);
  parent[p] = r;
  if (r == (int)p) {
    const int id = atomicAdd(count, 1);
    cid[p] = id;
    if (id < cap) {
      int32_t* s = stats + (long long)id * kCclStat;
      s[0] = 0; s[1] = 0x7fffffff; s[2] = -1; s[3] = 0x7fffffff; s[4] = -1; s[5] = key_is_root ? r : 0x7fffffff;
    }
  }
}

__global__ __launch_bounds__(256) void ccl_stats_kernel(const int32_t* __restrict__ parent, const int32_t* __restrict__ cid, int H, int W, int segs,
                                                        int cap, int32_t* stats, int key_is_root) {
  int y, seg, lane;
  if (!ccl_wave_pos(H, segs, y, seg, lane)) return;
  const int x = seg * 64 + lane;
  const int r = x < W ? parent[(long long)y * W + x] : -2;  // -1 background, -2 outside
  uint64_t todo = __ballot(r != -2);
  const int w2 = (W + 1) >> 1;
  while (todo) {
    const int first = __builtin_ctzll(todo);
    const int lead = __shfl(r, first);
    const uint64_t same = __ballot(r == lead);
    todo &= ~same;
    if (lane == first) {
      const int id = lead < 0 ? cap : cid[lead];
      if (lead < 0 || id < cap) {
        int32_t* s = stats + (long long)id * kCclStat;
        const int x0 = seg * 64 + __builtin_ctzll(same), x1 = seg * 64 + 63 - __clzll((long long)same);
        atomicAdd(&s[0], __popcll(same));
        atomicMin(&s[1], x0);
        atomicMax(&s[2], x1);
        atomicMin(&s[3], y);
        atomicMax(&s[4], y);
        // first 2x2 block in block-raster order: only pixels of the root's row pair can hold it
        if (!key_is_root && lead >= 0 && (y >> 1) == ((lead / W) >> 1)) atomicMin(&s[5], (y >> 1) * w2 + (x0 >> 1));
      }
    }
  }
}

__global__ __launch_bounds__(256) void ccl_relabel_kernel(const int32_t* __restrict__ parent, const int32_t* __restrict__ cid,
                                                          const int32_t* __restrict__ rank, long long n, int32_t* __restrict__ labels) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const int r = parent[p];
  labels[p] = r < 0 ? 0 : rank[cid[r]];
}

// out[p] = lut[labels[p]] (u8): keeps / drops whole components (every "remove regions whose statistic ..." step of encoder/ROI/*)
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

int64_t rhccq_ccl_work_bytes(int32_t H, int32_t W) {
  if (H <= 0 || W <= 0) return 0;
  const long long segs = (W + 63) / 64;
  return (long long)H * segs * 8 + 256;                     // the bit plane
}

int rhccq_ccl_roots(rhccq_ctx* ctx, const uint8_t* mask, int32_t H, int32_t W, int32_t connectivity, int32_t numbering, void* work, int64_t work_bytes,
                    int32_t* parent, int32_t* cid, int32_t cap, int32_t* stats, int32_t* count) {
  if (!ctx || !mask || !work || !parent || !cid || !stats || !count || H <= 0 || W <= 0 || cap < 0)
    return rhccq_fail(ctx, RHCCQ_E_ARG, "ccl_roots: bad argument");
  if (connectivity != 4 && connectivity != 8) return rhccq_fail(ctx, RHCCQ_E_ARG, "ccl_roots: connectivity must be 4 or 8");
  if ((long long)H * W >= 0x7fffffffLL) return rhccq_fail(ctx, RHCCQ_E_LIMIT, "ccl_roots: more than 2^31 pixels");
  if (work_bytes < rhccq_ccl_work_bytes(H, W)) return rhccq_fail(ctx, RHCCQ_E_ARG, "ccl_roots: work buffer too small");
  if (numbering != 0 && numbering != 1) return rhccq_fail(ctx, RHCCQ_E_ARG, "ccl_roots: numbering must be 0 (OpenCV) or 1 (raster)");
  const int key_is_root = (connectivity == 4 || numbering == 1) ? 1 : 0;
  const int segs = (W + 63) / 64;
  const long long n = (long long)H * W;
  const unsigned wgrid = (unsigned)(((long long)H * segs + 3) / 4), pgrid = (unsigned)((n + 255) / 256);
  uint64_t* bits = (uint64_t*)work;
  RHCCQ_HIP(ctx, hipMemsetAsync(count, 0, sizeof(int32_t), ctx->stream));
  hipLaunchKernelGGL(ccl_init_kernel, dim3(wgrid), dim3(256), 0, ctx->stream, mask, H, W, segs, bits, parent);
  if (connectivity == 4)
    hipLaunchKernelGGL(ccl_merge_kernel<4>, dim3(wgrid), dim3(256), 0, ctx->stream, bits, H, W, segs, parent);
  else
    hipLaunchKernelGGL(ccl_merge_kernel<8>, dim3(wgrid), dim3(256), 0, ctx->stream, bits, H, W, segs, parent);
  hipLaunchKernelGGL(ccl_flatten_kernel, dim3(pgrid), dim3(256), 0, ctx->stream, parent, n, cid, cap, stats, count, key_is_root);
  hipLaunchKernelGGL(ccl_stats_kernel, dim3(wgrid), dim3(256), 0, ctx->stream, parent, cid, H, W, segs, cap, stats, key_is_root);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_ccl_relabel(rhccq_ctx* ctx, const int32_t* parent, const int32_t* cid, const int32_t* rank, int64_t n_pixels, int32_t* labels) {
  if (!ctx || !parent || !cid || !rank || !labels || n_pixels <= 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "ccl_relabel: bad argument");
  hipLaunchKernelGGL(ccl_relabel_kernel, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, ctx->stream, parent, cid, rank, (long long)n_pixels, labels);
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
