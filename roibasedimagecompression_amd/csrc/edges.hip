// Edge front end of the ROI stage (reference encoder/ROI/edges.py:35-71,173-195): what `get_edge_map` takes from
// OpenCV -- cvtColor(RGB2GRAY), Sobel, threshold(OTSU), 21 x Canny -- as integer kernels.  PARITY UNPINNED (OpenCV is absent from
// the build container); every operation below is integer arithmetic restated from OpenCV's published implementation:
//   gray   = (4899 R + 9617 G + 1868 B + 8192) >> 14                                   (cvtColor, 8-bit RGB2GRAY, fixed point)
//   Sobel  = the 3x3 kernels [-1 0 1; -2 0 2; -1 0 1] and its transpose; BORDER_REFLECT_101 for cv2.Sobel's default,
//            BORDER_REPLICATE inside cv2.Canny
//   Canny  = L1 magnitude |dx| + |dy| (per pixel the channel with the largest one, first on ties), non-maximum suppression with
//            the fixed-point tangents TG22 = round(0.41421356 * 2^15) and tg67 = tg22 + 2 |dx| 2^15 (comparisons: "> left and >= right",
//            "> up and >= down", both ">" on the diagonals), hysteresis = 8-connected growth of the pixels above `high` through the
//            pixels above `low`.
// Non-maximum suppression does not depend on the thresholds, so it runs ONCE per image (nm[p] = magnitude of a local maximum,
// else 0); each of the 21 threshold pairs then costs one mask + one connected-component labelling (csrc/ccl.hip) + one per-label
// reduction (largest magnitude, sum and sum of squares of the gray values): a component belongs to the edge map iff its largest
// magnitude exceeds `high`, and the quality score of edges.py:73-86 needs nothing but those per-label numbers.
#include <algorithm>
#include <vector>

#include "rhccq_common.h"

namespace rhccq {

constexpr int kM2Bins = 2 * 1020 * 1020 + 1;   // gx^2 + gy^2 of a 3x3 Sobel on 8-bit data
constexpr int kM2Lds = 8192;

// BORDER_REFLECT_101 for any offset (a window wider than the image reflects more than once): period 2n - 2
__device__ __forceinline__ int reflect101(int i, int n) {
  if (n == 1) return 0;
  const int period = 2 * n - 2;
  i %= period;
  if (i < 0) i += period;
  return i < n ? i : period - i;
}
__device__ __forceinline__ int clampi(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }

__global__ __launch_bounds__(256) void edges_gray_kernel(const uint8_t* __restrict__ rgb, long long n, uint8_t* __restrict__ gray, int32_t* hist) {
  __shared__ int s_h[256];
  s_h[threadIdx.x] = 0;
  __syncthreads();
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < n; p += (long long)gridDim.x * 256) {
    const int g = (4899 * rgb[3 * p] + 9617 * rgb[3 * p + 1] + 1868 * rgb[3 * p + 2] + 8192) >> 14;
    gray[p] = (uint8_t)g;
    atomicAdd(&s_h[g], 1);
  }
  __syncthreads();
  if (s_h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], s_h[threadIdx.x]);
}

// 3x3 Sobel of one channel at (y, x); kReplicate: BORDER_REPLICATE (Canny), else BORDER_REFLECT_101 (cv2.Sobel's default)
template <int kCn, bool kReplicate>
__device__ __forceinline__ void sobel3(const uint8_t* __restrict__ img, int H, int W, int y, int x, int c, int& gx, int& gy) {
  const int ym = kReplicate ? clampi(y - 1, H) : reflect101(y - 1, H), yp = kReplicate ? clampi(y + 1, H) : reflect101(y + 1, H);
  const int xm = kReplicate ? clampi(x - 1, W) : reflect101(x - 1, W), xp = kReplicate ? clampi(x + 1, W) : reflect101(x + 1, W);
  auto at = [&](int yy, int xx) { return (int)img[((long long)yy * W + xx) * kCn + c]; };
  const int a = at(ym, xm), b = at(ym, x), cc = at(ym, xp), d = at(y, xm), f = at(y, xp), g = at(yp, xm), h = at(yp, x), i = at(yp, xp);
  gx = (cc + 2 * f + i) - (a + 2 * d + g);
  gy = (g + 2 * h + i) - (a + 2 * b + cc);
}

// histogram of gx^2 + gy^2 (BORDER_REFLECT_101): everything edges.py:88-160 derives from the float64 gradient magnitude
// (mean, standard deviation, percentiles of the non-zero values) follows from it on the host
__global__ __launch_bounds__(256) void edges_gradhist_kernel(const uint8_t* __restrict__ gray, int H, int W, int32_t* hist) {
  __shared__ int s_h[kM2Lds];
  for (int i = threadIdx.x; i < kM2Lds; i += 256) s_h[i] = 0;
  __syncthreads();
  const long long n = (long long)H * W;
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < n; p += (long long)gridDim.x * 256) {
    int gx, gy;
    int y, x;
    rhccq_row_col(p, W, y, x);
    sobel3<1, false>(gray, H, W, y, x, 0, gx, gy);
    const int m2 = gx * gx + gy * gy;
    if (m2 < kM2Lds) atomicAdd(&s_h[m2], 1);
    else atomicAdd(&hist[m2], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kM2Lds; i += 256)
    if (s_h[i]) atomicAdd(&hist[i], s_h[i]);
}

// Canny's gradient: per pixel the channel with the largest |dx| + |dy| (first on ties); mag u16, (dx, dy) as two int16
template <int kCn>
__global__ __launch_bounds__(256) void canny_grad_kernel(const uint8_t* __restrict__ img, int H, int W, uint16_t* __restrict__ mag, int32_t* __restrict__ dxy) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= (long long)H * W) return;
  int y, x;
  rhccq_row_col(p, W, y, x);
  int bx = 0, by = 0, bm = -1;
#pragma unroll
  for (int c = 0; c < kCn; ++c) {
    int gx, gy;
    sobel3<kCn, true>(img, H, W, y, x, c, gx, gy);
    const int m = abs(gx) + abs(gy);
    if (m > bm) { bm = m; bx = gx; by = gy; }
  }
  mag[p] = (uint16_t)bm;
  dxy[p] = (int32_t)(((uint32_t)(uint16_t)(int16_t)bx) | ((uint32_t)(uint16_t)(int16_t)by << 16));
}

// nm[p] = mag[p] when p is a local maximum along its gradient direction, else 0 (magnitudes outside the image count as 0)
__global__ __launch_bounds__(256) void canny_nms_kernel(const uint16_t* __restrict__ mag, const int32_t* __restrict__ dxy, int H, int W,
                                                        uint16_t* __restrict__ nm) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= (long long)H * W) return;
  int y, x;
  rhccq_row_col(p, W, y, x);
  const int m = mag[p];
  uint16_t out = 0;
  if (m > 0) {
    auto at = [&](int yy, int xx) { return (yy < 0 || yy >= H || xx < 0 || xx >= W) ? 0 : (int)mag[(long long)yy * W + xx]; };
    const int32_t d = dxy[p];
    const int xs = (int)(int16_t)(d & 0xffff), ys = (int)(int16_t)((uint32_t)d >> 16);
    const int ax = abs(xs), ay = abs(ys) << 15;
    const int tg22x = ax * 13573;                           // TG22 = (int)(0.4142135623730950488 * 2^15 + 0.5)
    bool is_max;
    if (ay < tg22x) is_max = m > at(y, x - 1) && m >= at(y, x + 1);
    else {
      const int tg67x = tg22x + (ax << 16);
      if (ay > tg67x) is_max = m > at(y - 1, x) && m >= at(y + 1, x);
      else {
        const int s = (xs ^ ys) < 0 ? -1 : 1;
        is_max = m > at(y - 1, x - s) && m > at(y + 1, x + s);
      }
    }
    if (is_max) out = (uint16_t)m;
  }
  nm[p] = out;
}

__global__ __launch_bounds__(256) void edges_above_kernel(const uint16_t* __restrict__ nm, long long n, int low, uint8_t* __restrict__ mask) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p < n) mask[p] = nm[p] > low;
}

// per label: red[l] = {largest value of `val16`, sum of `val8`, sum of val8^2, number of pixels} over the label's pixels (int64 each); one wave
// covers 64 consecutive pixels: when all its foreground lanes share one label (the common case on edge maps: short runs) the
// wave reduces first and issues three atomics
__global__ __launch_bounds__(256) void label_reduce_kernel(const int32_t* __restrict__ labels, const uint16_t* __restrict__ val16,
                                                           const uint8_t* __restrict__ val8, long long n, unsigned long long* red) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int l = p < n ? labels[p] : 0;
  const unsigned long long fg = __ballot(l != 0);
  if (!fg) return;
  unsigned long long v16 = l && val16 ? val16[p] : 0ull, v8 = l && val8 ? val8[p] : 0ull;
  const int first = __builtin_ctzll(fg);
  const int lead = __shfl(l, first);
  if (__ballot(l != 0 && l != lead) == 0) {
    unsigned long long mx = v16, s1 = v8, s2 = v8 * v8;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mx = max(mx, (unsigned long long)__shfl_down(mx, o, 64));
      s1 += __shfl_down(s1, o, 64);
      s2 += __shfl_down(s2, o, 64);
    }
    if (lane == 0) {
      atomicMax(&red[4ll * lead], mx);
      if (s1) { atomicAdd(&red[4ll * lead + 1], s1); atomicAdd(&red[4ll * lead + 2], s2); }
      atomicAdd(&red[4ll * lead + 3], (unsigned long long)__popcll(fg));
    }
  } else if (l) {
    atomicMax(&red[4ll * l], v16);
    if (v8) { atomicAdd(&red[4ll * l + 1], v8); atomicAdd(&red[4ll * l + 2], v8 * v8); }
    atomicAdd(&red[4ll * l + 3], 1ull);
  }
}

// ---- the same scores from ONE union-find grown over the thresholds (round 4) ---------------------------------------------------------------
// The sets {nm > low} nest for descending `low`: the components at a lower threshold are unions of the components at the higher one plus the
// pixels in between.  So the lows are visited in DESCENDING order and the forest is never rebuilt: a level adds its new pixels (each is its own
// root), links every new pixel with its 8-neighbours that are in the set (lock-free: the larger root under the smaller, as csrc/ccl.hip), and
// the verdict of a (low, high) pair needs no component numbering at all: a pixel above `high` marks its root with the pair's generation number,
// a second pass sums the pixels whose root carries it.  Every pixel is linked exactly once over the whole search (19 labellings from scratch
// before: ~0.5 ms each at 4K).  Same four numbers as rhccq_label_reduce + rhccq_edge_score, by definition.
template <bool kHalve>
__device__ __forceinline__ int hy_find(int32_t* parent, int x) {
  int p = __hip_atomic_load(parent + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (p != x) {
    const int gp = __hip_atomic_load(parent + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (kHalve && gp != p) __hip_atomic_store(parent + x, gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    x = p;
    p = gp;
  }
  return x;
}
__device__ __forceinline__ void hy_union(int32_t* parent, int a, int b) {
  while (true) {
    a = hy_find<true>(parent, a);
    b = hy_find<true>(parent, b);
    if (a == b) return;
    if (a > b) { const int t = a; a = b; b = t; }
    const int old = atomicCAS(&parent[b], b, a);
    if (old == b) return;
    b = old;
  }
}
// pixels with lo < nm <= hi_prev enter the forest as roots (hi_prev = 65535 + 1 for the first level: nm is 16 bits)
__global__ __launch_bounds__(256) void hy_add_kernel(const uint16_t* __restrict__ nm, long long n, int lo, int hi_prev, int32_t* __restrict__ parent) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const int v = nm[p];
  if (v > lo && v <= hi_prev) parent[p] = (int32_t)p;
}
// every new pixel is linked with its 8-neighbours that are in the set {nm > lo} (new or old)
__global__ __launch_bounds__(256) void hy_link_kernel(const uint16_t* __restrict__ nm, int H, int W, int lo, int hi_prev, int32_t* parent) {
  const long long n = (long long)H * W;
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const int v = nm[p];
  if (!(v > lo && v <= hi_prev)) return;
  int y, x;
  rhccq_row_col(p, W, y, x);
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      if (dy == 0 && dx == 0) continue;
      const int yy = y + dy, xx = x + dx;
      if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
      const long long q = (long long)yy * W + xx;
      const int vq = nm[q];
      if (vq <= lo) continue;
      // two NEW neighbours would both try: the smaller index does it (an old neighbour never does)
      if (vq <= hi_prev && q > p) continue;
      hy_union(parent, (int)p, (int)q);
    }
}
// root of every pixel of the set (-1 outside)
__global__ __launch_bounds__(256) void hy_roots_kernel(const uint16_t* __restrict__ nm, long long n, int lo, int32_t* parent, int32_t* __restrict__ root) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  root[p] = nm[p] > lo ? hy_find<true>(parent, (int)p) : -1;
}
// a pixel above `high` marks its component: flag[root] = gen (generations only grow: no clearing between pairs)
__global__ __launch_bounds__(256) void hy_mark_kernel(const uint16_t* __restrict__ nm, long long n, int high, const int32_t* __restrict__ root,
                                                      int32_t* flag, int gen) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  int r = -1;
  if (p < n && nm[p] > high) r = root[p];
  // PLAIN stores of the same value, one per distinct root of the wave: no atomic, no read of the flag.  (A noisy image has one giant component:
  // a million atomics -- or agent-scope loads -- on its root's flag serialise at the memory side, 360 us per pair at 4K; the marked components
  // are counted by the next pass instead, at their root pixels.)
  unsigned long long todo = __ballot(r >= 0);
  while (todo) {
    const int leader = __builtin_ctzll(todo);
    const int r0 = __shfl(r, leader, 64);
    const unsigned long long same = __ballot(r == r0);
    if ((int)(threadIdx.x & 63) == leader) flag[r0] = gen;
    todo &= ~same;
  }
}
// edge pixels, sum and sum of squares of gray over the marked components
__global__ __launch_bounds__(256) void hy_sums_kernel(const uint8_t* __restrict__ gray, long long n, const int32_t* __restrict__ root,
                                                      const int32_t* __restrict__ flag, int gen, unsigned long long* out) {
  __shared__ unsigned long long s_acc[4];
  if (threadIdx.x < 4) s_acc[threadIdx.x] = 0;
  __syncthreads();
  unsigned long long v[4] = {0, 0, 0, 0};
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < n; p += (long long)gridDim.x * 256) {
    const int r = root[p];
    if (r >= 0 && flag[r] == gen) {
      const unsigned long long g = gray ? gray[p] : 0;
      v[0] += (long long)r == p;                          // a marked component is counted at its root pixel
      v[1] += 1; v[2] += g; v[3] += g * g;
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    unsigned long long t = v[k];
    for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
    if ((threadIdx.x & 63) == 0 && t) atomicAdd(&s_acc[k], t);
  }
  __syncthreads();
  if (threadIdx.x < 4 && s_acc[threadIdx.x]) atomicAdd(&out[threadIdx.x], s_acc[threadIdx.x]);
}

// the same with a capacity: labels beyond `cap` are left out (the caller sees n > cap in the scores record and takes the two-step path)
__global__ __launch_bounds__(256) void label_reduce_capped_kernel(const int32_t* __restrict__ labels, const uint16_t* __restrict__ val16,
                                                                  const uint8_t* __restrict__ val8, long long n, int cap, unsigned long long* red) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  int l = p < n ? labels[p] : 0;
  if (l > cap) l = 0;
  const unsigned long long fg = __ballot(l != 0);
  if (!fg) return;
  unsigned long long v16 = l ? val16[p] : 0ull, v8 = l && val8 ? val8[p] : 0ull;
  const int first = __builtin_ctzll(fg);
  const int lead = __shfl(l, first);
  if (__ballot(l != 0 && l != lead) == 0) {
    unsigned long long mx = v16, s1 = v8, s2 = v8 * v8;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mx = max(mx, (unsigned long long)__shfl_down(mx, o, 64));
      s1 += __shfl_down(s1, o, 64);
      s2 += __shfl_down(s2, o, 64);
    }
    if (lane == 0) {
      atomicMax(&red[4ll * lead], mx);
      if (s1) { atomicAdd(&red[4ll * lead + 1], s1); atomicAdd(&red[4ll * lead + 2], s2); }
      atomicAdd(&red[4ll * lead + 3], (unsigned long long)__popcll(fg));
    }
  } else if (l) {
    atomicMax(&red[4ll * l], v16);
    if (v8) { atomicAdd(&red[4ll * l + 1], v8); atomicAdd(&red[4ll * l + 2], v8 * v8); }
    atomicAdd(&red[4ll * l + 3], 1ull);
  }
}

// edge_score_kernel with the component count read on the device: out[0..3] as there, out[4] = the component count itself
__global__ __launch_bounds__(256) void edge_score_dev_kernel(const unsigned long long* __restrict__ red, const int32_t* __restrict__ n_dev, int cap, int high,
                                                             unsigned long long* out) {
  __shared__ unsigned long long s_acc[4];
  const int n_all = *n_dev, n = min(n_all, cap);
  if (blockIdx.x * 256 > n) return;
  if (threadIdx.x < 4) s_acc[threadIdx.x] = 0;
  __syncthreads();
  const int l = blockIdx.x * 256 + threadIdx.x;
  unsigned long long v[4] = {0, 0, 0, 0};
  if (l <= n && l > 0 && red[4ll * l] > (unsigned long long)high) { v[0] = 1; v[1] = red[4ll * l + 3]; v[2] = red[4ll * l + 1]; v[3] = red[4ll * l + 2]; }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    unsigned long long t = v[k];
    for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
    if ((threadIdx.x & 63) == 0 && t) atomicAdd(&s_acc[k], t);
  }
  __syncthreads();
  if (threadIdx.x < 4 && s_acc[threadIdx.x]) atomicAdd(&out[threadIdx.x], s_acc[threadIdx.x]);
  if (blockIdx.x == 0 && threadIdx.x == 0) out[4] = (unsigned long long)n_all;
}

// hysteresis verdict per component and the four numbers the quality score needs, without a trip to the host: a component is an edge
// when its largest magnitude exceeds `high`; out = {edge components, edge pixels, sum of gray, sum of gray^2 over the edge pixels};
// lut (optional): 255 for edge components, 0 otherwise (label 0 = background: 0)
__global__ __launch_bounds__(256) void edge_score_kernel(const unsigned long long* __restrict__ red /* [n + 1][4] */, int n, int high,
                                                         unsigned long long* out, uint8_t* __restrict__ lut) {
  __shared__ unsigned long long s_acc[4];
  if (threadIdx.x < 4) s_acc[threadIdx.x] = 0;
  __syncthreads();
  const int l = blockIdx.x * 256 + threadIdx.x;             // label l, 0 = background
  unsigned long long v[4] = {0, 0, 0, 0};
  if (l <= n) {
    const bool edge = l > 0 && red[4ll * l] > (unsigned long long)high;
    if (lut) lut[l] = edge ? 255 : 0;
    if (edge) { v[0] = 1; v[1] = red[4ll * l + 3]; v[2] = red[4ll * l + 1]; v[3] = red[4ll * l + 2]; }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    unsigned long long t = v[k];
    for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
    if ((threadIdx.x & 63) == 0 && t) atomicAdd(&s_acc[k], t);
  }
  __syncthreads();
  if (threadIdx.x < 4 && s_acc[threadIdx.x]) atomicAdd(&out[threadIdx.x], s_acc[threadIdx.x]);
}

// number of non-zero pixels (kSum: sum of the pixel values) in the k x k window centred on each pixel, BORDER_REFLECT_101 (cv2.filter2D's
// default border), k odd <= 31
constexpr int kBoxTW = 64, kBoxTH = 16, kBoxMaxR = 15;
template <bool kSum, typename OutT>
__global__ __launch_bounds__(256) void box_count_kernel(const uint8_t* __restrict__ mask, int H, int W, int r, OutT* __restrict__ out) {
  __shared__ uint8_t t[kBoxTH + 2 * kBoxMaxR][kBoxTW + 2 * kBoxMaxR + 2];
  __shared__ OutT hsum[kBoxTH + 2 * kBoxMaxR][kBoxTW];
  const int tiles_x = (W + kBoxTW - 1) / kBoxTW;
  const int y0 = (blockIdx.x / tiles_x) * kBoxTH, x0 = (blockIdx.x % tiles_x) * kBoxTW;
  const int th = kBoxTH + 2 * r, tw = kBoxTW + 2 * r;
  for (int i = threadIdx.x; i < th * tw; i += 256) {
    const int ly = i / tw, lx = i % tw;
    const int y = reflect101(y0 + ly - r, H), x = reflect101(x0 + lx - r, W);
    // (tiles that stick out of the image reflect far coordinates back inside: those outputs are never stored)
    const uint8_t v = mask[(long long)clampi(y, H) * W + clampi(x, W)];
    t[ly][lx] = kSum ? v : (uint8_t)(v != 0);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < th * kBoxTW; i += 256) {
    const int ly = i / kBoxTW, lx = i % kBoxTW;
    unsigned s = 0;
    for (int d = 0; d <= 2 * r; ++d) s += t[ly][lx + d];
    hsum[ly][lx] = (OutT)s;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kBoxTH * kBoxTW; i += 256) {
    const int ly = i / kBoxTW, lx = i % kBoxTW;
    const int y = y0 + ly, x = x0 + lx;
    if (y >= H || x >= W) continue;
    unsigned s = 0;
    for (int d = 0; d <= 2 * r; ++d) s += hsum[ly + d][lx];
    out[(long long)y * W + x] = (OutT)s;
  }
}

}  // namespace rhccq

using namespace rhccq;

extern "C" {

int64_t rhccq_edges_m2_bins(void) { return kM2Bins; }

int rhccq_edges_gray(rhccq_ctx* ctx, const uint8_t* rgb, int64_t n_pixels, uint8_t* gray, int32_t* hist256) {
  if (!ctx || !rgb || !gray || !hist256 || n_pixels <= 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "edges_gray: bad argument");
  RHCCQ_HIP(ctx, hipMemsetAsync(hist256, 0, 256 * sizeof(int32_t), ctx->stream));
  const unsigned grid = (unsigned)((n_pixels + 255) / 256 < 2048 ? (n_pixels + 255) / 256 : 2048);
  hipLaunchKernelGGL(edges_gray_kernel, dim3(grid), dim3(256), 0, ctx->stream, rgb, (long long)n_pixels, gray, hist256);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_edges_grad_hist(rhccq_ctx* ctx, const uint8_t* gray, int32_t H, int32_t W, int32_t* hist_m2) {
  if (!ctx || !gray || !hist_m2 || H <= 0 || W <= 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "edges_grad_hist: bad argument");
  RHCCQ_HIP(ctx, hipMemsetAsync(hist_m2, 0, (size_t)kM2Bins * sizeof(int32_t), ctx->stream));
  const long long n = (long long)H * W;
  const unsigned grid = (unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  hipLaunchKernelGGL(edges_gradhist_kernel, dim3(grid), dim3(256), 0, ctx->stream, gray, H, W, hist_m2);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_canny_nms(rhccq_ctx* ctx, const uint8_t* img, int32_t H, int32_t W, int32_t channels, uint16_t* mag_tmp, int32_t* dxy_tmp, uint16_t* nm) {
  if (!ctx || !img || !mag_tmp || !dxy_tmp || !nm || H <= 0 || W <= 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "canny_nms: bad argument");
  if (channels != 1 && channels != 3) return rhccq_fail(ctx, RHCCQ_E_ARG, "canny_nms: 1 or 3 channels");
  const unsigned grid = (unsigned)(((long long)H * W + 255) / 256);
  if (channels == 1) hipLaunchKernelGGL(canny_grad_kernel<1>, dim3(grid), dim3(256), 0, ctx->stream, img, H, W, mag_tmp, dxy_tmp);
  else hipLaunchKernelGGL(canny_grad_kernel<3>, dim3(grid), dim3(256), 0, ctx->stream, img, H, W, mag_tmp, dxy_tmp);
  hipLaunchKernelGGL(canny_nms_kernel, dim3(grid), dim3(256), 0, ctx->stream, mag_tmp, dxy_tmp, H, W, nm);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_edges_above(rhccq_ctx* ctx, const uint16_t* nm, int64_t n_pixels, int32_t low, uint8_t* mask) {
  if (!ctx || !nm || !mask || n_pixels <= 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "edges_above: bad argument");
  hipLaunchKernelGGL(edges_above_kernel, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, ctx->stream, nm, (long long)n_pixels, low, mask);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_label_reduce(rhccq_ctx* ctx, const int32_t* labels, const uint16_t* val16, const uint8_t* val8, int64_t n_pixels, int32_t n_labels,
                       uint64_t* red) {
  if (!ctx || !labels || !red || n_pixels <= 0 || n_labels < 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "label_reduce: bad argument");
  RHCCQ_HIP(ctx, hipMemsetAsync(red, 0, 4 * sizeof(uint64_t) * ((size_t)n_labels + 1), ctx->stream));
  hipLaunchKernelGGL(label_reduce_kernel, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, ctx->stream, labels, val16, val8, (long long)n_pixels,
                     (unsigned long long*)red);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_edge_score(rhccq_ctx* ctx, const uint64_t* red, int32_t n_labels, int32_t high, uint64_t* out4, uint8_t* lut) {
  if (!ctx || !red || !out4 || n_labels < 0 || high < 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "edge_score: bad argument");
  RHCCQ_HIP(ctx, hipMemsetAsync(out4, 0, 4 * sizeof(uint64_t), ctx->stream));
  hipLaunchKernelGGL(edge_score_kernel, dim3((unsigned)((n_labels + 1 + 255) / 256)), dim3(256), 0, ctx->stream, (const unsigned long long*)red, n_labels, high,
                     (unsigned long long*)out4, lut);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

// The scores of SEVERAL Canny threshold pairs without a trip to the host in between (get_edge_map tries 20 pairs, encoder/ROI/edges.py:40-71): per
// distinct `low` one mask, one labelling of {nm > low} (unordered ids, the count stays on the device), one capped per-label reduction; per pair
// one verdict kernel.  out (device uint64[n_pairs][5]): edge components, edge pixels, sum gray, sum gray^2, and the number of components of
// {nm > low} -- when that exceeds `cap` the pair's four numbers are incomplete and the caller scores it through rhccq_label_reduce /
// rhccq_edge_score.  Pairs must be grouped by `low` (equal lows adjacent).  work: rhccq_canny_scores_bytes(H, W, cap) bytes.
int64_t rhccq_canny_scores_bytes(int32_t H, int32_t W, int32_t cap) {
  if (H <= 0 || W <= 0 || cap < 0) return 0;
  const size_t n = (size_t)H * W;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  return (int64_t)(al(n) + al(4 * n) + al(32 * ((size_t)cap + 1)) + 256 + al((size_t)rhccq_ccl_work_bytes(H, W, 0)));
}

int rhccq_canny_scores(rhccq_ctx* ctx, const uint16_t* nm, const uint8_t* gray, int32_t H, int32_t W, const int32_t* lows_host, const int32_t* highs_host,
                       int32_t n_pairs, int32_t cap, void* work, int64_t work_bytes, uint64_t* out) {
  if (!ctx || !nm || !lows_host || !highs_host || !work || !out || H <= 0 || W <= 0 || n_pairs <= 0 || cap < 1)
    return rhccq_fail(ctx, RHCCQ_E_ARG, "canny_scores: bad argument");
  if (work_bytes < rhccq_canny_scores_bytes(H, W, cap)) return rhccq_fail(ctx, RHCCQ_E_ARG, "canny_scores: work buffer too small");
  const size_t n = (size_t)H * W;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  char* base = (char*)work;
  uint8_t* mask = (uint8_t*)base; base += al(n);
  int32_t* labels = (int32_t*)base; base += al(4 * n);
  unsigned long long* red = (unsigned long long*)base; base += al(32 * ((size_t)cap + 1));
  int32_t* count = (int32_t*)base; base += 256;
  void* cwork = (void*)base;
  const int64_t cbytes = rhccq_ccl_work_bytes(H, W, 0);
  RHCCQ_HIP(ctx, hipMemsetAsync(out, 0, (size_t)n_pairs * 5 * sizeof(uint64_t), ctx->stream));
  for (int i = 0; i < n_pairs; ++i) {
    if (lows_host[i] < 0 || highs_host[i] < 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "canny_scores: negative threshold");
    if (i == 0 || lows_host[i] != lows_host[i - 1]) {
      if (int e = rhccq_edges_above(ctx, nm, (int64_t)n, lows_host[i], mask)) return e;
      if (int e = rhccq_ccl(ctx, mask, H, W, 8, 2, cwork, cbytes, 0, labels, nullptr, count)) return e;
      RHCCQ_HIP(ctx, hipMemsetAsync(red, 0, 32 * ((size_t)cap + 1), ctx->stream));
      hipLaunchKernelGGL(label_reduce_capped_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, labels, nm, gray, (long long)n, (int)cap, red);
    }
    hipLaunchKernelGGL(edge_score_dev_kernel, dim3((unsigned)(((size_t)cap + 1 + 255) / 256)), dim3(256), 0, ctx->stream, (const unsigned long long*)red,
                       (const int32_t*)count, (int)cap, (int)highs_host[i], (unsigned long long*)out + 5 * (size_t)i);
  }
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

// The same scores through ONE union-find grown over the thresholds (kernels above).  Pairs in any order; out as rhccq_canny_scores with
// out[i][4] = 0.  work: rhccq_canny_scores_nested_bytes(H, W) bytes (three int32 planes).
int64_t rhccq_canny_scores_nested_bytes(int32_t H, int32_t W) {
  if (H <= 0 || W <= 0) return 0;
  return (int64_t)(3 * (((size_t)H * W * 4 + 255) & ~(size_t)255));
}

int rhccq_canny_scores_nested(rhccq_ctx* ctx, const uint16_t* nm, const uint8_t* gray, int32_t H, int32_t W, const int32_t* lows_host,
                              const int32_t* highs_host, int32_t n_pairs, void* work, int64_t work_bytes, uint64_t* out) {
  if (!ctx || !nm || !lows_host || !highs_host || !work || !out || H <= 0 || W <= 0 || n_pairs <= 0 || (int64_t)H * W > INT32_MAX)
    return rhccq_fail(ctx, RHCCQ_E_ARG, "canny_scores_nested: bad argument");
  if (work_bytes < rhccq_canny_scores_nested_bytes(H, W)) return rhccq_fail(ctx, RHCCQ_E_ARG, "canny_scores_nested: work buffer too small");
  const long long n = (long long)H * W;
  const size_t plane = ((size_t)n * 4 + 255) & ~(size_t)255;
  int32_t* parent = (int32_t*)work;
  int32_t* root = (int32_t*)((char*)work + plane);
  int32_t* flag = (int32_t*)((char*)work + 2 * plane);
  std::vector<int> order((size_t)n_pairs);
  for (int i = 0; i < n_pairs; ++i) {
    if (lows_host[i] < 0 || highs_host[i] < 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "canny_scores_nested: negative threshold");
    order[(size_t)i] = i;
  }
  std::sort(order.begin(), order.end(), [&](int a, int b) { return lows_host[a] != lows_host[b] ? lows_host[a] > lows_host[b] : a < b; });   // descending low
  RHCCQ_HIP(ctx, hipMemsetAsync(out, 0, (size_t)n_pairs * 5 * sizeof(uint64_t), ctx->stream));
  RHCCQ_HIP(ctx, hipMemsetAsync(flag, 0, (size_t)n * 4, ctx->stream));
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  int hi_prev = 1 << 16, gen = 0, cur_low = -1;
  for (int oi = 0; oi < n_pairs; ++oi) {
    const int i = order[(size_t)oi];
    if (oi == 0 || lows_host[i] != cur_low) {
      cur_low = lows_host[i];
      hipLaunchKernelGGL(hy_add_kernel, grid, block, 0, ctx->stream, nm, n, cur_low, hi_prev, parent);
      hipLaunchKernelGGL(hy_link_kernel, grid, block, 0, ctx->stream, nm, (int)H, (int)W, cur_low, hi_prev, parent);
      hipLaunchKernelGGL(hy_roots_kernel, grid, block, 0, ctx->stream, nm, n, cur_low, parent, root);
      hi_prev = cur_low;
    }
    ++gen;
    unsigned long long* o = (unsigned long long*)out + 5 * (size_t)i;
    hipLaunchKernelGGL(hy_mark_kernel, grid, block, 0, ctx->stream, nm, n, (int)highs_host[i], (const int32_t*)root, flag, gen);
    hipLaunchKernelGGL(hy_sums_kernel, dim3(2048), block, 0, ctx->stream, gray, n, (const int32_t*)root, (const int32_t*)flag, gen, o);
  }
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

static int box_launch(rhccq_ctx* ctx, const uint8_t* mask, int32_t H, int32_t W, int32_t kernel_size, void* out, bool sum) {
  if (!ctx || !mask || !out || H <= 0 || W <= 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "box_count: bad argument");
  if (kernel_size < 1 || kernel_size > 2 * kBoxMaxR + 1 || !(kernel_size & 1)) return rhccq_fail(ctx, RHCCQ_E_LIMIT, "box_count: odd kernel sizes 1..31");
  const int r = kernel_size / 2;
  const unsigned grid = (unsigned)(((W + kBoxTW - 1) / kBoxTW) * (long long)((H + kBoxTH - 1) / kBoxTH));
  if (sum) hipLaunchKernelGGL((box_count_kernel<true, uint32_t>), dim3(grid), dim3(256), 0, ctx->stream, mask, H, W, r, (uint32_t*)out);
  else hipLaunchKernelGGL((box_count_kernel<false, uint16_t>), dim3(grid), dim3(256), 0, ctx->stream, mask, H, W, r, (uint16_t*)out);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_box_count(rhccq_ctx* ctx, const uint8_t* mask, int32_t H, int32_t W, int32_t kernel_size, uint16_t* out) {
  return box_launch(ctx, mask, H, W, kernel_size, out, false);
}

int rhccq_box_sum(rhccq_ctx* ctx, const uint8_t* plane, int32_t H, int32_t W, int32_t kernel_size, uint32_t* out) {
  return box_launch(ctx, plane, H, W, kernel_size, out, true);
}

}  // extern "C"
