// Binary-mask operators of the ROI clean-up chain (reference encoder/ROI/{roi,small_regions,small_gaps,thin_regions2}.py):
// what the reference takes from cv2.morphologyEx / dilate (rectangular and elliptical structuring elements), cv2.filter2D with
// one-directional kernels (small_gaps.py:221-319), cv2.distanceTransform(DIST_L2, 3) and cv2.Sobel on a 0/1 image.  PARITY UNPINNED
// (OpenCV is absent from the build container); restated from the published definitions, integer arithmetic throughout.
// Masks are u8 planes, "set" = non-zero; outputs are 0 / 255.
#include "rhccq_common.h"

namespace rhccq {

// BORDER_REFLECT_101 for any offset (several reflections on a narrow image): the pattern has period 2n - 2
__device__ __forceinline__ int m_reflect101(int i, int n) {
  if (n == 1) return 0;
  const int period = 2 * n - 2;
  i %= period;
  if (i < 0) i += period;
  return i < n ? i : period - i;
}

// ---- dilation by a structuring element given as one half-width per row (-1: the row is empty), rows dy = -r .. r; pixels outside
// the image are not set (cv2's default border value for dilate).  invert_in / invert_out turn it into the erosion by the same
// (symmetric) element with cv2's border rule for erode (outside = set): erode(A) = not dilate(not A).
// The general form (rhccq_morph_dilate_spans) gives every row dy = -up .. down its own span dx = -left .. right: what an
// even-sized OpenCV element needs (anchor k / 2: one column / row more before the anchor than behind it).
constexpr int kMorTW = 64, kMorTH = 16, kMorMaxR = 15;
struct MorphSpans { int r, up, down; signed char left[2 * kMorMaxR + 1], right[2 * kMorMaxR + 1]; };

__global__ __launch_bounds__(256) void morph_dilate_kernel(const uint8_t* __restrict__ in, int H, int W, MorphSpans se, int invert_in, int invert_out,
                                                           uint8_t* __restrict__ out) {
  __shared__ uint16_t pre[kMorTH + 2 * kMorMaxR][kMorTW + 2 * kMorMaxR + 2];   // per row: prefix count of set pixels
  const int r = se.r;
  const int tiles_x = (W + kMorTW - 1) / kMorTW;
  const int y0 = (blockIdx.x / tiles_x) * kMorTH, x0 = (blockIdx.x % tiles_x) * kMorTW;
  const int th = kMorTH + 2 * r, tw = kMorTW + 2 * r;
  for (int ly = threadIdx.x; ly < th; ly += 256) {          // one thread per staged row: a sequential prefix count
    const int y = y0 + ly - r;
    int acc = 0;
    pre[ly][0] = 0;
    for (int lx = 0; lx < tw; ++lx) {
      const int x = x0 + lx - r;
      bool v = false;
      if (y >= 0 && y < H && x >= 0 && x < W) v = (in[(long long)y * W + x] != 0) != (invert_in != 0);
      acc += v;
      pre[ly][lx + 1] = (uint16_t)acc;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kMorTH * kMorTW; i += 256) {
    const int ly = i / kMorTW, lx = i % kMorTW;
    const int y = y0 + ly, x = x0 + lx;
    if (y >= H || x >= W) continue;
    bool any = false;
    for (int dy = -se.up; dy <= se.down && !any; ++dy) {
      const int lf = se.left[dy + se.up], rt = se.right[dy + se.up];
      if (lf < 0) continue;
      const uint16_t* row = pre[ly + r + dy];
      any = row[lx + r + rt + 1] != row[lx + r - lf];
    }
    out[(long long)y * W + x] = (any != (invert_out != 0)) ? 255 : 0;
  }
}

// ---- compute_local_density (edges.py:173-195) for kernels up to 11 x 11 on ANY u8 plane: OpenCV's direct filter2D path, a float32
// accumulator over the taps in row-major order, every product and every sum rounded to float32 (the library is built with
// -ffp-contract=off), BORDER_REFLECT_101; `scale255`: the plane holds values above 1 and is divided by 255.0 first (float64 division,
// then float32, as `binary_map / 255.0` followed by astype(float32) does)
__global__ __launch_bounds__(256) void box_filter_seq_kernel(const uint8_t* __restrict__ plane, int H, int W, int k, int scale255, float* __restrict__ out) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= (long long)H * W) return;
  int y, x;
  rhccq_row_col(p, W, y, x);
  const int c = k / 2;
  const float term = 1.0f / (float)(k * k);
  float acc = 0.0f;
  for (int dy = 0; dy < k; ++dy) {
    const int yy = m_reflect101(y + dy - c, H);
    for (int dx = 0; dx < k; ++dx) {
      const int v = plane[(long long)yy * W + m_reflect101(x + dx - c, W)];
      const float f = scale255 ? (float)((double)v / 255.0) : (float)v;
      const float prod = term * f;
      acc = acc + prod;
    }
  }
  out[p] = acc;
}

// ---- element-wise combinations of masks: op 0: a | b, 1: a & b, 2: a & ~b, 3: ~a
__global__ __launch_bounds__(256) void mask_op_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, long long n, int op, uint8_t* __restrict__ out) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const bool x = a[p] != 0, y = b ? b[p] != 0 : false;
  const bool v = op == 0 ? (x || y) : op == 1 ? (x && y) : op == 2 ? (x && !y) : !x;
  out[p] = v ? 255 : 0;
}

// ---- bridge_small_gaps_fast (small_gaps.py:221-271): an unset pixel whose window count reaches `min_count` (its regional density
// exceeds the threshold) is set when, for one of the four direction pairs, BOTH opposite rays hold a set pixel within `reach` steps
// (the reference's one-directional filter2D kernels; coordinates beyond the image reflect, BORDER_REFLECT_101)
template <typename CntT>
__global__ __launch_bounds__(256) void gap_bridge_kernel(const uint8_t* __restrict__ in, const CntT* __restrict__ counts, int H, int W, long long min_count,
                                                         int reach, uint8_t* __restrict__ out) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= (long long)H * W) return;
  if (in[p] != 0) { out[p] = 255; return; }
  uint8_t v = 0;
  if ((long long)counts[p] >= min_count) {
    int y, x;
    rhccq_row_col(p, W, y, x);
    auto ray = [&](int dx, int dy) {
      for (int t = 1; t <= reach; ++t)
        if (in[(long long)m_reflect101(y + dy * t, H) * W + m_reflect101(x + dx * t, W)] != 0) return true;
      return false;
    };
    if ((ray(-1, 0) && ray(1, 0)) || (ray(0, -1) && ray(0, 1)) || (ray(-1, -1) && ray(1, 1)) || (ray(-1, 1) && ray(1, -1))) v = 255;
  }
  out[p] = v;
}

// ---- cv2.distanceTransform(mask, DIST_L2, 3): the 3x3 chamfer metric with OpenCV's fixed-point weights a = round(0.955 * 2^16),
// b = round(1.3693 * 2^16): distance to the nearest unset pixel = min over rows y' of chamfer(hz[y'][x], |y - y'|), where hz is
// the horizontal distance to the nearest unset pixel of row y' (the metric grows with |dx| for fixed |dy|, so the row's nearest
// unset pixel is its best); the row loop stops once a |dy| alone exceeds the best distance found.  The two-pass raster algorithm
// of OpenCV computes exactly this metric (paths never need to leave the image).  Output: fixed point, 16 fractional bits.
constexpr int kChamA = 62587, kChamB = 89738, kChamMax = 0x7fffffff >> 2;
constexpr int kHzNone = 0xffff;

// hz[y][x] = horizontal distance to the nearest unset pixel of row y (kHzNone: the row has none).  One workgroup per row: a thread owns
// a run of consecutive columns, notes its first / last unset column, the workgroup scans those (running max of "last" from the left,
// running min of "first" from the right), then every thread walks its columns once in each direction.
__global__ __launch_bounds__(256) void dist_hz_kernel(const uint8_t* __restrict__ mask, int H, int W, uint16_t* __restrict__ hz) {
  __shared__ int s_last[256], s_first[256];
  const int y = blockIdx.x, t = threadIdx.x;
  const uint8_t* row = mask + (long long)y * W;
  uint16_t* o = hz + (long long)y * W;
  const int cpt = (W + 255) / 256;
  const int x0 = min(W, t * cpt), x1 = min(W, x0 + cpt);
  const int kFar = 0x3fffffff;
  int last = -kFar, first = kFar;
  for (int x = x0; x < x1; ++x)
    if (row[x] == 0) { if (first == kFar) first = x; last = x; }
  s_last[t] = last;
  s_first[t] = first;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {                      // inclusive scans: max of last over threads <= t, min of first over threads >= t
    const int a = t >= d ? s_last[t - d] : -kFar, b = t + d < 256 ? s_first[t + d] : kFar;
    __syncthreads();
    s_last[t] = max(s_last[t], a);
    s_first[t] = min(s_first[t], b);
    __syncthreads();
  }
  int left = t > 0 ? s_last[t - 1] : -kFar;                 // nearest unset column before this thread's run
  const int right0 = t < 255 ? s_first[t + 1] : kFar;      // ... behind it
  for (int x = x0; x < x1; ++x) {
    if (row[x] == 0) left = x;
    const int d = left == -kFar ? kHzNone : x - left;
    o[x] = (uint16_t)(d >= kHzNone - 1 ? kHzNone : d);
  }
  int right = right0;
  for (int x = x1 - 1; x >= x0; --x) {
    if (row[x] == 0) right = x;
    int d = right == kFar ? kHzNone : right - x;
    if (d >= kHzNone - 1) d = kHzNone;
    if (d < (int)o[x]) o[x] = (uint16_t)d;
  }
}

__global__ __launch_bounds__(256) void dist_chamfer_kernel(const uint16_t* __restrict__ hz, int H, int W, int32_t* __restrict__ dist) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= (long long)H * W) return;
  int y, x;
  rhccq_row_col(p, W, y, x);
  const int h0 = hz[p];
  if (h0 == 0) { dist[p] = 0; return; }
  long long best = h0 == kHzNone ? (long long)kChamMax : (long long)h0 * kChamA;
  for (int dy = 1; (long long)dy * kChamA < best; ++dy) {
    bool any_row = false;
#pragma unroll
    for (int sgn = -1; sgn <= 1; sgn += 2) {
      const int yy = y + sgn * dy;
      if (yy < 0 || yy >= H) continue;
      any_row = true;
      const int dx = hz[(long long)yy * W + x];
      if (dx == kHzNone) continue;
      const int mn = dx < dy ? dx : dy, mx = dx < dy ? dy : dx;
      const long long c = (long long)mn * kChamB + (long long)(mx - mn) * kChamA;
      if (c < best) best = c;
    }
    if (!any_row) break;
  }
  dist[p] = (int32_t)(best > kChamMax ? kChamMax : best);
}

// ---- detect_meaningful_borders (roi.py:784-822), first half: squared 3x3 Sobel magnitude of the 0/1 image (BORDER_REFLECT_101),
// a value in 0..32, and its maximum over the image
__global__ __launch_bounds__(256) void binary_sobel_kernel(const uint8_t* __restrict__ mask, int H, int W, uint8_t* __restrict__ m2, int32_t* max_out) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  int v = 0;
  if (p < (long long)H * W) {
    int y, x;
    rhccq_row_col(p, W, y, x);
    const int ym = m_reflect101(y - 1, H), yp = m_reflect101(y + 1, H), xm = m_reflect101(x - 1, W), xp = m_reflect101(x + 1, W);
    auto at = [&](int yy, int xx) { return mask[(long long)yy * W + xx] != 0 ? 1 : 0; };
    const int a = at(ym, xm), b = at(ym, x), c = at(ym, xp), d = at(y, xm), f = at(y, xp), g = at(yp, xm), h = at(yp, x), i = at(yp, xp);
    const int gx = (c + 2 * f + i) - (a + 2 * d + g), gy = (g + 2 * h + i) - (a + 2 * b + c);
    v = gx * gx + gy * gy;
    m2[p] = (uint8_t)v;
  }
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_down(v, o, 64));
  if ((threadIdx.x & 63) == 0 && v > 0) atomicMax(max_out, v);
}

__global__ __launch_bounds__(256) void lut_u8_kernel(const uint8_t* __restrict__ in, const uint8_t* __restrict__ lut, long long n, uint8_t* __restrict__ out) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p < n) out[p] = lut[in[p]];
}

// out[p] = table[values[p]] (float32 table indexed by a u16 plane: window counts -> OpenCV's float32 densities)
__global__ __launch_bounds__(256) void lut_u16_f32_kernel(const uint16_t* __restrict__ values, const float* __restrict__ table, int n_table, long long n,
                                                          float* __restrict__ out) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p < n) out[p] = table[min((int)values[p], n_table - 1)];
}

// ---- per label: sum of a u16 plane (box counts) or of an int32 plane (fixed-point distances), int64 accumulators.  A workgroup
// walks a contiguous chunk of pixels and gathers its sums in an LDS table keyed by label (the background and the large
// components would otherwise serialise a hundred thousand atomics on one address: 69 ms at 4K); only the occupied slots reach
// global memory
constexpr int kLsSlots = 256, kLsEmpty = -1;
template <typename T>
__global__ __launch_bounds__(256) void label_sum_kernel(const int32_t* __restrict__ labels, const T* __restrict__ val, long long n, long long chunk,
                                                        unsigned long long* sums) {
  __shared__ int s_key[kLsSlots];
  __shared__ unsigned long long s_sum[kLsSlots];
  s_key[threadIdx.x] = kLsEmpty;
  s_sum[threadIdx.x] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const long long begin = (long long)blockIdx.x * chunk, end = min(n, begin + chunk);
  for (long long base = begin; base < end; base += 256) {
    const long long p = base + threadIdx.x;
    const int l = p < end ? labels[p] : -1;
    const unsigned long long v = l >= 0 ? (unsigned long long)val[p] : 0ull;
    unsigned long long todo = __ballot(l >= 0);
    while (todo) {
      const int first = __builtin_ctzll(todo);
      const int lead = __shfl(l, first);
      const unsigned long long same = __ballot(l == lead);
      todo &= ~same;
      unsigned long long part = l == lead ? v : 0ull;
      for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o, 64);
      if (lane == 0 && part) {
        int slot = (int)(((unsigned)lead * 2654435761u) >> 24);
        bool done = false;
        for (int probe = 0; probe < 8 && !done; ++probe, slot = (slot + 1) & (kLsSlots - 1)) {
          const int old = atomicCAS(&s_key[slot], kLsEmpty, lead);
          if (old == kLsEmpty || old == lead) { atomicAdd(&s_sum[slot], part); done = true; }
        }
        if (!done) atomicAdd(&sums[lead], part);
      }
    }
  }
  __syncthreads();
  if (s_key[threadIdx.x] != kLsEmpty && s_sum[threadIdx.x]) atomicAdd(&sums[s_key[threadIdx.x]], s_sum[threadIdx.x]);
}

// ---- histogram (int64[n_bins]) of a u16 plane over the pixels where `mask` is set; out = 255 where values[p] >= min_value and
// (mask is NULL or set)
__global__ __launch_bounds__(256) void masked_hist_kernel(const uint8_t* __restrict__ mask, const uint16_t* __restrict__ values, long long n, int n_bins,
                                                          unsigned long long* hist) {
  extern __shared__ unsigned int s_h[];
  for (int i = threadIdx.x; i < n_bins; i += 256) s_h[i] = 0;
  __syncthreads();
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < n; p += (long long)gridDim.x * 256)
    if (mask[p] != 0 && values[p] < n_bins) atomicAdd(&s_h[values[p]], 1u);
  __syncthreads();
  for (int i = threadIdx.x; i < n_bins; i += 256)
    if (s_h[i]) atomicAdd(&hist[i], (unsigned long long)s_h[i]);
}

__global__ __launch_bounds__(256) void value_mask_kernel(const uint8_t* __restrict__ mask, const uint16_t* __restrict__ values, long long n, int min_value,
                                                         uint8_t* __restrict__ out) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p < n) out[p] = ((int)values[p] >= min_value && (!mask || mask[p] != 0)) ? 255 : 0;
}

}  // namespace rhccq

using namespace rhccq;

extern "C" {

int rhccq_morph_dilate_spans(rhccq_ctx* ctx, const uint8_t* in, int32_t H, int32_t W, int32_t up, int32_t down, const int32_t* left,
                             const int32_t* right, int32_t invert_in, int32_t invert_out, uint8_t* out) {
  if (!ctx || !in || !out || !left || !right || H <= 0 || W <= 0 || in == out) return rhccq_fail(ctx, RHCCQ_E_ARG, "morph_dilate: bad argument");
  if (up < 0 || down < 0 || up > kMorMaxR || down > kMorMaxR) return rhccq_fail(ctx, RHCCQ_E_LIMIT, "morph_dilate: structuring elements up to 31 x 31");
  MorphSpans se;
  se.up = up;
  se.down = down;
  se.r = up > down ? up : down;
  for (int i = 0; i <= up + down; ++i) {
    if (left[i] > kMorMaxR || right[i] > kMorMaxR) return rhccq_fail(ctx, RHCCQ_E_LIMIT, "morph_dilate: structuring elements up to 31 x 31");
    if ((left[i] < 0) != (right[i] < 0)) return rhccq_fail(ctx, RHCCQ_E_ARG, "morph_dilate: a row is empty on one side only");
    se.left[i] = (signed char)(left[i] < 0 ? -1 : left[i]);
    se.right[i] = (signed char)(right[i] < 0 ? -1 : right[i]);
    if (left[i] > se.r) se.r = left[i];
    if (right[i] > se.r) se.r = right[i];
  }
  const unsigned grid = (unsigned)(((W + kMorTW - 1) / kMorTW) * (long long)((H + kMorTH - 1) / kMorTH));
  hipLaunchKernelGGL(morph_dilate_kernel, dim3(grid), dim3(256), 0, ctx->stream, in, H, W, se, invert_in, invert_out, out);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_morph_dilate(rhccq_ctx* ctx, const uint8_t* in, int32_t H, int32_t W, int32_t radius, const int32_t* half_widths, int32_t invert_in,
                       int32_t invert_out, uint8_t* out) {
  if (!half_widths || radius < 0 || radius > kMorMaxR) return rhccq_fail(ctx, RHCCQ_E_LIMIT, "morph_dilate: structuring elements up to 31 x 31");
  for (int i = 0; i <= 2 * radius; ++i)
    if (half_widths[i] > radius) return rhccq_fail(ctx, RHCCQ_E_ARG, "morph_dilate: a half-width exceeds the radius");
  return rhccq_morph_dilate_spans(ctx, in, H, W, radius, radius, half_widths, half_widths, invert_in, invert_out, out);
}

int rhccq_box_filter_seq(rhccq_ctx* ctx, const uint8_t* plane, int32_t H, int32_t W, int32_t kernel_size, int32_t scale255, float* out) {
  if (!ctx || !plane || !out || H <= 0 || W <= 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "box_filter_seq: bad argument");
  if (kernel_size < 1 || kernel_size > 11 || (kernel_size & 1) == 0) return rhccq_fail(ctx, RHCCQ_E_LIMIT, "box_filter_seq: odd kernels up to 11 x 11 (OpenCV's direct path)");
  hipLaunchKernelGGL(box_filter_seq_kernel, dim3((unsigned)(((long long)H * W + 255) / 256)), dim3(256), 0, ctx->stream, plane, H, W, kernel_size,
                     scale255, out);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_mask_op(rhccq_ctx* ctx, const uint8_t* a, const uint8_t* b, int64_t n, int32_t op, uint8_t* out) {
  if (!ctx || !a || !out || n <= 0 || op < 0 || op > 3 || (op != 3 && !b)) return rhccq_fail(ctx, RHCCQ_E_ARG, "mask_op: bad argument");
  hipLaunchKernelGGL(mask_op_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, a, b, (long long)n, op, out);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_gap_bridge(rhccq_ctx* ctx, const uint8_t* in, const void* counts, int32_t count_bytes, int32_t H, int32_t W, int64_t min_count, int32_t reach,
                     uint8_t* out) {
  if (!ctx || !in || !counts || !out || H <= 0 || W <= 0 || reach < 0 || in == out || (count_bytes != 2 && count_bytes != 4))
    return rhccq_fail(ctx, RHCCQ_E_ARG, "gap_bridge: bad argument");
  const unsigned grid = (unsigned)(((long long)H * W + 255) / 256);
  if (count_bytes == 2)
    hipLaunchKernelGGL(gap_bridge_kernel<uint16_t>, dim3(grid), dim3(256), 0, ctx->stream, in, (const uint16_t*)counts, H, W, (long long)min_count, reach, out);
  else
    hipLaunchKernelGGL(gap_bridge_kernel<uint32_t>, dim3(grid), dim3(256), 0, ctx->stream, in, (const uint32_t*)counts, H, W, (long long)min_count, reach, out);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_dist_chamfer(rhccq_ctx* ctx, const uint8_t* mask, int32_t H, int32_t W, uint16_t* hz_tmp, int32_t* dist) {
  if (!ctx || !mask || !hz_tmp || !dist || H <= 0 || W <= 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "dist_chamfer: bad argument");
  if (W >= kHzNone) return rhccq_fail(ctx, RHCCQ_E_LIMIT, "dist_chamfer: rows up to 65534 pixels");
  hipLaunchKernelGGL(dist_hz_kernel, dim3((unsigned)H), dim3(256), 0, ctx->stream, mask, H, W, hz_tmp);
  hipLaunchKernelGGL(dist_chamfer_kernel, dim3((unsigned)(((long long)H * W + 255) / 256)), dim3(256), 0, ctx->stream, hz_tmp, H, W, dist);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_binary_sobel(rhccq_ctx* ctx, const uint8_t* mask, int32_t H, int32_t W, uint8_t* m2, int32_t* max_out) {
  if (!ctx || !mask || !m2 || !max_out || H <= 0 || W <= 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "binary_sobel: bad argument");
  RHCCQ_HIP(ctx, hipMemsetAsync(max_out, 0, sizeof(int32_t), ctx->stream));
  hipLaunchKernelGGL(binary_sobel_kernel, dim3((unsigned)(((long long)H * W + 255) / 256)), dim3(256), 0, ctx->stream, mask, H, W, m2, max_out);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_lut_u8(rhccq_ctx* ctx, const uint8_t* in, const uint8_t* lut256, int64_t n, uint8_t* out) {
  if (!ctx || !in || !lut256 || !out || n <= 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "lut_u8: bad argument");
  hipLaunchKernelGGL(lut_u8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, in, lut256, (long long)n, out);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_lut_u16_f32(rhccq_ctx* ctx, const uint16_t* values, const float* table, int32_t n_table, int64_t n, float* out) {
  if (!ctx || !values || !table || !out || n <= 0 || n_table <= 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "lut_u16_f32: bad argument");
  hipLaunchKernelGGL(lut_u16_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, values, table, n_table, (long long)n, out);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_label_sum(rhccq_ctx* ctx, const int32_t* labels, const void* values, int32_t value_bytes, int64_t n_pixels, int32_t n_labels, uint64_t* sums) {
  if (!ctx || !labels || !values || !sums || n_pixels <= 0 || n_labels < 0 || (value_bytes != 2 && value_bytes != 4))
    return rhccq_fail(ctx, RHCCQ_E_ARG, "label_sum: bad argument");
  RHCCQ_HIP(ctx, hipMemsetAsync(sums, 0, sizeof(uint64_t) * ((size_t)n_labels + 1), ctx->stream));
  long long chunk = (((n_pixels + 2047) / 2048) + 255) & ~255ll;        // <= 2048 workgroups, whole 256-pixel rounds each
  if (chunk < 4096) chunk = 4096;
  const unsigned grid = (unsigned)((n_pixels + chunk - 1) / chunk);
  if (value_bytes == 2)
    hipLaunchKernelGGL(label_sum_kernel<uint16_t>, dim3(grid), dim3(256), 0, ctx->stream, labels, (const uint16_t*)values, (long long)n_pixels, chunk,
                       (unsigned long long*)sums);
  else
    hipLaunchKernelGGL(label_sum_kernel<int32_t>, dim3(grid), dim3(256), 0, ctx->stream, labels, (const int32_t*)values, (long long)n_pixels, chunk,
                       (unsigned long long*)sums);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_masked_hist(rhccq_ctx* ctx, const uint8_t* mask, const uint16_t* values, int64_t n_pixels, int32_t n_bins, uint64_t* hist) {
  if (!ctx || !mask || !values || !hist || n_pixels <= 0 || n_bins <= 0 || n_bins > 4096) return rhccq_fail(ctx, RHCCQ_E_ARG, "masked_hist: bad argument");
  RHCCQ_HIP(ctx, hipMemsetAsync(hist, 0, sizeof(uint64_t) * (size_t)n_bins, ctx->stream));
  const unsigned grid = (unsigned)((n_pixels + 255) / 256 < 1024 ? (n_pixels + 255) / 256 : 1024);
  hipLaunchKernelGGL(masked_hist_kernel, dim3(grid), dim3(256), (size_t)n_bins * 4, ctx->stream, mask, values, (long long)n_pixels, n_bins, (unsigned long long*)hist);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_value_mask(rhccq_ctx* ctx, const uint8_t* mask, const uint16_t* values, int64_t n_pixels, int32_t min_value, uint8_t* out) {
  if (!ctx || !values || !out || n_pixels <= 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "value_mask: bad argument");
  hipLaunchKernelGGL(value_mask_kernel, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, ctx->stream, mask, values, (long long)n_pixels, min_value, out);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

}  // extern "C"
