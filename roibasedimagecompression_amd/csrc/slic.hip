// Masked SLIC superpixels: encoder/subregions/slic.py:41-104 enhanced_slic_with_texture ->
// skimage.segmentation.slic(image, n_segments, compactness=10, sigma=1, mask=...).  PARITY UNPINNED (scikit-image is absent
// from the build container): restated from its published algorithm (_slic_cython, _enforce_label_connectivity_cython).
//
//   slic_assign_kernel   one assignment sweep: every masked pixel takes the centroid with the smallest
//                        spatial / step^2 (+ colour) distance among those whose window [c - 2 step, c + 2 step] holds it,
//                        the first one on ties (float64, the same operations in the same order as the restatement, so
//                        the label maps are identical);
//   rhccq_slic_connectivity_host   the sequential connectivity enforcement (raster scan + breadth-first floods) as a
//                        native HOST routine: it is an inherently serial scan over <= 500 x 500 labels.
// The image SLIC sees is the reference's <= 500-pixel downscale, so this stage is small by construction; the centroid
// seeding (RandomState(123) + scipy kmeans2), the Gaussian and the centroid means (numpy bincount = raster-order sums,
// what skimage's loop computes) stay on the host (api/slic.py).
#include <vector>

#include "rhccq_common.h"

namespace rhccq {

__global__ __launch_bounds__(256) void slic_assign_kernel(const double* __restrict__ img /* [H][W][3], scaled by 1/compactness */,
                                                          const uint8_t* __restrict__ mask, const double* __restrict__ seg /* [K][5] = y, x, c0, c1, c2 */,
                                                          int H, int W, int K, double step, int ignore_color, int32_t* __restrict__ labels) {
  extern __shared__ double s_seg[];                      // [K][5] then the windows int[K][4]
  int* s_win = reinterpret_cast<int*>(s_seg + (size_t)K * 5);
  for (int i = threadIdx.x; i < K * 5; i += 256) s_seg[i] = seg[i];
  for (int k = threadIdx.x; k < K; k += 256) {
    const double cy = seg[k * 5], cx = seg[k * 5 + 1];
    s_win[k * 4 + 0] = (int)fmax(cy - 2 * step, 0.0);
    s_win[k * 4 + 1] = (int)fmin(cy + 2 * step + 1, (double)H);
    s_win[k * 4 + 2] = (int)fmax(cx - 2 * step, 0.0);
    s_win[k * 4 + 3] = (int)fmin(cx + 2 * step + 1, (double)W);
  }
  __syncthreads();
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= (long long)H * W) return;
  int best = 0;
  if (mask[p]) {
    int y, x;
    rhccq_row_col(p, W, y, x);
    const double inv = 1.0 / (step * step);
    const double i0 = img[p * 3], i1 = img[p * 3 + 1], i2 = img[p * 3 + 2];
    double bd = 1.7976931348623157e308;
    for (int k = 0; k < K; ++k) {
      if (y < s_win[k * 4] || y >= s_win[k * 4 + 1] || x < s_win[k * 4 + 2] || x >= s_win[k * 4 + 3]) continue;
      const double dy = s_seg[k * 5] - (double)y, dx = s_seg[k * 5 + 1] - (double)x;
      double d = (dy * dy + dx * dx) * inv;
      if (!ignore_color) {
        const double e0 = i0 - s_seg[k * 5 + 2], e1 = i1 - s_seg[k * 5 + 3], e2 = i2 - s_seg[k * 5 + 4];
        double dc = 0.0;
        dc = dc + e0 * e0;
        dc = dc + e1 * e1;
        dc = dc + e2 * e2;
        d = d + dc;
      }
      if (bd > d) { bd = d; best = k + 1; }
    }
  }
  labels[p] = best;
}

// ---- skimage.transform.resize of enhanced_slic_with_texture (slic.py:42-44,82,101): scipy.ndimage.gaussian_filter (anti-aliasing) and
// scipy.ndimage.zoom(order 0 / 1, mode 'mirror', grid_mode) restated operation for operation (float64, no contraction):
//   correlate1d with a symmetric kernel: t = x[c] w[0]; for j = R .. 1: t += (x[c - j] + x[c + j]) w[j]   (outermost pair first)
//   zoom, order 1: t = 0; for (dy, dx) in (0,0), (0,1), (1,0), (1,1): t += (x[y_dy][x_dx] * wy[dy]) * wx[dx]
// Coordinates, weights and mirrored indices are tiny per-axis tables computed on the host exactly as scipy computes them.
__global__ __launch_bounds__(256) void gauss1d_kernel(const double* __restrict__ in, long long outer, int len, long long inner,
                                                      const double* __restrict__ w /* [R + 1]: centre, then offsets 1 .. R */, int R,
                                                      double* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= outer * len * inner) return;
  const long long k = i % inner, c = (i / inner) % len, o = i / (inner * len);
  const double* base = in + o * len * inner + k;
  auto at = [&](long long p) {                             // mode 'mirror': reflect about the centre of the edge samples
    if (len == 1) return base[0];
    const long long period = 2ll * len - 2;
    p %= period;
    if (p < 0) p += period;
    if (p >= len) p = period - p;
    return base[p * inner];
  };
  double t = at(c) * w[0];
  for (int j = R; j >= 1; --j) t += (at(c - j) + at(c + j)) * w[j];
  out[i] = t;
}

__global__ __launch_bounds__(256) void zoom_linear_kernel(const double* __restrict__ in, int W, int C, const int32_t* __restrict__ yi /* [2][oh] */,
                                                          const double* __restrict__ wy /* [2][oh] */, const int32_t* __restrict__ xi /* [2][ow] */,
                                                          const double* __restrict__ wx, int oh, int ow, double lo, double hi, double* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)oh * ow * C) return;
  const int ch = (int)(i % C), x = (int)((i / C) % ow), y = (int)(i / ((long long)C * ow));
  double t = 0.0;
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      double c = in[((long long)yi[dy * oh + y] * W + xi[dx * ow + x]) * C + ch];
      c *= wy[dy * oh + y];
      c *= wx[dx * ow + x];
      t += c;
    }
  out[i] = t < lo ? lo : (t > hi ? hi : t);                // resize(..., preserve_range) clips to the input's range
}

template <typename T>
__global__ __launch_bounds__(256) void zoom_nearest_kernel(const T* __restrict__ in, int W, int C, const int32_t* __restrict__ yi, const int32_t* __restrict__ xi,
                                                           int oh, int ow, T* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)oh * ow * C) return;
  const int ch = (int)(i % C), x = (int)((i / C) % ow), y = (int)(i / ((long long)C * ow));
  out[i] = in[((long long)yi[y] * W + xi[x]) * C + ch];
}

}  // namespace rhccq

using namespace rhccq;

extern "C" {

int rhccq_slic_assign(rhccq_ctx* ctx, const double* img, const uint8_t* mask, const double* seg, int32_t H, int32_t W, int32_t K, double step,
                      int32_t ignore_color, int32_t* labels) {
  if (!ctx || !img || !mask || !seg || !labels || H <= 0 || W <= 0 || K <= 0 || !(step > 0.0)) return rhccq_fail(ctx, RHCCQ_E_ARG, "slic_assign: bad argument");
  const size_t lds = (size_t)K * (5 * sizeof(double) + 4 * sizeof(int));
  if (lds > 60 * 1024) return rhccq_fail(ctx, RHCCQ_E_LIMIT, "slic_assign: more than ~1000 centroids");
  const unsigned grid = (unsigned)(((long long)H * W + 255) / 256);
  hipLaunchKernelGGL(slic_assign_kernel, dim3(grid), dim3(256), lds, ctx->stream, img, mask, seg, H, W, K, step, ignore_color, labels);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

// HOST routine (plain pointers to host memory): skimage _enforce_label_connectivity_cython for a 2-D label map,
// start_label 1, 0 = outside the mask
int rhccq_slic_connectivity_host(const int32_t* labels_host, int32_t H, int32_t W, int32_t min_size, int32_t max_size, int32_t* out_host) {
  if (!labels_host || !out_host || H <= 0 || W <= 0 || max_size < 1) return RHCCQ_E_ARG;
  const long long n = (long long)H * W;
  for (long long i = 0; i < n; ++i) out_host[i] = 0;
  std::vector<int32_t> comp((size_t)max_size + 4);
  const int ddy[4] = {0, 0, 1, -1}, ddx[4] = {1, -1, 0, 0};
  int32_t cur = 1;
  for (int y = 0; y < H; ++y) {
    for (int x = 0; x < W; ++x) {
      const long long p = (long long)y * W + x;
      if (out_host[p] >= 1 || labels_host[p] == 0) continue;
      int32_t adjacent = 0;
      const int32_t label = labels_host[p];
      out_host[p] = cur;
      int size = 1, visited = 0;
      comp[0] = (int32_t)p;
      while (visited < size && size < max_size) {
        const int cy = comp[visited] / W, cx = comp[visited] % W;
        for (int i = 0; i < 4; ++i) {
          const int ny = cy + ddy[i], nx = cx + ddx[i];
          if (ny < 0 || ny >= H || nx < 0 || nx >= W) continue;
          const long long q = (long long)ny * W + nx;
          if (labels_host[q] == label && out_host[q] == 0) {
            out_host[q] = cur;
            comp[size++] = (int32_t)q;
            if (size >= max_size) break;
          } else if (out_host[q] >= 1 && out_host[q] != cur) {
            adjacent = out_host[q];
          }
        }
        ++visited;
      }
      if (size < min_size) {
        for (int i = 0; i < size; ++i) out_host[comp[i]] = adjacent;
      } else {
        ++cur;
      }
    }
  }
  return 0;
}

int rhccq_gauss1d_f64(rhccq_ctx* ctx, const double* in, int64_t outer, int32_t len, int64_t inner, const double* weights, int32_t radius, double* out) {
  if (!ctx || !in || !weights || !out || in == out || outer <= 0 || len <= 0 || inner <= 0 || radius < 0)
    return rhccq_fail(ctx, RHCCQ_E_ARG, "gauss1d_f64: bad argument");
  const long long n = (long long)outer * len * inner;
  hipLaunchKernelGGL(gauss1d_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, in, (long long)outer, len, (long long)inner, weights, radius, out);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_zoom_linear_f64(rhccq_ctx* ctx, const double* in, int32_t H, int32_t W, int32_t C, const int32_t* yi, const double* wy, const int32_t* xi,
                          const double* wx, int32_t oh, int32_t ow, double lo, double hi, double* out) {
  if (!ctx || !in || !yi || !wy || !xi || !wx || !out || H <= 0 || W <= 0 || C <= 0 || oh <= 0 || ow <= 0)
    return rhccq_fail(ctx, RHCCQ_E_ARG, "zoom_linear_f64: bad argument");
  const long long n = (long long)oh * ow * C;
  hipLaunchKernelGGL(zoom_linear_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, in, W, C, yi, wy, xi, wx, oh, ow, lo, hi, out);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_zoom_nearest(rhccq_ctx* ctx, const void* in, int32_t elem_bytes, int32_t H, int32_t W, int32_t C, const int32_t* yi, const int32_t* xi, int32_t oh,
                       int32_t ow, void* out) {
  if (!ctx || !in || !yi || !xi || !out || H <= 0 || W <= 0 || C <= 0 || oh <= 0 || ow <= 0 || (elem_bytes != 1 && elem_bytes != 4))
    return rhccq_fail(ctx, RHCCQ_E_ARG, "zoom_nearest: bad argument");
  const long long n = (long long)oh * ow * C;
  const unsigned grid = (unsigned)((n + 255) / 256);
  if (elem_bytes == 1) hipLaunchKernelGGL(zoom_nearest_kernel<uint8_t>, dim3(grid), dim3(256), 0, ctx->stream, (const uint8_t*)in, W, C, yi, xi, oh, ow, (uint8_t*)out);
  else hipLaunchKernelGGL(zoom_nearest_kernel<int32_t>, dim3(grid), dim3(256), 0, ctx->stream, (const int32_t*)in, W, C, yi, xi, oh, ow, (int32_t*)out);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

}  // extern "C"
