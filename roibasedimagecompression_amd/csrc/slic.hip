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
    const int y = (int)(p / W), x = (int)(p - (long long)y * W);
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

}  // extern "C"
