// Split score of a region: encoder/subregions/split_score.py:15-142 calculate_split_score -- colour
// complexity (Lab standard deviations + Sobel gradient of the Lab planes) and texture complexity (uniform LBP(8,1)
// entropy, Sobel-gradient variance, intensity entropy, intensity standard deviation) of the masked pixels.  The
// reference builds it from scikit-image (rgb2lab, rgb2gray, filters.sobel, feature.local_binary_pattern); those are
// restated here from their published definitions (PARITY UNPINNED: scikit-image is absent from the build container),
// all in float64 like the reference.
//
// One streaming pass: a 32 x 8 pixel tile and its 1-pixel halo are converted once (gray + Lab) and staged in LDS;
// every thread then evaluates its pixel's four Sobel magnitudes and its LBP code from LDS only and the workgroup
// reduces 12 masked sums (fixed tree) and two small histograms (integer LDS atomics).  3 B read per pixel (+ 1 B mask).
#include "rhccq_common.h"

namespace rhccq {

constexpr int kSsTW = 32, kSsTH = 8, kSsLW = kSsTW + 2, kSsLH = kSsTH + 2;
constexpr int kSsSums = 12, kSsHist = 10 + 32;

struct SsTile {
  double g[kSsLH][kSsLW], L[kSsLH][kSsLW], A[kSsLH][kSsLW], B[kSsLH][kSsLW];
};

__device__ __forceinline__ double ss_lin(double v) { return v > 0.04045 ? pow((v + 0.055) / 1.055, 2.4) : v / 12.92; }
__device__ __forceinline__ double ss_f(double t) { return t > 0.008856 ? cbrt(t) : 7.787 * t + 16.0 / 116.0; }

__device__ __forceinline__ void ss_convert(const uint8_t* px, double& g, double& L, double& A, double& B) {
  const double r8 = (double)px[0] / 255.0, g8 = (double)px[1] / 255.0, b8 = (double)px[2] / 255.0;
  g = (r8 * 0.2125 + g8 * 0.7154) + b8 * 0.0721;                      // skimage rgb2gray
  const double r = ss_lin(r8), gg = ss_lin(g8), b = ss_lin(b8);       // skimage rgb2xyz / xyz2lab (D65, 2 degree observer)
  const double X = ((0.412453 * r + 0.357580 * gg) + 0.180423 * b) / 0.95047;
  const double Y = ((0.212671 * r + 0.715160 * gg) + 0.072169 * b) / 1.0;
  const double Z = ((0.019334 * r + 0.119193 * gg) + 0.950227 * b) / 1.08883;
  const double fx = ss_f(X), fy = ss_f(Y), fz = ss_f(Z);
  L = 116.0 * fy - 16.0;
  A = 500.0 * (fx - fy);
  B = 200.0 * (fy - fz);
}

// skimage.filters.sobel: sqrt((h^2 + v^2) / 2) with [1,2,1]/4 smoothing (the tile holds 'reflect' borders)
__device__ __forceinline__ double ss_sobel(const double (*p)[kSsLW], int y, int x) {
  const double h = ((p[y + 1][x - 1] + 2.0 * p[y + 1][x] + p[y + 1][x + 1]) / 4.0) - ((p[y - 1][x - 1] + 2.0 * p[y - 1][x] + p[y - 1][x + 1]) / 4.0);
  const double v = ((p[y - 1][x + 1] + 2.0 * p[y][x + 1] + p[y + 1][x + 1]) / 4.0) - ((p[y - 1][x - 1] + 2.0 * p[y][x - 1] + p[y + 1][x - 1]) / 4.0);
  return sqrt((h * h + v * v) / 2.0);
}

__global__ __launch_bounds__(256) void split_stats_kernel(const uint8_t* __restrict__ rgb, int H, int W, const uint8_t* __restrict__ mask,
                                                          double* __restrict__ partial, int* __restrict__ hist) {
  __shared__ SsTile t;
  __shared__ int s_hist[kSsHist];
  __shared__ double s_red[4][kSsSums];
  const int tiles_x = (W + kSsTW - 1) / kSsTW;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x % tiles_x;
  const int y0 = ty * kSsTH, x0 = tx * kSsTW;
  if (threadIdx.x < kSsHist) s_hist[threadIdx.x] = 0;
  for (int i = threadIdx.x; i < kSsLW * kSsLH; i += 256) {
    const int ly = i / kSsLW, lx = i - ly * kSsLW;
    int y = y0 + ly - 1, x = x0 + lx - 1;
    y = y < 0 ? -y - 1 : (y >= H ? 2 * H - 1 - y : y);                 // 'reflect': d c b a | a b c d | d c b a
    x = x < 0 ? -x - 1 : (x >= W ? 2 * W - 1 - x : x);
    y = min(max(y, 0), H - 1);                                        // (tiles overhanging the image by more than one pixel)
    x = min(max(x, 0), W - 1);
    ss_convert(rgb + ((size_t)y * W + x) * 3, t.g[ly][lx], t.L[ly][lx], t.A[ly][lx], t.B[ly][lx]);
  }
  __syncthreads();
  const int ly = (threadIdx.x >> 5) + 1, lx = (threadIdx.x & 31) + 1;
  const int y = y0 + ly - 1, x = x0 + lx - 1;
  double v[kSsSums];
#pragma unroll
  for (int q = 0; q < kSsSums; ++q) v[q] = 0.0;
  if (y < H && x < W) {
    const double g = t.g[ly][lx];
    const bool in = mask ? mask[(size_t)y * W + x] != 0 : g > 0.01;
    if (in) {
      const double L = t.L[ly][lx], A = t.A[ly][lx], B = t.B[ly][lx];
      const double sL = ss_sobel(t.L, ly, lx), sA = ss_sobel(t.A, ly, lx), sB = ss_sobel(t.B, ly, lx), sg = ss_sobel(t.g, ly, lx);
      // the reference adds sqrt(grad_x^2 + grad_y^2) with the SAME filter for both (split_score.py:47-50)
      const double gm = (sqrt(sL * sL + sL * sL) + sqrt(sA * sA + sA * sA)) + sqrt(sB * sB + sB * sB);
      v[0] = 1.0; v[1] = L; v[2] = L * L; v[3] = A; v[4] = A * A; v[5] = B; v[6] = B * B; v[7] = gm; v[8] = sg; v[9] = sg * sg;
      v[10] = g; v[11] = g * g;
      // uniform LBP(8, 1): points p = 0..7 at (-sin, cos)(2 pi p / 8), coordinates rounded to 5 decimals; zeros outside the image
      const double d = 0.70711;
      auto at = [&](int dy, int dx) -> double {
        const int yy = y + dy, xx = x + dx;
        return (yy >= 0 && yy < H && xx >= 0 && xx < W) ? t.g[ly + dy][lx + dx] : 0.0;
      };
      auto diag = [&](int sy, int sx) -> double {                      // point (sy * d, sx * d), sy, sx = +-1
        // floor / ceil of the coordinates: (-1, 0) for a negative one, (0, 1) for a positive one
        const int r0 = sy < 0 ? -1 : 0, r1 = r0 + 1, c0 = sx < 0 ? -1 : 0, c1 = c0 + 1;
        const double dr = sy < 0 ? (-d) - (-1.0) : d, dc = sx < 0 ? (-d) - (-1.0) : d;
        const double top = (1 - dc) * at(r0, c0) + dc * at(r0, c1);
        const double bottom = (1 - dc) * at(r1, c0) + dc * at(r1, c1);
        return (1 - dr) * top + dr * bottom;
      };
      double pv[8];
      pv[0] = at(0, 1);       // p = 0: (-0, 1)
      pv[1] = diag(-1, 1);    // p = 1: (-0.70711, 0.70711)
      pv[2] = at(-1, 0);      // p = 2: (-1, 0)
      pv[3] = diag(-1, -1);
      pv[4] = at(0, -1);
      pv[5] = diag(1, -1);
      pv[6] = at(1, 0);
      pv[7] = diag(1, 1);
      int bits = 0, ones = 0;
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int b = (pv[p] - g >= 0.0) ? 1 : 0;
        bits |= b << p;
        ones += b;
      }
      const int rot = ((bits >> 1) | (bits << 7)) & 255;
      const int changes = __popc(bits ^ rot);
      atomicAdd(&s_hist[changes <= 2 ? ones : 9], 1);
      atomicAdd(&s_hist[10 + min((int)(g * 32.0), 31)], 1);
    }
  }
  // fixed reduction tree: lanes by shuffles, then the four waves in order
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < kSsSums; ++q) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[q] = v[q] + __shfl_down(v[q], o, 64);
    if (lane == 0) s_red[w][q] = v[q];
  }
  __syncthreads();
  if (threadIdx.x < kSsSums) {
    const int q = threadIdx.x;
    partial[(size_t)blockIdx.x * kSsSums + q] = ((s_red[0][q] + s_red[1][q]) + s_red[2][q]) + s_red[3][q];
  }
  if (threadIdx.x < kSsHist && s_hist[threadIdx.x]) atomicAdd(&hist[threadIdx.x], s_hist[threadIdx.x]);
}

}  // namespace rhccq

using namespace rhccq;

extern "C" {

int64_t rhccq_split_stats_blocks(int32_t H, int32_t W) {
  if (H <= 0 || W <= 0) return 0;
  return (int64_t)((H + kSsTH - 1) / kSsTH) * ((W + kSsTW - 1) / kSsTW);
}

int rhccq_split_stats(rhccq_ctx* ctx, const uint8_t* rgb, int32_t H, int32_t W, const uint8_t* mask, double* partial, int64_t n_blocks,
                      int32_t* hist42) {
  if (!ctx || !rgb || !partial || !hist42 || H <= 0 || W <= 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "split_stats: bad argument");
  if (n_blocks != rhccq_split_stats_blocks(H, W)) return rhccq_fail(ctx, RHCCQ_E_ARG, "split_stats: n_blocks mismatch");
  if (n_blocks > 0x7fffffffll) return rhccq_fail(ctx, RHCCQ_E_LIMIT, "split_stats: image too large");
  RHCCQ_HIP(ctx, hipMemsetAsync(hist42, 0, sizeof(int32_t) * kSsHist, ctx->stream));
  hipLaunchKernelGGL(split_stats_kernel, dim3((unsigned)n_blocks), dim3(256), 0, ctx->stream, rgb, H, W, mask, partial, hist42);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

}  // extern "C"
