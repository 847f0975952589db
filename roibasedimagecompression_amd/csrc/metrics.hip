// Quality metrics of the decoder side: decoder/uncompression/comparison.py:30-80
// calculate_quality_metrics -- PSNR / MSE / RMSE / MAE / max error / per-channel MSE from exact integer error
// sums, and SSIM as skimage.metrics.structural_similarity(data_range=255, channel_axis=2, win_size=7) defines it
// (uniform 7x7 window, sample covariance, mean over the interior that a full window fits).
//
// Both kernels stream the two uint8 RGB images once.  The error sums are integers (order independent); the SSIM
// window statistics are integer sums over the 49 pixels (exact), only the final ratio is float64, and every
// workgroup writes its own partial sum so that the host adds them in a fixed order.
#include "rhccq_common.h"

namespace rhccq {

// ---- error sums: sums[c] = sum (a-b)^2 of channel c, sums[3] = sum |a-b|, sums[4] = max |a-b| -------------
__global__ __launch_bounds__(256) void error_sums_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, long long n_px,
                                                         unsigned long long* __restrict__ sums) {
  __shared__ unsigned long long red[5][4];
  unsigned long long sq[3] = {0, 0, 0}, ab = 0;
  unsigned mx = 0;
  // 4 pixels = 12 bytes = 3 dwords per lane and iteration
  const long long n4 = n_px >> 2;
  const uint32_t* a4 = reinterpret_cast<const uint32_t*>(a);
  const uint32_t* b4 = reinterpret_cast<const uint32_t*>(b);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    uint32_t wa[3], wb[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { wa[j] = a4[i * 3 + j]; wb[j] = b4[i * 3 + j]; }
#pragma unroll
    for (int j = 0; j < 12; ++j) {                        // byte j of the 12: channel j % 3
      const int va = (wa[j >> 2] >> ((j & 3) * 8)) & 255, vb = (wb[j >> 2] >> ((j & 3) * 8)) & 255;
      const unsigned d = (unsigned)abs(va - vb);
      sq[j % 3] += d * d;
      ab += d;
      mx = max(mx, d);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n_px & 3)) {       // tail pixels
    const long long p = (n4 << 2) + threadIdx.x;
    for (int c = 0; c < 3; ++c) {
      const unsigned d = (unsigned)abs((int)a[p * 3 + c] - (int)b[p * 3 + c]);
      sq[c] += d * d;
      ab += d;
      mx = max(mx, d);
    }
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned long long v[5] = {sq[0], sq[1], sq[2], ab, (unsigned long long)mx};
#pragma unroll
  for (int q = 0; q < 5; ++q) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long t = __shfl_down(v[q], o, 64);
      v[q] = q == 4 ? (t > v[q] ? t : v[q]) : v[q] + t;
    }
    if (lane == 0) red[q][w] = v[q];
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    const int q = threadIdx.x;
    unsigned long long t = red[q][0];
    for (int i = 1; i < 4; ++i) t = q == 4 ? (red[q][i] > t ? red[q][i] : t) : t + red[q][i];
    if (q == 4) atomicMax(&sums[4], t); else atomicAdd(&sums[q], t);   // 5 atomics per workgroup
  }
}

// ---- SSIM, 7x7 uniform window -----------------------------------------------------------------------------
constexpr int kSsimTile = 32, kSsimWin = 7, kSsimPad = 3, kSsimIn = kSsimTile + kSsimWin - 1;   // 38

// one workgroup: a 32x32 tile of window centres (interior coordinates), staged with its 3-pixel apron in LDS
__global__ __launch_bounds__(256) void ssim7_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, int H, int W,
                                                    double* __restrict__ partial /* [gridDim.y * gridDim.x][3] */) {
  __shared__ uint8_t sa[3][kSsimIn][kSsimIn + 2], sb[3][kSsimIn][kSsimIn + 2];
  __shared__ double red[3][4];
  const int oy0 = blockIdx.y * kSsimTile, ox0 = blockIdx.x * kSsimTile;   // interior coordinates: centre = (+3, +3)
  const int IH = H - 2 * kSsimPad, IW = W - 2 * kSsimPad;
  for (int i = threadIdx.x; i < kSsimIn * kSsimIn; i += 256) {
    const int r = i / kSsimIn, c = i % kSsimIn;
    const int y = min(oy0 + r, H - 1), x = min(ox0 + c, W - 1);           // clamped reads feed only discarded outputs
    const long long p = ((long long)y * W + x) * 3;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) { sa[ch][r][c] = a[p + ch]; sb[ch][r][c] = b[p + ch]; }
  }
  __syncthreads();
  const double C1 = (0.01 * 255.0) * (0.01 * 255.0), C2 = (0.03 * 255.0) * (0.03 * 255.0);
  const double cov_norm = 49.0 / 48.0;
  double acc[3] = {0.0, 0.0, 0.0};
  for (int o = threadIdx.x; o < kSsimTile * kSsimTile; o += 256) {
    const int r = o / kSsimTile, c = o % kSsimTile;
    if (oy0 + r >= IH || ox0 + c >= IW) continue;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      int sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;     // 49 * 65025 < 2^22
      for (int dy = 0; dy < kSsimWin; ++dy)
#pragma unroll
        for (int dx = 0; dx < kSsimWin; ++dx) {
          const int x = sa[ch][r + dy][c + dx], y = sb[ch][r + dy][c + dx];
          sx += x; sy += y; sxx += x * x; syy += y * y; sxy += x * y;
        }
      const double ux = (double)sx / 49.0, uy = (double)sy / 49.0;
      const double uxx = (double)sxx / 49.0, uyy = (double)syy / 49.0, uxy = (double)sxy / 49.0;
      const double vx = cov_norm * (uxx - ux * ux), vy = cov_norm * (uyy - uy * uy), vxy = cov_norm * (uxy - ux * uy);
      const double A1 = 2.0 * ux * uy + C1, A2 = 2.0 * vxy + C2, B1 = ux * ux + uy * uy + C1, B2 = vx + vy + C2;
      acc[ch] += (A1 * A2) / (B1 * B2);
    }
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    double v = acc[ch];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) red[ch][w] = v;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int ch = threadIdx.x;
    partial[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3 + ch] = ((red[ch][0] + red[ch][1]) + red[ch][2]) + red[ch][3];
  }
}

// ---- error tables by worst-channel error: tab[e] = {pixels, sum of squared differences (3 channels), sum of |differences|}
// over the pixels whose largest channel error is e (0..255).  Everything calculate_adaptive_quality_metrics
// (comparison.py:345-536) derives -- percentiles, outlier thresholds, metrics of the pixels below a threshold, the error
// histogram -- is a function of these 256 rows.
__global__ __launch_bounds__(256) void error_tables_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, long long n_px,
                                                           unsigned long long* __restrict__ tab /* [256][3] */, uint8_t* __restrict__ maxerr) {
  __shared__ unsigned long long s_tab[256][3];
  for (int i = threadIdx.x; i < 256 * 3; i += 256) (&s_tab[0][0])[i] = 0ull;
  __syncthreads();
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < n_px; p += (long long)gridDim.x * 256) {
    unsigned e = 0, sq = 0, ab = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const unsigned d = (unsigned)abs((int)a[p * 3 + c] - (int)b[p * 3 + c]);
      e = max(e, d);
      sq += d * d;
      ab += d;
    }
    atomicAdd(&s_tab[e][0], 1ull);
    atomicAdd(&s_tab[e][1], (unsigned long long)sq);
    atomicAdd(&s_tab[e][2], (unsigned long long)ab);
    if (maxerr) maxerr[p] = (uint8_t)e;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 256 * 3; i += 256) {
    const unsigned long long v = (&s_tab[0][0])[i];
    if (v) atomicAdd(&tab[i], v);
  }
}

}  // namespace rhccq

using namespace rhccq;

extern "C" {

int rhccq_error_sums(rhccq_ctx* ctx, const uint8_t* a, const uint8_t* b, int64_t n_pixels, uint64_t* sums5) {
  if (!ctx || !a || !b || !sums5 || n_pixels < 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "error_sums: bad argument");
  if (((uintptr_t)a & 3) || ((uintptr_t)b & 3)) return rhccq_fail(ctx, RHCCQ_E_ARG, "error_sums: images must be 4-byte aligned");
  RHCCQ_HIP(ctx, hipMemsetAsync(sums5, 0, 5 * sizeof(uint64_t), ctx->stream));
  if (n_pixels == 0) return 0;
  long long blocks = (n_pixels / 4 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  hipLaunchKernelGGL(error_sums_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a, b, (long long)n_pixels,
                     (unsigned long long*)sums5);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_error_tables(rhccq_ctx* ctx, const uint8_t* a, const uint8_t* b, int64_t n_pixels, uint64_t* tab768, uint8_t* maxerr) {
  if (!ctx || !a || !b || !tab768 || n_pixels < 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "error_tables: bad argument");
  RHCCQ_HIP(ctx, hipMemsetAsync(tab768, 0, 768 * sizeof(uint64_t), ctx->stream));
  if (n_pixels == 0) return 0;
  long long blocks = (n_pixels + 255) / 256;
  blocks = blocks > 2048 ? 2048 : blocks;
  hipLaunchKernelGGL(error_tables_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a, b, (long long)n_pixels,
                     (unsigned long long*)tab768, maxerr);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int64_t rhccq_ssim7_blocks(int32_t H, int32_t W) {
  if (H < 7 || W < 7) return 0;
  const int64_t by = (H - 6 + kSsimTile - 1) / kSsimTile, bx = (W - 6 + kSsimTile - 1) / kSsimTile;
  return by * bx;
}

int rhccq_ssim7_sums(rhccq_ctx* ctx, const uint8_t* a, const uint8_t* b, int32_t H, int32_t W, double* partial, int64_t n_blocks) {
  if (!ctx || !a || !b || !partial) return rhccq_fail(ctx, RHCCQ_E_ARG, "ssim7: bad argument");
  if (H < 7 || W < 7) return rhccq_fail(ctx, RHCCQ_E_ARG, "ssim7: win_size exceeds image extent");   // skimage raises ValueError
  if (n_blocks != rhccq_ssim7_blocks(H, W)) return rhccq_fail(ctx, RHCCQ_E_ARG, "ssim7: partial must hold rhccq_ssim7_blocks(H, W) x 3 doubles");
  const dim3 grid((unsigned)((W - 6 + kSsimTile - 1) / kSsimTile), (unsigned)((H - 6 + kSsimTile - 1) / kSsimTile));
  hipLaunchKernelGGL(ssim7_kernel, grid, dim3(256), 0, ctx->stream, a, b, (int)H, (int)W, partial);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

}  // extern "C"
