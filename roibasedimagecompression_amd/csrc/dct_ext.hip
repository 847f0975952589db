// EXTENSION -- not part of the reference (the reference has no block transform).
// BASELINE.json's north_star names a "batched per-block 2D DCT + per-region quantisation tile that
// stages 8x8/16x16 blocks in LDS"; this file provides it, validated against scipy.fft.dctn
// (type 2, norm='ortho') in tests/, and reported separately from the RHCCQ parity path.
//
// One 256-thread workgroup stages a 32x32 pixel super-tile (16 8x8 blocks or 4 16x16 blocks) in LDS
// (padded rows: conflict-free column reads), runs the separable DCT-II as two passes of float64
// dot products against an LDS-resident basis (K = 8/16 inner dimension per output, accumulate in
// f64 so that the quantised integers are robust), then writes coefficients (f32) and
// q = rint(coef / qstep[tile]) (int16).  HBM-bound: 4 B read + 6 B written per pixel.
#include "rhccq_common.h"

namespace rhccq {

constexpr int kTile = 32;

template <int B>
__global__ __launch_bounds__(256) void dct_quant_kernel(const float* __restrict__ plane, int H, int W, const float* __restrict__ qstep,
                                                        float* __restrict__ coef_out, int16_t* __restrict__ q_out) {
  __shared__ double basis[B][B + 1];        // basis[u][x] = s(u) cos((2x+1) u pi / 2B)
  __shared__ double tile[kTile][kTile + 1];
  __shared__ double tmp[kTile][kTile + 1];
  const int tid = threadIdx.x;
  for (int i = tid; i < B * B; i += 256) {
    const int u = i / B, x = i % B;
    const double s = u == 0 ? sqrt(1.0 / B) : sqrt(2.0 / B);
    basis[u][x] = s * cospi((double)((2 * x + 1) * u) / (double)(2 * B));
  }
  const int tiles_x = (W + kTile - 1) / kTile;     // edge tiles hold whole blocks (H, W multiples of B)
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x % tiles_x;
  const int r0 = ty * kTile, c0 = tx * kTile;
  // load: thread -> row tid/8, 4 consecutive columns (float4, coalesced 128 B per row)
  {
    const int r = tid >> 3, c = (tid & 7) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < H && c0 + c < W) v = *reinterpret_cast<const float4*>(plane + (size_t)(r0 + r) * W + c0 + c);
    tile[r][c] = v.x; tile[r][c + 1] = v.y; tile[r][c + 2] = v.z; tile[r][c + 3] = v.w;
  }
  __syncthreads();
  // row pass: tmp[r][bx*B + u] = sum_x tile[r][bx*B + x] * basis[u][x]
  for (int o = tid; o < kTile * kTile; o += 256) {
    const int r = o / kTile, cu = o % kTile, bx = cu / B, u = cu % B;
    double acc = 0.0;
#pragma unroll
    for (int x = 0; x < B; ++x) acc = fma(tile[r][bx * B + x], basis[u][x], acc);
    tmp[r][cu] = acc;
  }
  __syncthreads();
  // column pass + quantisation
  for (int o = tid; o < kTile * kTile; o += 256) {
    const int rv = o / kTile, c = o % kTile, by = rv / B, v = rv % B;
    double acc = 0.0;
#pragma unroll
    for (int y = 0; y < B; ++y) acc = fma(tmp[by * B + y][c], basis[v][y], acc);
    const int gr = r0 + rv, gc = c0 + c;
    if (gr >= H || gc >= W) continue;
    const double qs = (double)qstep[(size_t)(gr / B) * (W / B) + gc / B];
    if (coef_out) coef_out[(size_t)gr * W + gc] = (float)acc;
    q_out[(size_t)gr * W + gc] = (int16_t)rint(acc / qs);
  }
}

__global__ __launch_bounds__(256) void luma_kernel(const uint8_t* __restrict__ rgb, int64_t n_px, float* __restrict__ luma) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_px; i += (int64_t)gridDim.x * blockDim.x) {
    const float r = rgb[i * 3], g = rgb[i * 3 + 1], b = rgb[i * 3 + 2];
    luma[i] = (0.299f * r + 0.587f * g) + 0.114f * b;   // contraction is off for this library
  }
}

__global__ __launch_bounds__(256) void qstep_kernel(const uint8_t* __restrict__ roi, int H, int W, int B, float q_roi, float q_bg,
                                                    float* __restrict__ qstep) {
  const int tiles_x = W / B, n_tiles = (H / B) * tiles_x;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n_tiles; t += gridDim.x * blockDim.x) {
    bool any = false;
    if (roi) {
      const int r0 = (t / tiles_x) * B, c0 = (t % tiles_x) * B;
      for (int r = 0; r < B && !any; ++r)
        for (int c = 0; c < B; ++c)
          if (roi[(size_t)(r0 + r) * W + c0 + c]) { any = true; break; }
    }
    qstep[t] = any ? q_roi : q_bg;
  }
}

}  // namespace rhccq

using namespace rhccq;

extern "C" {

int rhccq_dct_quant(rhccq_ctx* ctx, const float* plane, int32_t H, int32_t W, int32_t block, const float* qstep, float* coef_out,
                    int16_t* q_out) {
  if (!ctx || !plane || !qstep || !q_out || H <= 0 || W <= 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "dct_quant: bad argument");
  if ((block != 8 && block != 16) || H % block || W % block) return rhccq_fail(ctx, RHCCQ_E_ARG, "dct_quant: block must be 8 or 16 and divide H and W");
  if (((uintptr_t)plane & 15u) != 0) return rhccq_fail(ctx, RHCCQ_E_ARG, "dct_quant: plane must be 16-byte aligned");
  const int grid = ((H + kTile - 1) / kTile) * ((W + kTile - 1) / kTile);
  if (block == 8) hipLaunchKernelGGL(dct_quant_kernel<8>, dim3(grid), dim3(256), 0, ctx->stream, plane, H, W, qstep, coef_out, q_out);
  else if (block == 16) hipLaunchKernelGGL(dct_quant_kernel<16>, dim3(grid), dim3(256), 0, ctx->stream, plane, H, W, qstep, coef_out, q_out);
  else return rhccq_fail(ctx, RHCCQ_E_ARG, "dct_quant: block must be 8 or 16");
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_luma_qstep(rhccq_ctx* ctx, const uint8_t* rgb, const uint8_t* roi_mask, int32_t H, int32_t W, int32_t block, float q_roi,
                     float q_bg, float* luma_out, float* qstep_out) {
  if (!ctx || !rgb || !luma_out || !qstep_out || H <= 0 || W <= 0 || (block != 8 && block != 16) || H % block || W % block)
    return rhccq_fail(ctx, RHCCQ_E_ARG, "luma_qstep: bad argument");
  const int64_t n = (int64_t)H * W;
  int64_t b = (n + 255) / 256;
  if (b > 2048) b = 2048;
  hipLaunchKernelGGL(luma_kernel, dim3((int)b), dim3(256), 0, ctx->stream, rgb, n, luma_out);
  const int n_tiles = (H / block) * (W / block);
  hipLaunchKernelGGL(qstep_kernel, dim3((n_tiles + 255) / 256), dim3(256), 0, ctx->stream, roi_mask, H, W, block, q_roi, q_bg, qstep_out);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

}  // extern "C"
