// K5: merge_region_components_simple (encoder/compression/merging.py:8-120), one component at a
// time: a first-occurrence pass (atomicMin of the raster position per palette entry, used by the host
// to assign first-seen global indices) and a paint pass (later launches overwrite, so painting the
// components in reversed list order reproduces "earlier component wins").  HBM-streaming kernels:
// 4 B index read (+ 4 B canvas write) per pixel.
#include "rhccq_common.h"

namespace rhccq {

__global__ __launch_bounds__(256) void merge_firstpos_kernel(const int32_t* __restrict__ idx, int h, int w, int top, int left,
                                                             int ch, int cw, int pal_n, int32_t* __restrict__ first_pos) {
  // (h * w <= INT32_MAX, checked by the entry point: 32-bit positions -- the 64-bit division and remainder per pixel this loop used to
  //  do were most of its 1.9 ms on the 8.3 M pixels of a 4K class)
  const unsigned n = (unsigned)h * (unsigned)w, uw = (unsigned)w;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const unsigned q = i / uw;
    const int r = (int)q + top, c = (int)(i - q * uw) + left;
    if (r < 0 || r >= ch || c < 0 || c >= cw) continue;
    const int32_t v = idx[i];
    if (v < 0 || v >= pal_n) continue;                  // merging.py:72
    int32_t* fp = first_pos + v;
    if ((int32_t)i < *fp) atomicMin(fp, (int32_t)i);
  }
}

__global__ __launch_bounds__(256) void merge_paint_kernel(const int32_t* __restrict__ idx, int h, int w, int top, int left, int ch,
                                                          int cw, const int32_t* __restrict__ lut, int pal_n, int32_t* __restrict__ canvas) {
  const int64_t n = (int64_t)h * w;
  const unsigned uw = (unsigned)w;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int r, c;
    if (n <= 0x7fffffff) {                                // (every image the pipeline sees: one 32-bit division instead of two 64-bit ones)
      const unsigned q = (unsigned)i / uw;
      r = (int)q + top;
      c = (int)((unsigned)i - q * uw) + left;
    } else {
      r = (int)(i / w) + top;
      c = (int)(i % w) + left;
    }
    if (r < 0 || r >= ch || c < 0 || c >= cw) continue;
    const int32_t v = idx[i];
    if (v < 0 || v >= pal_n) continue;
    const int32_t g = lut[v];
    if (g >= 0) canvas[(int64_t)r * cw + c] = g;       // black (lut < 0) is transparent
  }
}

}  // namespace rhccq

using namespace rhccq;

extern "C" {

int rhccq_merge_firstpos(rhccq_ctx* ctx, const int32_t* idx, int32_t h, int32_t w, int32_t top, int32_t left, int32_t canvas_h,
                         int32_t canvas_w, int32_t pal_n, int32_t* first_pos) {
  if (!ctx || !idx || !first_pos || h <= 0 || w <= 0 || canvas_h <= 0 || canvas_w <= 0 || (int64_t)h * w > INT32_MAX)
    return rhccq_fail(ctx, RHCCQ_E_ARG, "merge_firstpos: bad argument");
  int64_t b = ((int64_t)h * w + 255) / 256;
  if (b > 2048) b = 2048;
  hipLaunchKernelGGL(merge_firstpos_kernel, dim3((int)b), dim3(256), 0, ctx->stream, idx, h, w, top, left, canvas_h, canvas_w, pal_n, first_pos);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_merge_paint(rhccq_ctx* ctx, const int32_t* idx, int32_t h, int32_t w, int32_t top, int32_t left, int32_t canvas_h,
                      int32_t canvas_w, const int32_t* lut, int32_t pal_n, int32_t* canvas) {
  if (!ctx || !idx || !lut || !canvas || h <= 0 || w <= 0 || canvas_h <= 0 || canvas_w <= 0)
    return rhccq_fail(ctx, RHCCQ_E_ARG, "merge_paint: bad argument");
  int64_t b = ((int64_t)h * w + 255) / 256;
  if (b > 2048) b = 2048;
  hipLaunchKernelGGL(merge_paint_kernel, dim3((int)b), dim3(256), 0, ctx->stream, idx, h, w, top, left, canvas_h, canvas_w, lut, pal_n, canvas);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

}  // extern "C"
