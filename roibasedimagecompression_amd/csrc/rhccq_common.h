// Shared device helpers for the RHCCQ gfx950 kernels (wave64, 256 CUs, 160 KiB LDS/CU).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/rhccq.h"

struct rhccq_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  // small persistent device scratch (descriptor tables handed over from the host)
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  // tuning knobs (rhccq_ctx_set_int); -1 = built-in value
  int opt_init_lds_blocks = -1;
  int opt_init_max_items = -1;
  int opt_init_kernel = 0;   // 0 = newest k-means++ chain whose tables fit LDS (third, second, first generation), 1 = first always, 2 = second (else first)
  int opt_init_cands_per_wave = 1;   // third-generation chain: candidates one search wave finds and descends for (1, 2 or 3)
  int opt_reassign_lds = 1;          // mini-batch reassignment sweeps read an LDS copy of the weights when it fits (0: always global memory)
  int opt_reassign_order = 1;        // capped reassignment takes numpy's scalar-quicksort slots (1, default) or the stable order (0)
  // the lanes of rhccq_encode_frame (csrc/encode_frame.hip): sibling contexts with streams and arenas of their own, kept between frames
  void* frame_state = nullptr;
  void (*frame_state_free)(void*) = nullptr;
  int opt_init_shards = 1;   // workgroups per problem of the second-generation chain: 1 = one (default), 2 / 4 / 8 = at most that many
};

#define RHCCQ_HIP(ctx, expr)                                                        \
  do {                                                                              \
    hipError_t _e = (expr);                                                         \
    if (_e != hipSuccess) {                                                         \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(_e);               \
      return RHCCQ_E_HIP;                                                           \
    }                                                                               \
  } while (0)

#define RHCCQ_LAUNCH_CHECK(ctx)                                                     \
  do {                                                                              \
    hipError_t _e = hipGetLastError();                                              \
    if (_e != hipSuccess) {                                                         \
      (ctx)->err = std::string("kernel launch: ") + hipGetErrorString(_e);          \
      return RHCCQ_E_HIP;                                                           \
    }                                                                               \
  } while (0)

static inline int rhccq_fail(rhccq_ctx* ctx, int code, const char* msg) {
  if (ctx) ctx->err = msg;
  return code;
}

// copy a small host table into the context's device scratch (async on the stream; the host
// buffer is copied by value into a pinned-free staging std::string kept alive by the ctx)
int rhccq_upload(rhccq_ctx* ctx, const void* host, size_t bytes, void** dev_out);

namespace rhccq {

constexpr int kWave = 64;

__device__ __forceinline__ uint32_t key_r(uint32_t k) { return (k >> 16) & 255u; }
__device__ __forceinline__ uint32_t key_g(uint32_t k) { return (k >> 8) & 255u; }
__device__ __forceinline__ uint32_t key_b(uint32_t k) { return k & 255u; }

// exact integer squared distance between two packed colours (top byte 0): |a|^2 + |b|^2 - 2 a.b with the
// packed 4 x u8 dot product of gfx950 (v_dot4_u32_u8): 3 instructions + 2 instead of unpack / sub / mul chains
__device__ __forceinline__ unsigned norm2_key(uint32_t a) { return __builtin_amdgcn_udot4(a, a, 0u, false); }
__device__ __forceinline__ int dist2_keys(uint32_t a, uint32_t b) {
  return (int)(norm2_key(a) + norm2_key(b) - 2u * __builtin_amdgcn_udot4(a, b, 0u, false));
}
// same with the centre's norm precomputed (wave-uniform candidates)
__device__ __forceinline__ unsigned dist2_keys_n(uint32_t a, unsigned na, uint32_t b) {
  return na + norm2_key(b) - 2u * __builtin_amdgcn_udot4(a, b, 0u, false);
}

// ---- KM64 float64 arithmetic shared by K7 / K8 (this library is built with -ffp-contract=off: the only
// fused operations are the explicit fma() calls below) --------------------------------------------------
// sklearn's Lloyd E-step evaluates ||c||^2 + (-2) <x, c> with <x, c> from an OpenBLAS dgemm (FMA chain over
// k = 0,1,2 starting from the rounded product x0*c0) and ||c||^2 from numpy's einsum (two-lane SSE2
// accumulation: (c0^2 + c2^2) + c1^2).  Exact real-arithmetic ties are common on integer colour lattices,
// so these roundings decide labels; oracle/km64_estep.c restates the same expression with libm's fma.
__device__ __forceinline__ double km64_csq(double c0, double c1, double c2) { return (c0 * c0 + c2 * c2) + c1 * c1; }
__device__ __forceinline__ double km64_dot(double x0, double x1, double x2, double c0, double c1, double c2) {
  return fma(x2, c2, fma(x1, c1, x0 * c0));
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;  // valid in lane 0
}

// block-wide sum; `red` must hold blockDim.x/64 elements; result returned to every thread
// row and column of linear pixel index p in rows of W: ONE 32-bit division for every image of fewer than 2^32 pixels (all of them),
// instead of a 64-bit division and a 64-bit remainder -- ~200 instructions that were most of what the light per-pixel kernels did
__device__ __forceinline__ void rhccq_row_col(long long p, int W, int& y, int& x) {
  if ((unsigned long long)p >> 32) {
    y = (int)(p / W);
    x = (int)(p - (long long)y * W);
  } else {
    const unsigned q = (unsigned)p / (unsigned)W;
    y = (int)q;
    x = (int)((unsigned)p - q * (unsigned)W);
  }
}

template <typename T>
__device__ __forceinline__ T block_sum(T v, T* red) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  T t = 0;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

// block-wide exclusive scan of one value per thread (blockDim.x multiple of 64, <= 1024);
// `red` holds blockDim.x/64 + 1 elements; total returned through *total
template <typename T>
__device__ __forceinline__ T block_exscan(T v, T* red, T* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  T inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    T t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  __syncthreads();
  if (lane == 63) red[w] = inc;
  __syncthreads();
  T base = 0, tot = 0;
  for (int i = 0; i < nw; ++i) {
    if (i < w) base += red[i];
    tot += red[i];
  }
  if (total) *total = tot;
  return base + inc - v;
}

}  // namespace rhccq
