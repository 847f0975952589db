// Context management and host-only entry points of librhccq_hip.so.
#include <cmath>
#include <cstring>

#include "rhccq_common.h"

int rhccq_upload(rhccq_ctx* ctx, const void* host, size_t bytes, void** dev_out) {
  (void)ctx; (void)host; (void)bytes; (void)dev_out;
  return RHCCQ_E_ARG;  // tables are passed by value in kernel arguments or as torch tensors
}

extern "C" {

int rhccq_abi_version(void) { return 1; }

int rhccq_ctx_create(int device, void* hip_stream, rhccq_ctx** out) {
  if (!out) return RHCCQ_E_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return RHCCQ_E_HIP;
  if (hipSetDevice(device) != hipSuccess) return RHCCQ_E_HIP;
  rhccq_ctx* c = new rhccq_ctx();
  c->device = device;
  // NULL = the device's default (null) stream, which is what torch uses unless told otherwise;
  // the context never creates streams of its own so that torch allocations / memsets and these
  // kernels stay ordered on one stream
  c->stream = (hipStream_t)hip_stream;
  c->own_stream = false;
  *out = c;
  return 0;
}

void rhccq_ctx_destroy(rhccq_ctx* ctx) {
  if (!ctx) return;
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char* rhccq_last_error(const rhccq_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int rhccq_ctx_set_int(rhccq_ctx* ctx, int32_t option, int64_t value) {
  if (!ctx) return RHCCQ_E_ARG;
  switch (option) {
    case RHCCQ_OPT_INIT_LDS_BLOCKS:
      if (value < 0 || value > 4096) return rhccq_fail(ctx, RHCCQ_E_ARG, "RHCCQ_OPT_INIT_LDS_BLOCKS: 0..4096");
      ctx->opt_init_lds_blocks = (int)value;
      return 0;
    case RHCCQ_OPT_INIT_MAX_ITEMS:
      if (value < 1 || value > 12288) return rhccq_fail(ctx, RHCCQ_E_ARG, "RHCCQ_OPT_INIT_MAX_ITEMS: 1..12288");
      ctx->opt_init_max_items = (int)value;
      return 0;
    case RHCCQ_OPT_INIT_KERNEL:
      if (value < 0 || value > 4) return rhccq_fail(ctx, RHCCQ_E_ARG, "RHCCQ_OPT_INIT_KERNEL: 0 .. 4");
      ctx->opt_init_kernel = (int)value;
      return 0;
    case RHCCQ_OPT_INIT_CANDS_PER_WAVE:
      if (value < 1 || value > 3) return rhccq_fail(ctx, RHCCQ_E_ARG, "RHCCQ_OPT_INIT_CANDS_PER_WAVE: 1, 2 or 3");
      ctx->opt_init_cands_per_wave = (int)value;
      return 0;
    case RHCCQ_OPT_INIT_SHARDS:
      if (value != 1 && value != 2 && value != 4 && value != 8) return rhccq_fail(ctx, RHCCQ_E_ARG, "RHCCQ_OPT_INIT_SHARDS: 1, 2, 4 or 8");
      ctx->opt_init_shards = (int)value;
      return 0;
    default:
      return rhccq_fail(ctx, RHCCQ_E_ARG, "unknown option");
  }
}

int rhccq_sync(rhccq_ctx* ctx) {
  if (!ctx) return RHCCQ_E_ARG;
  RHCCQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

void* rhccq_stream(rhccq_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int rhccq_ctx_set_stream(rhccq_ctx* ctx, void* hip_stream) {
  if (!ctx) return RHCCQ_E_ARG;
  ctx->stream = (hipStream_t)hip_stream;                // not owned; later entry points launch on it
  return 0;
}

// compute_clustering_params (encoder/compression/clustering.py:127-133), same float64 expressions
int rhccq_params(int64_t n_colors, double quality, double* eps_host, int64_t* max_colors_host) {
  if (!eps_host || !max_colors_host) return RHCCQ_E_ARG;
  if (quality == 0.0) return RHCCQ_E_ARG;               // the reference divides by zero here
  double eps = 128.0 - 1.28 * quality;
  double v = (-(quality / 100.0) * (double)n_colors + (double)n_colors) / quality;
  int64_t mc = (int64_t)std::ceil(v);
  if (eps == 0.0) eps = 1.0;
  if (mc == 0) mc = 1;
  *eps_host = eps;
  *max_colors_host = mc;
  return 0;
}

// exact-rational eps^2: a double is m * 2^e, so eps^2 is integral iff (m^2) * 2^(2e) is
int rhccq_eps_threshold(double eps, int32_t* thr_host, int32_t* boundary_host, double* r2_host) {
  if (!thr_host || !boundary_host || !r2_host || !(eps >= 0.0) || eps > 1024.0) return RHCCQ_E_ARG;
  const double r = eps / 255.0;
  *r2_host = r * r;
  // eps < 2^10 with a 53-bit mantissa: eps = M / 2^43 for an integer M < 2^53; eps^2 = M^2 / 2^86.
  const double scaled = std::ldexp(eps, 43);
  const unsigned __int128 M = (unsigned __int128)(unsigned long long)scaled;  // exact: eps*2^43 is an integer < 2^53
  const unsigned __int128 M2 = M * M;                                            // < 2^106
  const unsigned __int128 ip = M2 >> 86;
  const bool integral = (M2 & ((((unsigned __int128)1) << 86) - 1)) == 0;
  if (integral) {
    *thr_host = (int32_t)ip - 1;
    *boundary_host = (int32_t)ip;
  } else {
    *thr_host = (int32_t)ip;
    *boundary_host = -1;
  }
  return 0;
}

}  // extern "C"
