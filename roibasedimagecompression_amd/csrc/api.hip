// Context management and host-only entry points of librhccq_hip.so.
#include <cmath>
#include <cstring>
#include <vector>

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
  if (ctx->frame_state && ctx->frame_state_free) ctx->frame_state_free(ctx->frame_state);
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
      if (value < 0 || value > 5) return rhccq_fail(ctx, RHCCQ_E_ARG, "RHCCQ_OPT_INIT_KERNEL: 0 .. 5");
      ctx->opt_init_kernel = (int)value;
      return 0;
    case RHCCQ_OPT_INIT_CANDS_PER_WAVE:
      if (value < 1 || value > 3) return rhccq_fail(ctx, RHCCQ_E_ARG, "RHCCQ_OPT_INIT_CANDS_PER_WAVE: 1, 2 or 3");
      ctx->opt_init_cands_per_wave = (int)value;
      return 0;
    case RHCCQ_OPT_REASSIGN_LDS:
      if (value != 0 && value != 1) return rhccq_fail(ctx, RHCCQ_E_ARG, "RHCCQ_OPT_REASSIGN_LDS: 0 or 1");
      ctx->opt_reassign_lds = (int)value;
      return 0;
    case RHCCQ_OPT_REASSIGN_ORDER:
      if (value != 0 && value != 1) return rhccq_fail(ctx, RHCCQ_E_ARG, "RHCCQ_OPT_REASSIGN_ORDER: 0 or 1");
      ctx->opt_reassign_order = (int)value;
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

// merge_region_components_simple in palette space (encoder/compression/merging.py:8-120) on the host: the components' palettes
// are painted in REVERSED order, every component's entries in the order of their first raster positions; a colour gets the
// index of its first appearance in that sequence (0 = canvas black), the merged first position of an index is the smallest
// of its members'.  keys[c] / fp[c]: counts[c] entries; entries with key 0 or fp >= fp_none do not appear.
// Out: gkeys[0] = 0 and the merged palette behind it, gfp (fp_none for index 0), lut[c][i] = merged index of entry i (0 when it
// does not appear); *n_out = merged entries including index 0.  Streaming radix passes instead of five numpy sorts (2-3 ms per
// class of a 4K frame, on the critical path between level 1 and level 2).  Pure host code.
int rhccq_merge_palettes_host(int32_t n_comp, const uint32_t* const* keys, const int64_t* const* fp, const int32_t* counts, int64_t fp_none,
                              uint32_t* gkeys, int64_t* gfp, int32_t* const* lut, int64_t* n_out) {
  if (n_comp < 0 || !counts || !gkeys || !gfp || !n_out || (n_comp > 0 && (!keys || !fp || !lut))) return RHCCQ_E_ARG;
  size_t total = 0;
  for (int c = 0; c < n_comp; ++c) {
    if (counts[c] < 0 || (counts[c] > 0 && (!keys[c] || !fp[c] || !lut[c]))) return RHCCQ_E_ARG;
    total += (size_t)counts[c];
  }
  // 1. the painting sequence: components reversed, entries by first position (LSD radix on the position bits in use)
  std::vector<uint64_t> ord, tmp;
  std::vector<uint32_t> skey;                              // colour of sequence entry s
  std::vector<int64_t> sfp;
  std::vector<int32_t*> slot;                              // where its merged index goes
  skey.reserve(total); sfp.reserve(total); slot.reserve(total);
  for (int c = n_comp - 1; c >= 0; --c) {
    const uint32_t* k = keys[c];
    const int64_t* f = fp[c];
    const int32_t m = counts[c];
    if (m >= (1 << 24)) return RHCCQ_E_LIMIT;
    ord.clear();
    for (int32_t i = 0; i < m; ++i) {
      lut[c][i] = 0;
      if (k[i] != 0u && f[i] < fp_none) {
        if (f[i] < 0 || f[i] >= (1ll << 40)) return RHCCQ_E_ARG;
        ord.push_back(((uint64_t)f[i] << 24) | (uint64_t)(uint32_t)i);     // first positions of distinct entries are distinct
      }
    }
    uint64_t top = 0;
    for (uint64_t v : ord) top |= v >> 24;
    tmp.resize(ord.size());
    for (int shift = 24; (top >> (shift - 24)) != 0; shift += 11) {
      size_t cnt[2049] = {0};
      for (uint64_t v : ord) ++cnt[((v >> shift) & 2047) + 1];
      for (int b = 0; b < 2048; ++b) cnt[b + 1] += cnt[b];
      for (uint64_t v : ord) tmp[cnt[(v >> shift) & 2047]++] = v;
      ord.swap(tmp);
    }
    for (uint64_t v : ord) {
      const int32_t i = (int32_t)(v & 0xffffffu);
      skey.push_back(k[i]); sfp.push_back(f[i]); slot.push_back(&lut[c][i]);
    }
  }
  // 2. equal colours next to each other, sequence order kept inside a run (stable LSD radix over the 24 colour bits): streaming
  // passes only -- a hash table of this size (1 MB) cost a cache miss per entry
  const size_t M = skey.size();
  if (M >= (1ull << 32)) return RHCCQ_E_LIMIT;
  std::vector<uint64_t> a(M), b2(M);
  for (size_t q = 0; q < M; ++q) a[q] = ((uint64_t)skey[q] << 32) | (uint64_t)q;
  for (int shift = 32; shift < 56; shift += 8) {
    size_t cnt[257] = {0};
    for (uint64_t v : a) ++cnt[((v >> shift) & 255) + 1];
    for (int q = 0; q < 256; ++q) cnt[q + 1] += cnt[q];
    for (uint64_t v : a) b2[cnt[(v >> shift) & 255]++] = v;
    a.swap(b2);
  }
  // 3. a run's first member is the colour's first appearance; appearances in sequence order number the merged palette
  std::vector<uint32_t> leader(M);                         // sequence entry -> sequence entry of its colour's first appearance
  for (size_t q = 0; q < M;) {
    const uint32_t key = (uint32_t)(a[q] >> 32), lead = (uint32_t)a[q];
    size_t e = q;
    while (e < M && (uint32_t)(a[e] >> 32) == key) leader[(uint32_t)a[e++]] = lead;
    q = e;
  }
  std::vector<int32_t> rank(M);
  gkeys[0] = 0u;
  gfp[0] = fp_none;
  int64_t n = 1;
  for (size_t q = 0; q < M; ++q) {
    if (leader[q] == q) {
      rank[q] = (int32_t)n;
      gkeys[n] = skey[q];
      gfp[n] = sfp[q];
      ++n;
    }
  }
  for (size_t q = 0; q < M; ++q) {
    const int32_t g = rank[leader[q]];
    *slot[q] = g;
    if (sfp[q] < gfp[g]) gfp[g] = sfp[q];
  }
  *n_out = n;
  return 0;
}

// out[idx[i]] = min(out[idx[i]], val[i]) for i < count (first positions of the entries a clustering maps together: the smallest
// survives, merging.py:77-79 / clustering.py:373-377 carried through the levels).  numpy has no fast scatter-min (ufunc.at is slow,
// sort + reduceat costs a sort): one loop on the host.  Returns 0, or RHCCQ_E_ARG for an index outside [0, n_out).
int rhccq_scatter_min_host(int64_t n_out, const int32_t* idx, const int64_t* val, int64_t count, int64_t* out) {
  if (n_out < 0 || count < 0 || (count > 0 && (!idx || !val || !out))) return RHCCQ_E_ARG;
  for (int64_t i = 0; i < count; ++i) {
    const int64_t t = idx[i];
    if (t < 0 || t >= n_out) return RHCCQ_E_ARG;
    if (val[i] < out[t]) out[t] = val[i];
  }
  return 0;
}

// The bookkeeping behind one clustered palette (clustering.py:296-377 for clusters that need no split): from the member sums of
// the k clusters (uint64 r, g, b, count each) the floor-mean colour of every non-empty cluster in label order behind `nblack`
// black rows, and the label -> new palette index table (uint16-valued like the reference's mapping_array).  Returns the
// number of non-empty clusters, or -1 when a cluster holds more than `mc` colours (it needs the k-means split: host path).
// A dozen numpy passes per palette between level 1 and level 2 of a frame (~0.5 ms each) as one loop.  Pure host code.
int64_t rhccq_cluster_plan_host(const unsigned long long* sums, int64_t k, int64_t mc, int32_t nblack, uint32_t* new_keys, int32_t* lut) {
  if (!sums || k < 0 || nblack < 0 || !new_keys || !lut) return -2;
  for (int64_t j = 0; j < k; ++j)
    if ((int64_t)sums[4 * j + 3] > mc) return -1;
  for (int32_t b = 0; b < nblack; ++b) new_keys[b] = 0u;
  int64_t n = 0;
  for (int64_t j = 0; j < k; ++j) {
    const unsigned long long c = sums[4 * j + 3];
    int64_t leaf = 0;
    if (c) {
      new_keys[nblack + n] = ((uint32_t)(sums[4 * j] / c) << 16) | ((uint32_t)(sums[4 * j + 1] / c) << 8) | (uint32_t)(sums[4 * j + 2] / c);
      leaf = n++;
    }
    lut[j] = (int32_t)((nblack + leaf) & 0xFFFF);
  }
  return n;
}

}  // extern "C"
