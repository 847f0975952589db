// EXTENSION -- not part of the reference (the reference clusters PALETTE colours; it has no
// pixel-space clustering).  BASELINE.json's north_star names "a fixed-radius neighbour search over (x,y,Lab) pixel
// features using an LDS-staged spatial grid with coalesced HBM reads of the pixel buffer" and "a wavefront-frontier
// region-growing pass for cluster expansion"; this file provides that variant as a self-contained operator,
// validated against a brute-force restatement in oracle/ (tests/), and reported separately from the RHCCQ path.
//
// Definition (the builder's own, integer labels bit-exact by construction):
//   * features of pixel p: (x, y) and a float32 CIE-Lab computed with a fixed sequence of individually rounded
//     float32 operations: 8-bit sRGB -> linear through a 256-entry table, XYZ by three dot products evaluated left
//     to right and scaled by the reciprocal white point, and the Lab transfer function f(t) (cube root / linear toe)
//     by linear interpolation in a 1025-point table of f(i / 1024) -- both tables come from the caller (computed in
//     float64, rounded once), so the GPU and the restatement in oracle/ share them bit for bit;
//   * q is a neighbour of p when dx^2 + dy^2 <= radius^2 and
//       ((dL^2 + da^2) + db^2) + w(dx, dy) <= eps^2,   w = spatial_weight^2 * (dx^2 + dy^2)   (float32);
//   * p is a core pixel when it has at least min_pts neighbours (itself included);
//   * clusters = connected components of the core pixels under the neighbour relation, named by their smallest
//     pixel index + 1; a non-core pixel takes the smallest cluster name among its core neighbours, 0 (noise) if none.
//
// px_neighbours_kernel (the "neighbour pass"): one streaming pass, 3 B read + 4 B written per pixel.  A 64 x 16
// pixel tile and its apron are converted to Lab once and staged in LDS as three padded float planes; each thread
// then tests the window of its 4 consecutive pixels against LDS only.
// px_union_kernel / px_label_kernel (the expansion): lock-free union-find over the core pixels (roots = smallest
// index, compare-and-swap linking of the larger root under the smaller), then one labelling pass.
#include "rhccq_common.h"

namespace rhccq {

constexpr int kPxTW = 64, kPxTH = 16, kPxMaxR = 4, kPxMaxOff = 64;
constexpr int kPxLW = kPxTW + 2 * kPxMaxR, kPxLH = kPxTH + 2 * kPxMaxR;   // 72 x 24 staged pixels at most

struct PxWindow {          // window offsets in a fixed order (dy major, dx minor), self first
  int n, n_forward;        // n_forward: offsets [1, 1 + n_forward) are the "forward" half (dy > 0 or dy == 0 and dx > 0)
  signed char dx[kPxMaxOff], dy[kPxMaxOff];
  float w[kPxMaxOff];
};

constexpr int kPxLin = 256, kPxFSteps = 1024, kPxTables = kPxLin + 2 * kPxFSteps;   // caller's table: lin[256], then (f, df)[1024]

// f(t) by linear interpolation: entry i holds f(i / 1024) and the float32 difference to f((i + 1) / 1024)
__device__ __forceinline__ float px_f(const float* __restrict__ ftab, float t) {
  const float u = t * 1024.0f;
  const int i = min((int)u, kPxFSteps - 1);
  const float frac = u - (float)i;
  const float2 fd = *reinterpret_cast<const float2*>(ftab + 2 * i);
  return fd.x + frac * fd.y;
}

__device__ __forceinline__ void px_lab(const float* __restrict__ tab, unsigned r8, unsigned g8, unsigned b8, float& L, float& A, float& B) {
  const float r = tab[r8], g = tab[g8], b = tab[b8];
  const float X = ((0.4124564f * r + 0.3575761f * g) + 0.1804375f * b) * 1.0521111f;     // 1 / 0.95047
  const float Y = (0.2126729f * r + 0.7151522f * g) + 0.0721750f * b;
  const float Z = ((0.0193339f * r + 0.1191920f * g) + 0.9503041f * b) * 0.9184170f;     // 1 / 1.08883
  const float* ftab = tab + kPxLin;
  const float fx = px_f(ftab, X), fy = px_f(ftab, Y), fz = px_f(ftab, Z);
  L = 116.0f * fy - 16.0f;
  A = 500.0f * (fx - fy);
  B = 200.0f * (fy - fz);
}

struct PxTile {
  float L[kPxLH][kPxLW + 1], A[kPxLH][kPxLW + 1], B[kPxLH][kPxLW + 1];
};

// stage the tile + apron as Lab; pixels outside the image get L = +inf (never within eps of anything)
// Staging for the neighbour pass: columns [x0 - 4, x0 + 68) in 18 quads of 4 pixels per row (x0 is a multiple of
// 64), rows [y0 - R, y0 + 16 + R); LDS column index = x - x0 + 4.  A quad that lies inside the image and starts on
// a 4-byte boundary is read as three dwords, and those reads are issued one tile ahead (px_prefetch) so that their
// latency hides behind the window tests of the current tile; edge quads are read bytewise at conversion time.
constexpr int kPxApron = 4, kPxQuads = (kPxTW + 2 * kPxApron) / 4, kPxQuadRounds = 2;   // <= 18 * 24 = 432 quads = 2 rounds of 256
struct PxPrefetch { uint32_t w[kPxQuadRounds][3]; };

template <int kR>
__device__ __forceinline__ void px_quad_pos(int i, int y0, int x0, int& ly, int& lx, int& y, int& x) {
  ly = i / kPxQuads;
  const int qd = i - ly * kPxQuads;
  lx = 4 * qd;
  y = y0 + ly - kR;
  x = x0 - kPxApron + 4 * qd;
}
__device__ __forceinline__ bool px_quad_fast(int y, int x, int H, int W) {
  return y >= 0 && y < H && x >= 0 && x + 3 < W && ((((size_t)y * W + x) * 3) & 3) == 0;
}

template <int kR>
__device__ __forceinline__ void px_prefetch(PxPrefetch& pr, const uint8_t* __restrict__ rgb, int H, int W, int y0, int x0) {
  constexpr int n_quads = kPxQuads * (kPxTH + 2 * kR);
#pragma unroll
  for (int r = 0; r < kPxQuadRounds; ++r) {
    const int i = threadIdx.x + r * 256;
    int ly, lx, y, x;
    px_quad_pos<kR>(i, y0, x0, ly, lx, y, x);
    if (i < n_quads && px_quad_fast(y, x, H, W)) {
      const uint32_t* src = reinterpret_cast<const uint32_t*>(rgb + ((size_t)y * W + x) * 3);
      pr.w[r][0] = src[0]; pr.w[r][1] = src[1]; pr.w[r][2] = src[2];
    }
  }
}

template <int kR>
__device__ __forceinline__ void px_stage_r(PxTile& t, const float* s_lut, const PxPrefetch& pr, const uint8_t* __restrict__ rgb, int H, int W,
                                           int y0, int x0) {
  constexpr int n_quads = kPxQuads * (kPxTH + 2 * kR);
#pragma unroll
  for (int r = 0; r < kPxQuadRounds; ++r) {
    const int i = threadIdx.x + r * 256;
    if (i >= n_quads) break;
    int ly, lx, y, x;
    px_quad_pos<kR>(i, y0, x0, ly, lx, y, x);
    float L[4], A[4], B[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { L[k] = INFINITY; A[k] = 0.0f; B[k] = 0.0f; }
    if (px_quad_fast(y, x, H, W)) {
      const uint32_t w0 = pr.w[r][0], w1 = pr.w[r][1], w2 = pr.w[r][2];
      px_lab(s_lut, w0 & 255u, (w0 >> 8) & 255u, (w0 >> 16) & 255u, L[0], A[0], B[0]);
      px_lab(s_lut, w0 >> 24, w1 & 255u, (w1 >> 8) & 255u, L[1], A[1], B[1]);
      px_lab(s_lut, (w1 >> 16) & 255u, w1 >> 24, w2 & 255u, L[2], A[2], B[2]);
      px_lab(s_lut, (w2 >> 8) & 255u, (w2 >> 16) & 255u, w2 >> 24, L[3], A[3], B[3]);
    } else if (y >= 0 && y < H) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (x + k < 0 || x + k >= W) continue;
        const uint8_t* px = rgb + ((size_t)y * W + (x + k)) * 3;
        px_lab(s_lut, px[0], px[1], px[2], L[k], A[k], B[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { t.L[ly][lx + k] = L[k]; t.A[ly][lx + k] = A[k]; t.B[ly][lx + k] = B[k]; }
  }
}

__device__ __forceinline__ void px_stage(PxTile& t, const float* s_lut, const uint8_t* __restrict__ rgb, int H, int W, int y0, int x0, int R) {
  const int lw = kPxTW + 2 * R, lh = kPxTH + 2 * R;
  for (int i = threadIdx.x; i < lw * lh; i += 256) {
    const int ly = i / lw, lx = i - ly * lw;
    const int y = y0 + ly - R, x = x0 + lx - R;
    float L = INFINITY, A = 0.0f, B = 0.0f;
    if (y >= 0 && y < H && x >= 0 && x < W) {
      const uint8_t* px = rgb + ((size_t)y * W + x) * 3;
      px_lab(s_lut, px[0], px[1], px[2], L, A, B);
    }
    t.L[ly][lx] = L; t.A[ly][lx] = A; t.B[ly][lx] = B;
  }
}

__device__ __forceinline__ bool px_near(const PxTile& t, int ly, int lx, int qy, int qx, float w, float eps2) {
  const float dL = t.L[ly][lx] - t.L[qy][qx], dA = t.A[ly][lx] - t.A[qy][qx], dB = t.B[ly][lx] - t.B[qy][qx];
  const float d2 = ((dL * dL + dA * dA) + dB * dB) + w;
  return d2 <= eps2;
}

template <int R>
__global__ __launch_bounds__(256) void px_neighbours_kernel(const uint8_t* __restrict__ rgb, int H, int W, float eps2, float ws2, int min_pts,
                                                            const float* __restrict__ lin_lut, int32_t* __restrict__ parent,
                                                            uint8_t* __restrict__ count_out) {
  __shared__ PxTile t;
  __shared__ __attribute__((aligned(16))) float s_lut[kPxTables];
  for (int i = threadIdx.x; i < kPxTables; i += 256) s_lut[i] = lin_lut[i];
  const int tiles_x = (W + kPxTW - 1) / kPxTW, n_tiles = tiles_x * ((H + kPxTH - 1) / kPxTH);
  // persistent workgroups: the tables are read once per workgroup, and the pixels of the next tile are requested
  // before the window tests of the current one
  PxPrefetch pr;
  if ((int)blockIdx.x < n_tiles) px_prefetch<R>(pr, rgb, H, W, ((int)blockIdx.x / tiles_x) * kPxTH, ((int)blockIdx.x % tiles_x) * kPxTW);
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
  __syncthreads();                                         // tables ready / previous tile's readers done
  const int y0 = (tile / tiles_x) * kPxTH, x0 = (tile % tiles_x) * kPxTW;
  px_stage_r<R>(t, s_lut, pr, rgb, H, W, y0, x0);
  {
    const int nt = tile + (int)gridDim.x;
    if (nt < n_tiles) px_prefetch<R>(pr, rgb, H, W, (nt / tiles_x) * kPxTH, (nt % tiles_x) * kPxTW);
  }
  __syncthreads();
  const int row = threadIdx.x >> 4, c4 = (threadIdx.x & 15) * 4;
  const int y = y0 + row;
  if (y >= H) continue;
  // a thread owns 4 consecutive pixels = two float2 pairs: the distance arithmetic runs on packed float32
  // instructions (v_pk_add / v_pk_mul), element-wise identical to the scalar form
  typedef float f2 __attribute__((ext_vector_type(2)));
  const int ly = row + R, lx = c4 + kPxApron;
  const f2 pL0 = {t.L[ly][lx], t.L[ly][lx + 1]}, pL1 = {t.L[ly][lx + 2], t.L[ly][lx + 3]};
  const f2 pA0 = {t.A[ly][lx], t.A[ly][lx + 1]}, pA1 = {t.A[ly][lx + 2], t.A[ly][lx + 3]};
  const f2 pB0 = {t.B[ly][lx], t.B[ly][lx + 1]}, pB1 = {t.B[ly][lx + 2], t.B[ly][lx + 3]};
  int cnt[4] = {1, 1, 1, 1};                                // offset 0 is the pixel itself: distance 0 <= eps^2
  // the window is known at compile time: the loops unroll into straight-line code with immediate LDS offsets
#pragma unroll
  for (int dy = -R; dy <= R; ++dy)
#pragma unroll
  for (int dx = -R; dx <= R; ++dx) {
    if (dx * dx + dy * dy > R * R || (dx == 0 && dy == 0)) continue;
    const int qy = ly + dy, qx = lx + dx;
    const float w = ws2 * (float)(dx * dx + dy * dy);
    const f2 qL0 = {t.L[qy][qx], t.L[qy][qx + 1]}, qL1 = {t.L[qy][qx + 2], t.L[qy][qx + 3]};
    const f2 qA0 = {t.A[qy][qx], t.A[qy][qx + 1]}, qA1 = {t.A[qy][qx + 2], t.A[qy][qx + 3]};
    const f2 qB0 = {t.B[qy][qx], t.B[qy][qx + 1]}, qB1 = {t.B[qy][qx + 2], t.B[qy][qx + 3]};
    const f2 dL0 = pL0 - qL0, dL1 = pL1 - qL1, dA0 = pA0 - qA0, dA1 = pA1 - qA1, dB0 = pB0 - qB0, dB1 = pB1 - qB1;
    const f2 e0 = ((dL0 * dL0 + dA0 * dA0) + dB0 * dB0) + w, e1 = ((dL1 * dL1 + dA1 * dA1) + dB1 * dB1) + w;
    cnt[0] += e0.x <= eps2; cnt[1] += e0.y <= eps2; cnt[2] += e1.x <= eps2; cnt[3] += e1.y <= eps2;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int x = x0 + c4 + k;
    if (x >= W) break;
    const size_t p = (size_t)y * W + x;
    parent[p] = cnt[k] >= min_pts ? (int32_t)p : -1;
    if (count_out) count_out[p] = (uint8_t)min(cnt[k], 255);
  }
  }
}

// kCoherent: agent-scope relaxed atomic loads while other workgroups are linking roots (a plain load may be served
// by this CU's or XCD's cache; a stale parent is still an ancestor, so the walk stays correct, but fresh values
// mean fewer failed compare-and-swaps)
template <bool kCoherent>
__device__ __forceinline__ int px_find(const int32_t* parent, int x) {
  auto ld = [&](int i) { return kCoherent ? __hip_atomic_load(parent + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : parent[i]; };
  int p = ld(x);
  while (p != x) { x = p; p = ld(x); }
  return x;
}
__device__ __forceinline__ void px_union(int32_t* parent, int a, int b) {
  while (true) {
    a = px_find<true>(parent, a);
    b = px_find<true>(parent, b);
    if (a == b) return;
    if (a > b) { const int s = a; a = b; b = s; }
    const int old = atomicCAS(&parent[b], b, a);          // link the larger root under the smaller
    if (old == b) return;
    b = old;                                              // somebody re-rooted b first: retry from there
  }
}

__global__ __launch_bounds__(256) void px_union_kernel(const uint8_t* __restrict__ rgb, int H, int W, int R, float eps2, PxWindow win,
                                                       const float* __restrict__ lin_lut, int32_t* parent) {
  __shared__ PxTile t;
  __shared__ __attribute__((aligned(16))) float s_lut[kPxTables];
  for (int i = threadIdx.x; i < kPxTables; i += 256) s_lut[i] = lin_lut[i];
  __syncthreads();
  const int tiles_x = (W + kPxTW - 1) / kPxTW;
  const int y0 = (blockIdx.x / tiles_x) * kPxTH, x0 = (blockIdx.x % tiles_x) * kPxTW;
  px_stage(t, s_lut, rgb, H, W, y0, x0, R);
  __syncthreads();
  const int row = threadIdx.x >> 4, c4 = (threadIdx.x & 15) * 4;
  const int y = y0 + row;
  if (y >= H) return;
  for (int k = 0; k < 4; ++k) {
    const int x = x0 + c4 + k;
    if (x >= W) break;
    const int p = y * W + x;
    if (parent[p] < 0) continue;                          // not a core pixel (parent >= 0 <=> core, invariant)
    const int ly = row + R, lx = c4 + k + R;
    for (int o = 1; o <= win.n_forward; ++o) {            // each unordered pair once
      const int qy = y + win.dy[o], qx = x + win.dx[o];
      if (qy >= H || qx < 0 || qx >= W) continue;
      if (!px_near(t, ly, lx, ly + win.dy[o], lx + win.dx[o], win.w[o], eps2)) continue;
      const int q = qy * W + qx;
      if (parent[q] < 0) continue;
      px_union(parent, p, q);
    }
  }
}

// every core pixel points straight at its root before the labelling pass (other threads may still read the old
// pointer: either value is an ancestor)
__global__ __launch_bounds__(256) void px_flatten_kernel(int32_t* parent, long long n) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= n || parent[p] < 0) return;
  parent[p] = px_find<true>(parent, (int)p);
}

__global__ __launch_bounds__(256) void px_label_kernel(const uint8_t* __restrict__ rgb, int H, int W, int R, float eps2, PxWindow win,
                                                       const float* __restrict__ lin_lut, const int32_t* __restrict__ parent,
                                                       int32_t* __restrict__ labels) {
  __shared__ PxTile t;
  __shared__ __attribute__((aligned(16))) float s_lut[kPxTables];
  for (int i = threadIdx.x; i < kPxTables; i += 256) s_lut[i] = lin_lut[i];
  __syncthreads();
  const int tiles_x = (W + kPxTW - 1) / kPxTW;
  const int y0 = (blockIdx.x / tiles_x) * kPxTH, x0 = (blockIdx.x % tiles_x) * kPxTW;
  px_stage(t, s_lut, rgb, H, W, y0, x0, R);
  __syncthreads();
  const int row = threadIdx.x >> 4, c4 = (threadIdx.x & 15) * 4;
  const int y = y0 + row;
  if (y >= H) return;
  for (int k = 0; k < 4; ++k) {
    const int x = x0 + c4 + k;
    if (x >= W) break;
    const int p = y * W + x;
    int lab = 0;
    if (parent[p] >= 0) {
      lab = px_find<false>(parent, p) + 1;
    } else {
      const int ly = row + R, lx = c4 + k + R;
      int best = 0x7fffffff;
      for (int o = 1; o < win.n; ++o) {
        const int qy = y + win.dy[o], qx = x + win.dx[o];
        if (qy < 0 || qy >= H || qx < 0 || qx >= W) continue;
        if (!px_near(t, ly, lx, ly + win.dy[o], lx + win.dx[o], win.w[o], eps2)) continue;
        const int q = qy * W + qx;
        if (parent[q] < 0) continue;
        best = min(best, px_find<false>(parent, q));
      }
      lab = best == 0x7fffffff ? 0 : best + 1;
    }
    labels[p] = lab;
  }
}

static int make_window(rhccq_ctx* ctx, int R, float spatial_weight, PxWindow* w) {
  if (R < 0 || R > kPxMaxR) return rhccq_fail(ctx, RHCCQ_E_ARG, "px: radius must be 0..4");
  const float ws2 = spatial_weight * spatial_weight;
  int n = 0;
  w->dx[n] = 0; w->dy[n] = 0; w->w[n] = ws2 * 0.0f; ++n;                     // self first
  // forward half next (dy > 0, or dy == 0 and dx > 0), then the mirrored backward half
  for (int pass = 0; pass < 2; ++pass)
    for (int dy = -R; dy <= R; ++dy)
      for (int dx = -R; dx <= R; ++dx) {
        if (dx == 0 && dy == 0) continue;
        if (dx * dx + dy * dy > R * R) continue;
        const bool fwd = dy > 0 || (dy == 0 && dx > 0);
        if (fwd != (pass == 0)) continue;
        if (n >= kPxMaxOff) return rhccq_fail(ctx, RHCCQ_E_LIMIT, "px: window too large");
        w->dx[n] = (signed char)dx; w->dy[n] = (signed char)dy;
        w->w[n] = ws2 * (float)(dx * dx + dy * dy);
        ++n;
        if (pass == 0) w->n_forward = n - 1;
      }
  if (R == 0) w->n_forward = 0;
  w->n = n;
  return 0;
}

}  // namespace rhccq

using namespace rhccq;

extern "C" {

int rhccq_px_neighbours(rhccq_ctx* ctx, const uint8_t* rgb, int32_t H, int32_t W, int32_t radius, float eps, float spatial_weight,
                        int32_t min_pts, const float* lin_lut, int32_t* parent_out, uint8_t* count_out) {
  if (!ctx || !rgb || !lin_lut || !parent_out || H <= 0 || W <= 0 || min_pts < 1 || !(eps >= 0.0f))
    return rhccq_fail(ctx, RHCCQ_E_ARG, "px_neighbours: bad argument");
  if ((int64_t)H * W > 0x7fffffffll) return rhccq_fail(ctx, RHCCQ_E_LIMIT, "px_neighbours: more than 2^31 pixels");
  PxWindow win{};
  if (int e = make_window(ctx, radius, spatial_weight, &win)) return e;
  const int tiles = ((W + kPxTW - 1) / kPxTW) * ((H + kPxTH - 1) / kPxTH);
  const dim3 grid(tiles < 1024 ? tiles : 1024), block(256);     // persistent workgroups (146 VGPRs: 3 waves per SIMD)
  const float eps2 = eps * eps, ws2 = spatial_weight * spatial_weight;
#define RHCCQ_PX_LAUNCH(RR)                                                                                                             \
  hipLaunchKernelGGL(px_neighbours_kernel<RR>, grid, block, 0, ctx->stream, rgb, (int)H, (int)W, eps2, ws2, (int)min_pts, lin_lut, parent_out, \
                     count_out)
  switch (radius) {
    case 0: RHCCQ_PX_LAUNCH(0); break;
    case 1: RHCCQ_PX_LAUNCH(1); break;
    case 2: RHCCQ_PX_LAUNCH(2); break;
    case 3: RHCCQ_PX_LAUNCH(3); break;
    default: RHCCQ_PX_LAUNCH(4); break;
  }
#undef RHCCQ_PX_LAUNCH
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

int rhccq_px_expand(rhccq_ctx* ctx, const uint8_t* rgb, int32_t H, int32_t W, int32_t radius, float eps, float spatial_weight,
                    const float* lin_lut, int32_t* parent, int32_t* labels_out) {
  if (!ctx || !rgb || !lin_lut || !parent || !labels_out || H <= 0 || W <= 0 || !(eps >= 0.0f))
    return rhccq_fail(ctx, RHCCQ_E_ARG, "px_expand: bad argument");
  PxWindow win{};
  if (int e = make_window(ctx, radius, spatial_weight, &win)) return e;
  const int tiles = ((W + kPxTW - 1) / kPxTW) * ((H + kPxTH - 1) / kPxTH);
  hipLaunchKernelGGL(px_union_kernel, dim3(tiles), dim3(256), 0, ctx->stream, rgb, (int)H, (int)W, (int)radius, eps * eps, win, lin_lut, parent);
  const long long n = (long long)H * W;
  hipLaunchKernelGGL(px_flatten_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, parent, n);
  hipLaunchKernelGGL(px_label_kernel, dim3(tiles), dim3(256), 0, ctx->stream, rgb, (int)H, (int)W, (int)radius, eps * eps, win, lin_lut,
                     (const int32_t*)parent, labels_out);
  RHCCQ_LAUNCH_CHECK(ctx);
  return 0;
}

}  // extern "C"
