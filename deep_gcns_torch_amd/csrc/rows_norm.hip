// Node-wise normalisation of (rows, C) feature matrices for gfx950: BatchNorm1d (training and eval) with an
// optional fused ReLU, forward and backward.
//
// Replaces nn.BatchNorm1d as returned by norm_layer('batch', C)   gcn_lib/sparse/torch_nn.py:23-34
// and the Lin -> BatchNorm1d -> ReLU run inside MLP               gcn_lib/sparse/torch_nn.py:50-71
// around GENConv (examples/ogb/ogbn_arxiv/model.py:90-106: norm -> relu -> dropout -> conv -> residual).
//
// All kernels are HBM streaming passes over row-major data; a thread owns one VEC-wide channel group, so the
// per-channel coefficients live in registers and every access is a coalesced 16-byte load:
//   forward  : rows_stats (read x)               -> partial sums  -> dgcn_bn_finalize_f32 (dense_bn.hip)
//              rows_bn_apply (read x, write y)      y = [relu](scale*x + shift)
//   backward : rows_bn_bwd_stats (read g, x[, y]) -> partial sums of g' and g'*xhat, g' = g*[y>0]
//              rows_bn_bwd_finalize                 dgamma, dbeta, the two batch-mean coefficients
//              rows_bn_bwd_apply (read g, x[, y], write dx)   dx = scale*(g' - c1 - xhat*c2)
// Sums are accumulated per workgroup in a fixed order and combined in fp64: deterministic.
//
// The pre-activation run  norm -> ReLU -> dropout  of the 'res+' blocks (examples/ogb/ogbn_arxiv/model.py:90-106) and of
// the reversible BasicBlock (eff_gcn_modules/rev/rev_layer.py:35-51) is ONE apply pass here: the dropout mask is either a
// counter hash of (seed, element index) -- regenerated, never stored -- or the shared mask tensor the reversible model
// hands to every layer; the backward recomputes the ReLU mask from x and the saved coefficients instead of reading y,
// and can add the gradient of a skip connection in its apply pass.

#include "dgcn_common.h"

namespace dgcn {
namespace {

// ---- dropout inside the row kernels ------------------------------------------------------------------------------
// mode 0: none.  mode 1: keep element i iff u16(i) >= thr, u16 = 16 bits of a murmur-style hash of (seed, i >> 2)
// (four elements per hash pair; keep probability 1 - thr / 65536, kept values scaled by inv_keep).  mode 2: multiply by
// mask[i] (SharedDropout: the model's own mask tensor, already scaled).
struct DropArgs {
  int mode;
  const float* mask;
  int64_t mld;          // row stride of mask (floats): a chunk view of the model's (N, hidden) mask is used in place
  uint32_t s0, s1, thr;
  float inv_keep;
};

__device__ __forceinline__ uint32_t drop_mix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}

// the four 16-bit draws of element group q (elements 4q .. 4q+3 of the flattened (rows, C) array)
__device__ __forceinline__ void drop_rand4(uint64_t q, uint32_t s0, uint32_t s1, uint32_t (&r)[4]) {
  const uint32_t lo = static_cast<uint32_t>(q), hi = static_cast<uint32_t>(q >> 32);
  uint32_t h0 = drop_mix32(lo ^ s0);
  h0 = drop_mix32(h0 + hi * 0x9E3779B1u + s1);
  const uint32_t h1 = drop_mix32(h0 ^ 0x68E31DA4u);
  r[0] = h0 & 0xffffu; r[1] = h0 >> 16; r[2] = h1 & 0xffffu; r[3] = h1 >> 16;
}

// factor[j] = what element (r, c + j) of the (rows, C) array is multiplied with; (r * C + c) % VEC == 0
template <int VEC>
__device__ __forceinline__ void drop_factors(const DropArgs& D, int64_t r, int c, int C, float (&f)[VEC]) {
#pragma unroll
  for (int j = 0; j < VEC; ++j) f[j] = 1.f;
  const int64_t base = r * C + c;
  if (D.mode == 1) {
    uint32_t r[4];
    drop_rand4(static_cast<uint64_t>(base) >> 2, D.s0, D.s1, r);
    if constexpr (VEC == 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) f[j] = (r[j] >= D.thr) ? D.inv_keep : 0.f;
    } else {
      f[0] = (r[base & 3] >= D.thr) ? D.inv_keep : 0.f;
    }
  } else if (D.mode == 2) {
    load_vec<VEC>(f, D.mask + r * D.mld + c);
  }
}

constexpr int kMaxParts = 512;
constexpr int kFinThreads = 1024;
constexpr int kFinTile = 64;
constexpr int kFinRows = kFinThreads / kFinTile;

struct RowsGeom {
  int cg;    // channel groups per row (C / VEC)
  int rpp;   // rows per pass of one workgroup (256 / cg)
};

__host__ __device__ inline RowsGeom rows_geom(int C, int vec) {
  RowsGeom g;
  g.cg = C / vec;
  g.rpp = kWgThreads / g.cg;
  return g;
}

// ---- forward statistics: partial[wg][0][c] = sum x, partial[wg][1][c] = sum x^2 over the workgroup's row slab
template <int VEC>
__global__ __launch_bounds__(kWgThreads) void rows_stats_kernel(const float* __restrict__ x, int64_t ld, int64_t rows,
                                                               int C, float* __restrict__ partial, int64_t slab) {
  __shared__ float red[2][kWgThreads * VEC];
  const RowsGeom G = rows_geom(C, VEC);
  const int cgi = threadIdx.x % G.cg;
  const int rl = threadIdx.x / G.cg;
  const int64_t r0 = blockIdx.x * slab;
  const int64_t r1 = min(rows, r0 + slab);
  float s[VEC], q[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { s[j] = 0.f; q[j] = 0.f; }
  if (rl < G.rpp) {
    const float* p = x + cgi * VEC;
    int64_t r = r0 + rl;
    for (; r + 3 * G.rpp < r1; r += 4 * G.rpp) {   // four independent loads in flight
      float a[4][VEC];
#pragma unroll
      for (int u = 0; u < 4; ++u) load_vec<VEC>(a[u], p + (r + static_cast<int64_t>(u) * G.rpp) * ld);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) { s[j] += a[u][j]; q[j] = fmaf(a[u][j], a[u][j], q[j]); }
      }
    }
    for (; r < r1; r += G.rpp) {
      float a[VEC];
      load_vec<VEC>(a, p + r * ld);
#pragma unroll
      for (int j = 0; j < VEC; ++j) { s[j] += a[j]; q[j] = fmaf(a[j], a[j], q[j]); }
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    red[0][(rl * G.cg + cgi) * VEC + j] = s[j];
    red[1][(rl * G.cg + cgi) * VEC + j] = q[j];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += kWgThreads) {
    float t0 = 0.f, t1 = 0.f;
    for (int k = 0; k < G.rpp; ++k) { t0 += red[0][k * C + c]; t1 += red[1][k * C + c]; }   // fixed order
    partial[(static_cast<int64_t>(blockIdx.x) * 2) * C + c] = t0;
    partial[(static_cast<int64_t>(blockIdx.x) * 2 + 1) * C + c] = t1;
  }
}

// ---- forward apply: y = [relu](scale*x + shift)
template <int VEC, bool RELU>
__global__ __launch_bounds__(kWgThreads) void rows_bn_apply_kernel(const float* __restrict__ x, int64_t ld,
                                                                  const float* __restrict__ bnbuf,
                                                                  float* __restrict__ y, int64_t rows, int C,
                                                                  const DropArgs D) {
  const RowsGeom G = rows_geom(C, VEC);
  const int cgi = threadIdx.x % G.cg;
  const int rl = threadIdx.x / G.cg;
  if (rl >= G.rpp) return;
  float sc[VEC], sh[VEC];
  load_vec<VEC>(sc, bnbuf + cgi * VEC);
  load_vec<VEC>(sh, bnbuf + C + cgi * VEC);
  const int64_t step = static_cast<int64_t>(gridDim.x) * G.rpp;
  for (int64_t r = static_cast<int64_t>(blockIdx.x) * G.rpp + rl; r < rows; r += step) {
    float a[VEC], o[VEC];
    load_vec<VEC>(a, x + r * ld + cgi * VEC);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      o[j] = fmaf(a[j], sc[j], sh[j]);
      if (RELU) o[j] = fmaxf(o[j], 0.f);
    }
    if (D.mode) {
      float f[VEC];
      drop_factors<VEC>(D, r, cgi * VEC, C, f);
#pragma unroll
      for (int j = 0; j < VEC; ++j) o[j] *= f[j];
    }
    store_vec<VEC>(y + r * C + cgi * VEC, o);
  }
}

// ---- backward statistics: partial[wg][0][c] = sum g', partial[wg][1][c] = sum g'*xhat
// g' = g * (dropout factor) * [ReLU mask].  RELU: 0 none, 1 mask = [y > 0] read from the forward output,
// 2 mask = [scale*x + shift > 0] recomputed from x (nothing but g and x is read).
template <int VEC, int RELU>
__global__ __launch_bounds__(kWgThreads) void rows_bn_bwd_stats_kernel(const float* __restrict__ g,
                                                                      const float* __restrict__ x, int64_t ld,
                                                                      const float* __restrict__ y,
                                                                      const float* __restrict__ bnbuf, int64_t rows,
                                                                      int C, float* __restrict__ partial,
                                                                      int64_t slab, const DropArgs D) {
  __shared__ float red[2][kWgThreads * VEC];
  const RowsGeom G = rows_geom(C, VEC);
  const int cgi = threadIdx.x % G.cg;
  const int rl = threadIdx.x / G.cg;
  const int64_t r0 = blockIdx.x * slab;
  const int64_t r1 = min(rows, r0 + slab);
  float s[VEC], q[VEC], mean[VEC], istd[VEC], sc[VEC], sh[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { s[j] = 0.f; q[j] = 0.f; mean[j] = 0.f; istd[j] = 0.f; sc[j] = 0.f; sh[j] = 0.f; }
  if (rl < G.rpp) {
    load_vec<VEC>(mean, bnbuf + 2 * C + cgi * VEC);
    load_vec<VEC>(istd, bnbuf + 3 * C + cgi * VEC);
    if (RELU == 2) {
      load_vec<VEC>(sc, bnbuf + cgi * VEC);
      load_vec<VEC>(sh, bnbuf + C + cgi * VEC);
    }
    for (int64_t r = r0 + rl; r < r1; r += 2 * G.rpp) {
      float gg[2][VEC], xx[2][VEC], yy[2][VEC], ff[2][VEC];
      bool ok[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int64_t rr = r + static_cast<int64_t>(u) * G.rpp;
        ok[u] = rr < r1;
#pragma unroll
        for (int j = 0; j < VEC; ++j) { gg[u][j] = 0.f; xx[u][j] = 0.f; yy[u][j] = 1.f; ff[u][j] = 1.f; }
        if (ok[u]) {
          load_vec<VEC>(gg[u], g + rr * C + cgi * VEC);
          load_vec<VEC>(xx[u], x + rr * ld + cgi * VEC);
          if (RELU == 1) load_vec<VEC>(yy[u], y + rr * C + cgi * VEC);
          if (D.mode) drop_factors<VEC>(D, rr, cgi * VEC, C, ff[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          bool dead = false;
          if (RELU == 1) dead = !(yy[u][j] > 0.f);
          if (RELU == 2) dead = !(fmaf(xx[u][j], sc[j], sh[j]) > 0.f);
          const float gp = dead ? 0.f : gg[u][j] * ff[u][j];
          const float xh = (xx[u][j] - mean[j]) * istd[j];
          s[j] += gp;
          q[j] = fmaf(gp, xh, q[j]);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    red[0][(rl * G.cg + cgi) * VEC + j] = s[j];
    red[1][(rl * G.cg + cgi) * VEC + j] = q[j];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += kWgThreads) {
    float t0 = 0.f, t1 = 0.f;
    for (int k = 0; k < G.rpp; ++k) { t0 += red[0][k * C + c]; t1 += red[1][k * C + c]; }
    partial[(static_cast<int64_t>(blockIdx.x) * 2) * C + c] = t0;
    partial[(static_cast<int64_t>(blockIdx.x) * 2 + 1) * C + c] = t1;
  }
}

// ---- backward finalize: coef[0] = dgamma = sum g'*xhat, coef[1] = dbeta = sum g', coef[2] = c1, coef[3] = c2
__global__ __launch_bounds__(kFinThreads) void rows_bn_bwd_finalize_kernel(const float* __restrict__ partial,
                                                                          int nparts, int C, double count,
                                                                          int training, float* __restrict__ coef) {
  __shared__ double red[kFinRows][2][kFinTile];
  const int cc = threadIdx.x % kFinTile;
  const int r = threadIdx.x / kFinTile;
  for (int c0 = 0; c0 < C; c0 += kFinTile) {
    const int c = c0 + cc;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
#pragma unroll 4
      for (int p = r; p < nparts; p += kFinRows) {
        s1 += static_cast<double>(partial[(static_cast<int64_t>(p) * 2) * C + c]);
        s2 += static_cast<double>(partial[(static_cast<int64_t>(p) * 2 + 1) * C + c]);
      }
    }
    red[r][0][cc] = s1;
    red[r][1][cc] = s2;
    __syncthreads();
    if (r == 0 && c < C) {
      double t1 = 0.0, t2 = 0.0;
      for (int q = 0; q < kFinRows; ++q) { t1 += red[q][0][cc]; t2 += red[q][1][cc]; }
      coef[c] = static_cast<float>(t2);
      coef[C + c] = static_cast<float>(t1);
      coef[2 * C + c] = training ? static_cast<float>(t1 / count) : 0.f;
      coef[3 * C + c] = training ? static_cast<float>(t2 / count) : 0.f;
    }
    __syncthreads();
  }
}

// ---- backward apply: dx = scale*(g' - c1 - xhat*c2) [+ gadd]   (gadd: the gradient of a skip connection around the block)
template <int VEC, int RELU>
__global__ __launch_bounds__(kWgThreads) void rows_bn_bwd_apply_kernel(const float* __restrict__ g,
                                                                      const float* __restrict__ x, int64_t ld,
                                                                      const float* __restrict__ y,
                                                                      const float* __restrict__ bnbuf,
                                                                      const float* __restrict__ coef,
                                                                      float* __restrict__ dx, int64_t rows, int C,
                                                                      const DropArgs D,
                                                                      const float* __restrict__ gadd) {
  const RowsGeom G = rows_geom(C, VEC);
  const int cgi = threadIdx.x % G.cg;
  const int rl = threadIdx.x / G.cg;
  if (rl >= G.rpp) return;
  float sc[VEC], sh[VEC], mean[VEC], istd[VEC], c1[VEC], c2[VEC];
  load_vec<VEC>(sc, bnbuf + cgi * VEC);
  load_vec<VEC>(sh, bnbuf + C + cgi * VEC);
  load_vec<VEC>(mean, bnbuf + 2 * C + cgi * VEC);
  load_vec<VEC>(istd, bnbuf + 3 * C + cgi * VEC);
  load_vec<VEC>(c1, coef + 2 * C + cgi * VEC);
  load_vec<VEC>(c2, coef + 3 * C + cgi * VEC);
  const int64_t step = static_cast<int64_t>(gridDim.x) * G.rpp;
  for (int64_t r = static_cast<int64_t>(blockIdx.x) * G.rpp + rl; r < rows; r += step) {
    float gg[VEC], xx[VEC], yy[VEC], ff[VEC], ga[VEC], o[VEC];
    load_vec<VEC>(gg, g + r * C + cgi * VEC);
    load_vec<VEC>(xx, x + r * ld + cgi * VEC);
#pragma unroll
    for (int j = 0; j < VEC; ++j) { yy[j] = 1.f; ff[j] = 1.f; ga[j] = 0.f; }
    if (RELU == 1) load_vec<VEC>(yy, y + r * C + cgi * VEC);
    if (gadd) load_vec<VEC>(ga, gadd + r * C + cgi * VEC);
    if (D.mode) drop_factors<VEC>(D, r, cgi * VEC, C, ff);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      bool dead = false;
      if (RELU == 1) dead = !(yy[j] > 0.f);
      if (RELU == 2) dead = !(fmaf(xx[j], sc[j], sh[j]) > 0.f);
      const float gp = dead ? 0.f : gg[j] * ff[j];
      const float xh = (xx[j] - mean[j]) * istd[j];
      o[j] = sc[j] * (gp - c1[j] - xh * c2[j]) + ga[j];
    }
    store_vec<VEC>(dx + r * C + cgi * VEC, o);
  }
}

int pick_vec(int C, int64_t ld, const void* a, const void* b, const void* c, const void* d) {
  auto al = [](const void* p) { return !p || (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  if (C % 4 == 0 && C / 4 <= kWgThreads && ld % 4 == 0 && al(a) && al(b) && al(c) && al(d)) return 4;
  return 1;
}

int stream_grid(int64_t rows, int rpp) {
  int64_t wgs = (rows + rpp - 1) / rpp;
  const int64_t cap = static_cast<int64_t>(num_cus()) * 16;
  if (wgs > cap) wgs = cap;
  return static_cast<int>(wgs < 1 ? 1 : wgs);
}

}  // namespace
}  // namespace dgcn

using namespace dgcn;

extern "C" int32_t dgcn_rows_num_partials(int64_t rows, int32_t C) {
  if (rows <= 0 || C <= 0) return 0;
  const int64_t want = (rows + 63) / 64;   // at least 64 rows per workgroup
  return static_cast<int32_t>(want < kMaxParts ? want : kMaxParts);
}

extern "C" int dgcn_rows_stats_f32(const float* x, int64_t ld, int64_t rows, int32_t C, float* partial,
                                   void* stream) {
  if (!x || !partial) return DGCN_E_NULL;
  if (rows <= 0 || C <= 0 || ld < C) return DGCN_E_SHAPE;
  const int vec = pick_vec(C, ld, x, nullptr, nullptr, nullptr);
  if (vec == 1 && C > kWgThreads) return DGCN_E_SHAPE;
  const int nparts = dgcn_rows_num_partials(rows, C);
  const int64_t slab = (rows + nparts - 1) / nparts;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (vec == 4) {
    hipLaunchKernelGGL(rows_stats_kernel<4>, dim3(nparts), dim3(kWgThreads), 0, s, x, ld, rows, C, partial, slab);
  } else {
    hipLaunchKernelGGL(rows_stats_kernel<1>, dim3(nparts), dim3(kWgThreads), 0, s, x, ld, rows, C, partial, slab);
  }
  return launch_status();
}

namespace dgcn {
namespace {

inline DropArgs drop_none() { return DropArgs{0, nullptr, 0, 0u, 0u, 0u, 1.f}; }

// drop_mode 1: seed = (s0, s1), thr in [0, 65535] (drop probability thr / 65536); drop_mode 2: mask (rows, C), row stride mld
inline int drop_make(int32_t drop_mode, const float* mask, int64_t mld, uint32_t s0, uint32_t s1, uint32_t thr,
                     int32_t C, DropArgs* D) {
  *D = drop_none();
  if (drop_mode == 0) return DGCN_OK;
  if (drop_mode == 1) {
    if (thr > 65535u) return DGCN_E_SHAPE;
    D->mode = 1; D->s0 = s0; D->s1 = s1; D->thr = thr;
    D->inv_keep = 65536.f / static_cast<float>(65536u - thr);
    return DGCN_OK;
  }
  if (drop_mode == 2) {
    if (!mask) return DGCN_E_NULL;
    if (mld < C) return DGCN_E_SHAPE;
    D->mode = 2; D->mask = mask; D->mld = mld;
    return DGCN_OK;
  }
  return DGCN_E_MODE;
}

int rows_bn_apply_impl(const float* x, int64_t ld, const float* bnbuf, int32_t relu, float* y, int64_t rows,
                       int32_t C, const DropArgs& D, void* stream) {
  if (!x || !bnbuf || !y) return DGCN_E_NULL;
  if (rows < 0 || C <= 0 || ld < C) return DGCN_E_SHAPE;
  if (rows == 0) return DGCN_OK;
  int vec = pick_vec(C, ld, x, y, bnbuf, D.mask);
  if (vec == 4 && D.mode == 2 && D.mld % 4 != 0) vec = 1;
  if (vec == 1 && C > kWgThreads) return DGCN_E_SHAPE;
  const RowsGeom G = rows_geom(C, vec);
  const int grid = stream_grid(rows, G.rpp);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (vec == 4) {
    if (relu) hipLaunchKernelGGL((rows_bn_apply_kernel<4, true>), dim3(grid), dim3(kWgThreads), 0, s, x, ld, bnbuf, y, rows, C, D);
    else hipLaunchKernelGGL((rows_bn_apply_kernel<4, false>), dim3(grid), dim3(kWgThreads), 0, s, x, ld, bnbuf, y, rows, C, D);
  } else {
    if (relu) hipLaunchKernelGGL((rows_bn_apply_kernel<1, true>), dim3(grid), dim3(kWgThreads), 0, s, x, ld, bnbuf, y, rows, C, D);
    else hipLaunchKernelGGL((rows_bn_apply_kernel<1, false>), dim3(grid), dim3(kWgThreads), 0, s, x, ld, bnbuf, y, rows, C, D);
  }
  return launch_status();
}

// relu_mode: 0 none, 1 from y, 2 recomputed from x
int rows_bn_bwd_stats_impl(const float* g, const float* x, int64_t ld, const float* y, const float* bnbuf,
                           float* partial, int64_t rows, int32_t C, int relu_mode, const DropArgs& D, void* stream) {
  if (!g || !x || !bnbuf || !partial || (relu_mode == 1 && !y)) return DGCN_E_NULL;
  if (rows <= 0 || C <= 0 || ld < C) return DGCN_E_SHAPE;
  int vec = pick_vec(C, ld, x, g, y, bnbuf);
  if (vec == 4 && D.mask && ((reinterpret_cast<uintptr_t>(D.mask) & 15u) || D.mld % 4 != 0)) vec = 1;
  if (vec == 1 && C > kWgThreads) return DGCN_E_SHAPE;
  const int nparts = dgcn_rows_num_partials(rows, C);
  const int64_t slab = (rows + nparts - 1) / nparts;
  hipStream_t s = static_cast<hipStream_t>(stream);
#define DGCN_BN_STATS(V, R) hipLaunchKernelGGL((rows_bn_bwd_stats_kernel<V, R>), dim3(nparts), dim3(kWgThreads), 0, s, g, x, ld, y, bnbuf, rows, C, partial, slab, D)
  if (vec == 4) {
    if (relu_mode == 1) DGCN_BN_STATS(4, 1); else if (relu_mode == 2) DGCN_BN_STATS(4, 2); else DGCN_BN_STATS(4, 0);
  } else {
    if (relu_mode == 1) DGCN_BN_STATS(1, 1); else if (relu_mode == 2) DGCN_BN_STATS(1, 2); else DGCN_BN_STATS(1, 0);
  }
#undef DGCN_BN_STATS
  return launch_status();
}

int rows_bn_bwd_apply_impl(const float* g, const float* x, int64_t ld, const float* y, const float* bnbuf,
                           const float* coef, float* dx, int64_t rows, int32_t C, int relu_mode, const DropArgs& D,
                           const float* gadd, void* stream) {
  if (!g || !x || !bnbuf || !coef || !dx || (relu_mode == 1 && !y)) return DGCN_E_NULL;
  if (rows < 0 || C <= 0 || ld < C) return DGCN_E_SHAPE;
  if (rows == 0) return DGCN_OK;
  int vec = pick_vec(C, ld, x, g, y, dx);
  auto mis = [](const void* p) { return p && (reinterpret_cast<uintptr_t>(p) & 15u); };
  if (vec == 4 && (mis(bnbuf) || mis(coef) || mis(D.mask) || mis(gadd) || (D.mode == 2 && D.mld % 4 != 0))) vec = 1;
  if (vec == 1 && C > kWgThreads) return DGCN_E_SHAPE;
  const RowsGeom G = rows_geom(C, vec);
  const int grid = stream_grid(rows, G.rpp);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define DGCN_BN_APPLY(V, R) hipLaunchKernelGGL((rows_bn_bwd_apply_kernel<V, R>), dim3(grid), dim3(kWgThreads), 0, s, g, x, ld, y, bnbuf, coef, dx, rows, C, D, gadd)
  if (vec == 4) {
    if (relu_mode == 1) DGCN_BN_APPLY(4, 1); else if (relu_mode == 2) DGCN_BN_APPLY(4, 2); else DGCN_BN_APPLY(4, 0);
  } else {
    if (relu_mode == 1) DGCN_BN_APPLY(1, 1); else if (relu_mode == 2) DGCN_BN_APPLY(1, 2); else DGCN_BN_APPLY(1, 0);
  }
#undef DGCN_BN_APPLY
  return launch_status();
}

}  // namespace
}  // namespace dgcn

extern "C" int dgcn_rows_bn_apply_f32(const float* x, int64_t ld, const float* bnbuf, int32_t relu, float* y,
                                      int64_t rows, int32_t C, void* stream) {
  return rows_bn_apply_impl(x, ld, bnbuf, relu, y, rows, C, drop_none(), stream);
}

extern "C" int dgcn_rows_bn_bwd_stats_f32(const float* g, const float* x, int64_t ld, const float* y,
                                          const float* bnbuf, float* partial, int64_t rows, int32_t C,
                                          void* stream) {
  return rows_bn_bwd_stats_impl(g, x, ld, y, bnbuf, partial, rows, C, y ? 1 : 0, drop_none(), stream);
}

// ---- norm -> [ReLU] -> [dropout] in one apply pass, and its backward (ReLU mask recomputed from x) ----
extern "C" int dgcn_rows_bn_act_apply_f32(const float* x, int64_t ld, const float* bnbuf, int32_t relu,
                                          int32_t drop_mode, const float* drop_mask, int64_t drop_mask_ld, uint32_t seed0, uint32_t seed1,
                                          uint32_t drop_thr, float* y, int64_t rows, int32_t C, void* stream) {
  DropArgs D;
  if (const int rc = drop_make(drop_mode, drop_mask, drop_mask_ld, seed0, seed1, drop_thr, C, &D)) return rc;
  return rows_bn_apply_impl(x, ld, bnbuf, relu, y, rows, C, D, stream);
}

extern "C" int dgcn_rows_bn_act_bwd_stats_f32(const float* g, const float* x, int64_t ld, const float* bnbuf,
                                              int32_t relu, int32_t drop_mode, const float* drop_mask, int64_t drop_mask_ld,
                                              uint32_t seed0, uint32_t seed1, uint32_t drop_thr, float* partial,
                                              int64_t rows, int32_t C, void* stream) {
  DropArgs D;
  if (const int rc = drop_make(drop_mode, drop_mask, drop_mask_ld, seed0, seed1, drop_thr, C, &D)) return rc;
  return rows_bn_bwd_stats_impl(g, x, ld, nullptr, bnbuf, partial, rows, C, relu ? 2 : 0, D, stream);
}

extern "C" int dgcn_rows_bn_act_bwd_apply_f32(const float* g, const float* x, int64_t ld, const float* bnbuf,
                                              const float* coef, int32_t relu, int32_t drop_mode,
                                              const float* drop_mask, int64_t drop_mask_ld, uint32_t seed0, uint32_t seed1,
                                              uint32_t drop_thr, const float* gadd, float* dx, int64_t rows,
                                              int32_t C, void* stream) {
  DropArgs D;
  if (const int rc = drop_make(drop_mode, drop_mask, drop_mask_ld, seed0, seed1, drop_thr, C, &D)) return rc;
  return rows_bn_bwd_apply_impl(g, x, ld, nullptr, bnbuf, coef, dx, rows, C, relu ? 2 : 0, D, gadd, stream);
}

extern "C" int dgcn_rows_bn_bwd_finalize_f32(const float* partial, int32_t nparts, int32_t C, double count,
                                             int32_t training, float* coef, void* stream) {
  if (!partial || !coef) return DGCN_E_NULL;
  if (C <= 0 || nparts <= 0 || count <= 0.0) return DGCN_E_SHAPE;
  hipLaunchKernelGGL(rows_bn_bwd_finalize_kernel, dim3(1), dim3(kFinThreads), 0, static_cast<hipStream_t>(stream),
                     partial, nparts, C, count, training, coef);
  return launch_status();
}

extern "C" int dgcn_rows_bn_bwd_apply_f32(const float* g, const float* x, int64_t ld, const float* y,
                                          const float* bnbuf, const float* coef, float* dx, int64_t rows,
                                          int32_t C, void* stream) {
  return rows_bn_bwd_apply_impl(g, x, ld, y, bnbuf, coef, dx, rows, C, y ? 1 : 0, drop_none(), nullptr, stream);
}

// =====================================================================================================
// LayerNorm over the channel dimension of (rows, C) features, optional fused ReLU.
// Replaces nn.LayerNorm from norm_layer('layer', C)  (gcn_lib/sparse/torch_nn.py:23-34; the default norm of the
// ogbn-proteins / ogbg-ppa / RevGCN configurations) and the Lin -> LayerNorm -> ReLU run of MLP (:50-71).
// A row lives in the registers of LPR lanes (Q float4 each), 64/LPR rows per wave; mean and variance are two
// register passes (no E[x^2]-E[x]^2 cancellation), reductions are xor-shuffles inside the row's lanes.
//   forward : y = [relu]((x - mean) * rstd * gamma + beta); mean, rstd saved per row        (1 read + 1 write)
//   backward: gh = g' * gamma;  dx = rstd * (gh - mean_c(gh) - xhat * mean_c(gh * xhat))   (2-3 reads + 1 write)
//             dgamma = sum_rows g' * xhat, dbeta = sum_rows g' as per-workgroup partials [nparts][2][C]
//             (slot 0 = sum g', slot 1 = sum g' xhat; the caller sums the <= 1024 partials).
// Stock kernels run at ~1 TB/s on this shape class; these are streaming passes.
// =====================================================================================================
namespace dgcn {
namespace {

constexpr int kLnMaxParts = 1024;

template <int LPR>
__device__ __forceinline__ float row_sum(float v) {
#pragma unroll
  for (int off = 1; off < LPR; off <<= 1) v += __shfl_xor(v, off);
  return v;
}

struct LnGeom {
  int lpr;   // lanes per row (power of two, <= 64)
  int q;     // float4 per lane (1, 2 or 4)
};

inline LnGeom ln_geom(int C) {
  const int quads = C / 4;
  LnGeom g;
  g.lpr = 4;
  while (g.lpr < quads && g.lpr < kWave) g.lpr <<= 1;
  const int need = (quads + g.lpr - 1) / g.lpr;
  g.q = need <= 1 ? 1 : (need <= 2 ? 2 : 4);
  return g;
}

template <int LPR, int Q, bool RELU>
__global__ __launch_bounds__(kWgThreads) void rows_ln_fwd_kernel(const float* __restrict__ x, int64_t ld,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float eps,
                                                                float* __restrict__ y, float* __restrict__ mean_out,
                                                                float* __restrict__ rstd_out, int64_t rows, int C,
                                                                const DropArgs D) {
  constexpr int RPW = kWave / LPR;
  const int lane = lane_id();
  const int sub = lane / LPR, li = lane % LPR;
  const int64_t wave = static_cast<int64_t>(blockIdx.x) * kWavesPerWg + (threadIdx.x >> 6);
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerWg;
  float gm[Q][4], bt[Q][4];
  bool ok[Q];
#pragma unroll
  for (int k = 0; k < Q; ++k) {
    const int c = (li + k * LPR) * 4;
    ok[k] = c < C;
#pragma unroll
    for (int j = 0; j < 4; ++j) { gm[k][j] = 1.f; bt[k][j] = 0.f; }
    if (ok[k] && gamma) load_vec<4>(gm[k], gamma + c);
    if (ok[k] && beta) load_vec<4>(bt[k], beta + c);
  }
  const float inv_c = 1.f / static_cast<float>(C);
  for (int64_t r0 = wave * RPW; r0 < rows; r0 += nwaves * RPW) {
    const int64_t r = r0 + sub;
    const bool live = r < rows;
    float v[Q][4];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < Q; ++k) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[k][j] = 0.f;
      if (live && ok[k]) load_vec<4>(v[k], x + r * ld + (li + k * LPR) * 4);
      s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
    }
    const float mean = row_sum<LPR>(s) * inv_c;
    float q2 = 0.f;
#pragma unroll
    for (int k = 0; k < Q; ++k) {
      if (ok[k]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = v[k][j] - mean; q2 = fmaf(d, d, q2); }
      }
    }
    const float var = row_sum<LPR>(q2) * inv_c;
    const float rstd = rsqrtf(var + eps);
    if (live) {
#pragma unroll
      for (int k = 0; k < Q; ++k) {
        if (ok[k]) {
          float o[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            o[j] = fmaf((v[k][j] - mean) * rstd, gm[k][j], bt[k][j]);
            if (RELU) o[j] = fmaxf(o[j], 0.f);
          }
          if (D.mode) {
            float f[4];
            drop_factors<4>(D, r, (li + k * LPR) * 4, C, f);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] *= f[j];
          }
          store_vec<4>(y + r * C + (li + k * LPR) * 4, o);
        }
      }
      if (li == 0) { mean_out[r] = mean; rstd_out[r] = rstd; }
    }
  }
}

// RELU here: true = some ReLU mask applies; it is [y > 0] when y is given, else [xhat*gamma + beta > 0] recomputed
// from x (beta may be null = 0).  D: dropout factors multiplied into g.  gadd: gradient of a skip connection.
template <int LPR, int Q, bool RELU>
__global__ __launch_bounds__(kWgThreads) void rows_ln_bwd_kernel(const float* __restrict__ g,
                                                                const float* __restrict__ x, int64_t ld,
                                                                const float* __restrict__ y,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ mean_in,
                                                                const float* __restrict__ rstd_in,
                                                                float* __restrict__ dx, float* __restrict__ partial,
                                                                int64_t rows, int C,
                                                                const float* __restrict__ beta, const DropArgs D,
                                                                const float* __restrict__ gadd) {
  constexpr int RPW = kWave / LPR;
  __shared__ float red[kWavesPerWg][2][LPR * Q * 4];
  const int lane = lane_id();
  const int sub = lane / LPR, li = lane % LPR;
  const int wv = threadIdx.x >> 6;
  const int64_t wave = static_cast<int64_t>(blockIdx.x) * kWavesPerWg + wv;
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerWg;
  float gm[Q][4], bt[Q][4], dg[Q][4], db[Q][4];
  bool ok[Q];
#pragma unroll
  for (int k = 0; k < Q; ++k) {
    const int c = (li + k * LPR) * 4;
    ok[k] = c < C;
#pragma unroll
    for (int j = 0; j < 4; ++j) { gm[k][j] = 1.f; bt[k][j] = 0.f; dg[k][j] = 0.f; db[k][j] = 0.f; }
    if (ok[k] && gamma) load_vec<4>(gm[k], gamma + c);
    if (RELU && !y && ok[k] && beta) load_vec<4>(bt[k], beta + c);
  }
  const float inv_c = 1.f / static_cast<float>(C);
  for (int64_t r0 = wave * RPW; r0 < rows; r0 += nwaves * RPW) {
    const int64_t r = r0 + sub;
    const bool live = r < rows;
    const float mean = live ? mean_in[r] : 0.f;
    const float rstd = live ? rstd_in[r] : 0.f;
    float gp[Q][4], xh[Q][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < Q; ++k) {
      float xv[4] = {0.f, 0.f, 0.f, 0.f}, yv[4] = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) gp[k][j] = 0.f;
      if (live && ok[k]) {
        const int c = (li + k * LPR) * 4;
        load_vec<4>(gp[k], g + r * C + c);
        load_vec<4>(xv, x + r * ld + c);
        if (RELU && y) load_vec<4>(yv, y + r * C + c);
        if (D.mode) {
          float f[4];
          drop_factors<4>(D, r, c, C, f);
#pragma unroll
          for (int j = 0; j < 4; ++j) gp[k][j] *= f[j];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xh[k][j] = (live && ok[k]) ? (xv[j] - mean) * rstd : 0.f;
        if (RELU) {
          const bool pos = y ? (yv[j] > 0.f) : (fmaf(xh[k][j], gm[k][j], bt[k][j]) > 0.f);
          if (!pos) gp[k][j] = 0.f;
        }
        const float gh = gp[k][j] * gm[k][j];
        s1 += gh;
        s2 = fmaf(gh, xh[k][j], s2);
        db[k][j] += gp[k][j];
        dg[k][j] = fmaf(gp[k][j], xh[k][j], dg[k][j]);
      }
    }
    const float m1 = row_sum<LPR>(s1) * inv_c;
    const float m2 = row_sum<LPR>(s2) * inv_c;
    if (live && dx) {
#pragma unroll
      for (int k = 0; k < Q; ++k) {
        if (ok[k]) {
          float o[4], ga[4] = {0.f, 0.f, 0.f, 0.f};
          if (gadd) load_vec<4>(ga, gadd + r * C + (li + k * LPR) * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = rstd * (gp[k][j] * gm[k][j] - m1 - xh[k][j] * m2) + ga[j];
          store_vec<4>(dx + r * C + (li + k * LPR) * 4, o);
        }
      }
    }
  }
  if (partial) {
    // lanes with the same channels (the RPW sub-rows of the wave), then the four waves, in a fixed order
#pragma unroll
    for (int k = 0; k < Q; ++k) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int off = LPR; off < kWave; off <<= 1) {
          db[k][j] += __shfl_xor(db[k][j], off);
          dg[k][j] += __shfl_xor(dg[k][j], off);
        }
        if (lane < LPR) {
          red[wv][0][(k * LPR + li) * 4 + j] = db[k][j];
          red[wv][1][(k * LPR + li) * 4 + j] = dg[k][j];
        }
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < LPR * Q * 4; i += kWgThreads) {
      const int k = i / (LPR * 4), rem = i % (LPR * 4);
      const int c = ((rem / 4) + k * LPR) * 4 + (rem % 4);
      if (c < C) {
        partial[(static_cast<int64_t>(blockIdx.x) * 2) * C + c] = ((red[0][0][i] + red[1][0][i]) + red[2][0][i]) + red[3][0][i];
        partial[(static_cast<int64_t>(blockIdx.x) * 2 + 1) * C + c] = ((red[0][1][i] + red[1][1][i]) + red[2][1][i]) + red[3][1][i];
      }
    }
  }
}

int ln_grid(int64_t rows, int lpr) {
  const int rpw = kWave / lpr;
  int64_t wgs = (rows + static_cast<int64_t>(rpw) * kWavesPerWg - 1) / (static_cast<int64_t>(rpw) * kWavesPerWg);
  if (wgs > kLnMaxParts) wgs = kLnMaxParts;
  return static_cast<int>(wgs < 1 ? 1 : wgs);
}

bool ln_ok(int C, int64_t ld, const void* a, const void* b, const void* c, const void* d, const void* e) {
  auto al = [](const void* p) { return !p || (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  return C > 0 && C % 4 == 0 && C <= 1024 && ld % 4 == 0 && ld >= C && al(a) && al(b) && al(c) && al(d) && al(e);
}

#define DGCN_LN_DISPATCH(KERNEL, RELU_V, ...)                                                                      \
  do {                                                                                                              \
    const LnGeom G = ln_geom(C);                                                                                    \
    const int grid = ln_grid(rows, G.lpr);                                                                          \
    const dim3 gd(grid), bd(kWgThreads);                                                                            \
    hipStream_t s = static_cast<hipStream_t>(stream);                                                               \
    if (G.q == 1) {                                                                                                 \
      switch (G.lpr) {                                                                                              \
        case 4: hipLaunchKernelGGL((KERNEL<4, 1, RELU_V>), gd, bd, 0, s, __VA_ARGS__); break;                       \
        case 8: hipLaunchKernelGGL((KERNEL<8, 1, RELU_V>), gd, bd, 0, s, __VA_ARGS__); break;                       \
        case 16: hipLaunchKernelGGL((KERNEL<16, 1, RELU_V>), gd, bd, 0, s, __VA_ARGS__); break;                     \
        case 32: hipLaunchKernelGGL((KERNEL<32, 1, RELU_V>), gd, bd, 0, s, __VA_ARGS__); break;                     \
        default: hipLaunchKernelGGL((KERNEL<64, 1, RELU_V>), gd, bd, 0, s, __VA_ARGS__); break;                     \
      }                                                                                                             \
    } else if (G.q == 2) {                                                                                          \
      hipLaunchKernelGGL((KERNEL<64, 2, RELU_V>), gd, bd, 0, s, __VA_ARGS__);                                       \
    } else {                                                                                                        \
      hipLaunchKernelGGL((KERNEL<64, 4, RELU_V>), gd, bd, 0, s, __VA_ARGS__);                                       \
    }                                                                                                               \
  } while (0)

}  // namespace
}  // namespace dgcn

extern "C" int32_t dgcn_rows_ln_num_partials(int64_t rows, int32_t C) {
  if (rows <= 0 || C <= 0 || C % 4 != 0 || C > 1024) return 0;
  return ln_grid(rows, ln_geom(C).lpr);
}

namespace dgcn {
namespace {

int rows_ln_fwd_impl(const float* x, int64_t ld, const float* gamma, const float* beta, float eps, int32_t relu,
                     float* y, float* mean, float* rstd, int64_t rows, int32_t C, const DropArgs& D, void* stream) {
  if (!x || !y || !mean || !rstd) return DGCN_E_NULL;
  if (rows < 0) return DGCN_E_SHAPE;
  if (!ln_ok(C, ld, x, y, gamma, beta, D.mask) || (D.mode == 2 && D.mld % 4 != 0)) {
    return (C > 0 && C % 4 == 0 && C <= 1024 && ld >= C) ? DGCN_E_ALIGN : DGCN_E_SHAPE;
  }
  if (rows == 0) return DGCN_OK;
  if (relu) DGCN_LN_DISPATCH(rows_ln_fwd_kernel, true, x, ld, gamma, beta, eps, y, mean, rstd, rows, C, D);
  else DGCN_LN_DISPATCH(rows_ln_fwd_kernel, false, x, ld, gamma, beta, eps, y, mean, rstd, rows, C, D);
  return launch_status();
}

int rows_ln_bwd_impl(const float* g, const float* x, int64_t ld, const float* y, const float* gamma,
                     const float* beta, const float* mean, const float* rstd, int32_t relu, const DropArgs& D,
                     const float* gadd, float* dx, float* partial, int64_t rows, int32_t C, void* stream) {
  if (!g || !x || !mean || !rstd || (!dx && !partial)) return DGCN_E_NULL;
  if (rows <= 0) return DGCN_E_SHAPE;
  auto mis = [](const void* p) { return p && (reinterpret_cast<uintptr_t>(p) & 15u); };
  if (!ln_ok(C, ld, x, g, y, gamma, dx) || mis(beta) || mis(D.mask) || mis(gadd) || (D.mode == 2 && D.mld % 4 != 0)) {
    return (C > 0 && C % 4 == 0 && C <= 1024 && ld >= C) ? DGCN_E_ALIGN : DGCN_E_SHAPE;
  }
  if (relu) DGCN_LN_DISPATCH(rows_ln_bwd_kernel, true, g, x, ld, y, gamma, mean, rstd, dx, partial, rows, C, beta, D, gadd);
  else DGCN_LN_DISPATCH(rows_ln_bwd_kernel, false, g, x, ld, y, gamma, mean, rstd, dx, partial, rows, C, beta, D, gadd);
  return launch_status();
}

}  // namespace
}  // namespace dgcn

extern "C" int dgcn_rows_ln_fwd_f32(const float* x, int64_t ld, const float* gamma, const float* beta, float eps,
                                    int32_t relu, float* y, float* mean, float* rstd, int64_t rows, int32_t C,
                                    void* stream) {
  return rows_ln_fwd_impl(x, ld, gamma, beta, eps, relu, y, mean, rstd, rows, C, drop_none(), stream);
}

extern "C" int dgcn_rows_ln_bwd_f32(const float* g, const float* x, int64_t ld, const float* y, const float* gamma,
                                    const float* mean, const float* rstd, float* dx, float* partial, int64_t rows,
                                    int32_t C, void* stream) {
  return rows_ln_bwd_impl(g, x, ld, y, gamma, nullptr, mean, rstd, y ? 1 : 0, drop_none(), nullptr, dx, partial, rows,
                          C, stream);
}

// ---- LayerNorm -> [ReLU] -> [dropout] in one pass; the backward recomputes the ReLU mask from x ----
extern "C" int dgcn_rows_ln_act_fwd_f32(const float* x, int64_t ld, const float* gamma, const float* beta, float eps,
                                        int32_t relu, int32_t drop_mode, const float* drop_mask, int64_t drop_mask_ld, uint32_t seed0,
                                        uint32_t seed1, uint32_t drop_thr, float* y, float* mean, float* rstd,
                                        int64_t rows, int32_t C, void* stream) {
  DropArgs D;
  if (const int rc = drop_make(drop_mode, drop_mask, drop_mask_ld, seed0, seed1, drop_thr, C, &D)) return rc;
  return rows_ln_fwd_impl(x, ld, gamma, beta, eps, relu, y, mean, rstd, rows, C, D, stream);
}

extern "C" int dgcn_rows_ln_act_bwd_f32(const float* g, const float* x, int64_t ld, const float* gamma,
                                        const float* beta, const float* mean, const float* rstd, int32_t relu,
                                        int32_t drop_mode, const float* drop_mask, int64_t drop_mask_ld, uint32_t seed0, uint32_t seed1,
                                        uint32_t drop_thr, const float* gadd, float* dx, float* partial,
                                        int64_t rows, int32_t C, void* stream) {
  DropArgs D;
  if (const int rc = drop_make(drop_mode, drop_mask, drop_mask_ld, seed0, seed1, drop_thr, C, &D)) return rc;
  return rows_ln_bwd_impl(g, x, ld, nullptr, gamma, beta, mean, rstd, relu, D, gadd, dx, partial, rows, C, stream);
}

// =====================================================================================================
// MsgNorm (gcn_lib/sparse/torch_message.py:88-99) fused with the residual of GENConv.forward (torch_vertex.py:70-74):
//   y_r = [x_r +] m_r / max(||m_r||_2, 1e-12) * ||x_r||_2 * s          s = msg_scale (device scalar)
// One pass forward (read x, m; write y), one pass backward (read g, x, m; write dx, dm; per-workgroup partial ds);
// the stock composition is 7 elementwise / reduction kernels forward and about twice that backward.
//   u = m / b, b = max(||m||, eps), a = ||x||, q = <u, g>
//   dm = (a s / b) (g - u q)   (only the g term when ||m|| <= eps: F.normalize clamps the norm, no gradient through it)
//   dx = [g +] s q x / a       (0 when a = 0)
//   ds = sum_r a q
// =====================================================================================================
namespace dgcn {
namespace {

constexpr float kMsgNormEps = 1e-12f;   // F.normalize default

template <int LPR, int Q, bool ADD_X>
__global__ __launch_bounds__(kWgThreads) void rows_msgnorm_fwd_kernel(const float* __restrict__ x, int64_t ldx,
                                                                     const float* __restrict__ m,
                                                                     const float* __restrict__ scale,
                                                                     float* __restrict__ y, int64_t rows, int C) {
  constexpr int RPW = kWave / LPR;
  const int lane = lane_id();
  const int sub = lane / LPR, li = lane % LPR;
  const int64_t wave = static_cast<int64_t>(blockIdx.x) * kWavesPerWg + (threadIdx.x >> 6);
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerWg;
  const float s = *scale;
  for (int64_t r0 = wave * RPW; r0 < rows; r0 += nwaves * RPW) {
    const int64_t r = r0 + sub;
    const bool live = r < rows;
    float xv[Q][4], mv[Q][4];
    float sx = 0.f, sm = 0.f;
#pragma unroll
    for (int k = 0; k < Q; ++k) {
      const int c = (li + k * LPR) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) { xv[k][j] = 0.f; mv[k][j] = 0.f; }
      if (live && c < C) {
        load_vec<4>(xv[k], x + r * ldx + c);
        load_vec<4>(mv[k], m + r * C + c);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) { sx = fmaf(xv[k][j], xv[k][j], sx); sm = fmaf(mv[k][j], mv[k][j], sm); }
    }
    const float a = sqrtf(row_sum<LPR>(sx));
    const float b = fmaxf(sqrtf(row_sum<LPR>(sm)), kMsgNormEps);
    const float f = a * s / b;
    if (live) {
#pragma unroll
      for (int k = 0; k < Q; ++k) {
        const int c = (li + k * LPR) * 4;
        if (c < C) {
          float o[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = ADD_X ? fmaf(mv[k][j], f, xv[k][j]) : mv[k][j] * f;
          store_vec<4>(y + r * C + c, o);
        }
      }
    }
  }
}

template <int LPR, int Q, bool ADD_X>
__global__ __launch_bounds__(kWgThreads) void rows_msgnorm_bwd_kernel(const float* __restrict__ g,
                                                                     const float* __restrict__ x, int64_t ldx,
                                                                     const float* __restrict__ m,
                                                                     const float* __restrict__ scale,
                                                                     float* __restrict__ dx, float* __restrict__ dm,
                                                                     float* __restrict__ ds_partial, int64_t rows,
                                                                     int C) {
  constexpr int RPW = kWave / LPR;
  __shared__ float red[kWavesPerWg];
  const int lane = lane_id();
  const int sub = lane / LPR, li = lane % LPR;
  const int wv = threadIdx.x >> 6;
  const int64_t wave = static_cast<int64_t>(blockIdx.x) * kWavesPerWg + wv;
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerWg;
  const float s = *scale;
  float ds = 0.f;
  for (int64_t r0 = wave * RPW; r0 < rows; r0 += nwaves * RPW) {
    const int64_t r = r0 + sub;
    const bool live = r < rows;
    float xv[Q][4], mv[Q][4], gv[Q][4];
    float sx = 0.f, sm = 0.f, mg = 0.f;
#pragma unroll
    for (int k = 0; k < Q; ++k) {
      const int c = (li + k * LPR) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) { xv[k][j] = 0.f; mv[k][j] = 0.f; gv[k][j] = 0.f; }
      if (live && c < C) {
        load_vec<4>(xv[k], x + r * ldx + c);
        load_vec<4>(mv[k], m + r * C + c);
        load_vec<4>(gv[k], g + r * C + c);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sx = fmaf(xv[k][j], xv[k][j], sx);
        sm = fmaf(mv[k][j], mv[k][j], sm);
        mg = fmaf(mv[k][j], gv[k][j], mg);
      }
    }
    const float a = sqrtf(row_sum<LPR>(sx));
    const float nm = sqrtf(row_sum<LPR>(sm));
    const float b = fmaxf(nm, kMsgNormEps);
    const float q = row_sum<LPR>(mg) / b;            // <u, g>
    const float f = a * s / b;
    const float proj = (nm > kMsgNormEps) ? q / b : 0.f;   // u q / b = m q / b^2: zero when the norm was clamped
    const float cx = (a > 0.f) ? s * q / a : 0.f;
    if (live) {
      if (li == 0) ds += a * q;
#pragma unroll
      for (int k = 0; k < Q; ++k) {
        const int c = (li + k * LPR) * 4;
        if (c < C) {
          float om[4], ox[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            om[j] = f * (gv[k][j] - mv[k][j] * proj);
            ox[j] = ADD_X ? fmaf(cx, xv[k][j], gv[k][j]) : cx * xv[k][j];
          }
          if (dm) store_vec<4>(dm + r * C + c, om);
          if (dx) store_vec<4>(dx + r * C + c, ox);
        }
      }
    }
  }
  // ds: lanes -> wave -> workgroup, fixed order
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) ds += __shfl_xor(ds, off);
  if (lane == 0) red[wv] = ds;
  __syncthreads();
  if (threadIdx.x == 0 && ds_partial) ds_partial[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

}  // namespace
}  // namespace dgcn

extern "C" int dgcn_rows_msgnorm_fwd_f32(const float* x, int64_t ldx, const float* m, const float* scale,
                                         int32_t add_x, float* y, int64_t rows, int32_t C, void* stream) {
  if (!x || !m || !scale || !y) return DGCN_E_NULL;
  if (rows < 0) return DGCN_E_SHAPE;
  if (!ln_ok(C, ldx, x, m, y, nullptr, nullptr)) return (C > 0 && C % 4 == 0 && C <= 1024 && ldx >= C) ? DGCN_E_ALIGN : DGCN_E_SHAPE;
  if (rows == 0) return DGCN_OK;
  if (add_x) DGCN_LN_DISPATCH(rows_msgnorm_fwd_kernel, true, x, ldx, m, scale, y, rows, C);
  else DGCN_LN_DISPATCH(rows_msgnorm_fwd_kernel, false, x, ldx, m, scale, y, rows, C);
  return launch_status();
}

extern "C" int dgcn_rows_msgnorm_bwd_f32(const float* g, const float* x, int64_t ldx, const float* m,
                                         const float* scale, int32_t add_x, float* dx, float* dm, float* ds_partial,
                                         int64_t rows, int32_t C, void* stream) {
  if (!g || !x || !m || !scale) return DGCN_E_NULL;
  if (rows <= 0) return DGCN_E_SHAPE;
  if (!ln_ok(C, ldx, x, m, g, dx, dm)) return (C > 0 && C % 4 == 0 && C <= 1024 && ldx >= C) ? DGCN_E_ALIGN : DGCN_E_SHAPE;
  if (add_x) DGCN_LN_DISPATCH(rows_msgnorm_bwd_kernel, true, g, x, ldx, m, scale, dx, dm, ds_partial, rows, C);
  else DGCN_LN_DISPATCH(rows_msgnorm_bwd_kernel, false, g, x, ldx, m, scale, dx, dm, ds_partial, rows, C);
  return launch_status();
}
