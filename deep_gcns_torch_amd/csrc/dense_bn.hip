// BatchNorm-around-the-max kernels of the dense EdgeConv2d path (gfx950).
//
// EdgeConv2d = max_l BN(act(conv([x_i ; x_j - x_i])))  (gcn_lib/dense/torch_vertex.py:31-35 with
// BasicConv = conv -> act -> norm, gcn_lib/dense/torch_nn.py:48-60).  The edge kernel
// (dense_edge.hip) leaves, per (b,n,c), vmax = max_l a, vmin = min_l a and per-workgroup partial sums
// of a and a^2.  Because BatchNorm2d is a per-channel affine map y = scale*a + shift,
//     max_l y = scale * (scale >= 0 ? vmax : vmin) + shift,
// so the whole normalisation is node-sized work.  These kernels do it in a handful of launches
// (the layer is launch/latency bound at B*N = 32768 points):
//   bn_finalize      partial sums -> mean/var -> scale/shift (+ running-stat update), one workgroup,
//                    fixed summation order (deterministic), fp64 accumulation
//   bn_apply         (B,N,C) point-major extremes -> (B,C,N,1) channel-major output, LDS transpose
//   bn_bwd_prep      g (B,C,N) -> gsel = g*scale point-major + partial sums of g and g*sel
//   bn_bwd_finalize  -> dgamma, dbeta and the per-channel coefficients (gsum, gsq) with which the edge
//                    backward reproduces the exact BatchNorm backward (dense over all B*N*k edges)
//   reduce_parts     sum of the dQ partials written by the LDS-accumulating edge backward

#include "dgcn_common.h"

namespace dgcn {
namespace {

constexpr int kTile = 64;
constexpr int kFinThreads = 1024;            // finalize kernels: 16 row groups x 64 channels
constexpr int kFinRows = kFinThreads / kTile;

// ---- bn_finalize -----------------------------------------------------------------------
// bnbuf layout: [0]=scale [1]=shift [2]=mean [3]=invstd, each [C]
struct BnFinalizeParams {
  const float* stats;  // [nparts][2][C] or null (eval mode)
  int nparts;
  int C;
  double count;        // B*N*k
  const float* gamma;  // [C] or null (=1)
  const float* beta;   // [C] or null (=0)
  float* running_mean; // [C] or null
  float* running_var;  // [C] or null
  int64_t* num_batches;  // scalar or null
  int training;        // 1: batch statistics (+ running update), 0: running statistics
  float momentum;
  float eps;
  float* bnbuf;        // [4][C]
};

__global__ __launch_bounds__(kFinThreads) void bn_finalize_kernel(const BnFinalizeParams P) {
  __shared__ double red[kFinRows][2][kTile];
  const int cc = threadIdx.x % kTile;
  const int r = threadIdx.x / kTile;  // interleaved share of the partial rows
  for (int c0 = 0; c0 < P.C; c0 += kTile) {
    const int c = c0 + cc;
    double s1 = 0.0, s2 = 0.0;
    if (P.training && c < P.C) {
#pragma unroll 8
      for (int p = r; p < P.nparts; p += kFinRows) {
        s1 += static_cast<double>(P.stats[(static_cast<int64_t>(p) * 2) * P.C + c]);
        s2 += static_cast<double>(P.stats[(static_cast<int64_t>(p) * 2 + 1) * P.C + c]);
      }
    }
    red[r][0][cc] = s1;
    red[r][1][cc] = s2;
    __syncthreads();
    if (r == 0 && c < P.C) {
      double mean, var;
      if (P.training) {
        double t1 = 0.0, t2 = 0.0;
        for (int q = 0; q < kFinRows; ++q) { t1 += red[q][0][cc]; t2 += red[q][1][cc]; }   // fixed order
        mean = t1 / P.count;
        var = t2 / P.count - mean * mean;  // biased batch variance
        if (var < 0.0) var = 0.0;
        if (P.running_mean) {
          const double m = P.momentum;
          const double unbiased = var * (P.count / (P.count > 1.0 ? P.count - 1.0 : 1.0));
          P.running_mean[c] = static_cast<float>((1.0 - m) * P.running_mean[c] + m * mean);
          P.running_var[c] = static_cast<float>((1.0 - m) * P.running_var[c] + m * unbiased);
        }
      } else {
        mean = P.running_mean[c];
        var = P.running_var[c];
      }
      const double invstd = 1.0 / sqrt(var + static_cast<double>(P.eps));
      const double g = P.gamma ? P.gamma[c] : 1.0;
      const double b = P.beta ? P.beta[c] : 0.0;
      const double scale = g * invstd;
      P.bnbuf[c] = static_cast<float>(scale);
      P.bnbuf[P.C + c] = static_cast<float>(b - mean * scale);
      P.bnbuf[2 * P.C + c] = static_cast<float>(mean);
      P.bnbuf[3 * P.C + c] = static_cast<float>(invstd);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0 && P.training && P.num_batches) *P.num_batches += 1;
}

// ---- bn_apply: out[b,c,n] = scale_c * sel[b,n,c] + shift_c ------------------------------
struct BnApplyParams {
  const float* vmax;   // [B,N,C]
  const float* vmin;   // [B,N,C] or null
  const float* bnbuf;  // [4][C] or null (identity)
  float* out;          // [B,C,N]
  int B, N, C;
  const float* res;     // (B,C,N) with element strides, added as res_scale * res (the block's skip connection), or null
  int64_t rb, rc, rn;
  float res_scale;
};

__global__ __launch_bounds__(kWgThreads) void bn_apply_kernel(const BnApplyParams P) {
  __shared__ float tile[kTile][kTile + 1];
  const int tiles_n = (P.N + kTile - 1) / kTile;
  const int tiles_c = (P.C + kTile - 1) / kTile;
  const int t = blockIdx.x;
  const int b = t / (tiles_n * tiles_c);
  const int tn = (t / tiles_c) % tiles_n;
  const int tc = t % tiles_c;
  const int n0 = tn * kTile, c0 = tc * kTile;
  const int lx = threadIdx.x % kTile;  // fast index
  const int ly = threadIdx.x / kTile;  // 0..3
  // read: rows = points, fast = channels (coalesced along c).  All 16 loads of a thread are issued before the first
  // use (clamped addresses instead of a branch around each load: a branch per element makes every load wait for its
  // own round trip); a lane reads vmax OR vmin, whichever its channel's scale selects
  {
    const int c = c0 + lx;
    const bool cok = c < P.C;
    const int cr = cok ? c : 0;
    float scale = 1.f, shift = 0.f;
    if (P.bnbuf) { scale = P.bnbuf[cr]; shift = P.bnbuf[P.C + cr]; }
    const float* __restrict__ src = (scale >= 0.f || !P.vmin) ? P.vmax : P.vmin;
    float v[kTile / 4];
#pragma unroll
    for (int u = 0; u < kTile / 4; ++u) {
      const int n = n0 + ly + 4 * u;
      v[u] = src[(static_cast<int64_t>(b) * P.N + min(n, P.N - 1)) * P.C + cr];
    }
#pragma unroll
    for (int u = 0; u < kTile / 4; ++u) {
      const int n = n0 + ly + 4 * u;
      tile[ly + 4 * u][lx] = (cok && n < P.N) ? fmaf(scale, v[u], shift) : 0.f;
    }
  }
  __syncthreads();
  // write: rows = channels, fast = points (coalesced along n)
  {
    const int n = n0 + lx;
    const bool nok = n < P.N;
    const int nr = nok ? n : 0;
    float rv[kTile / 4];
    if (P.res) {
#pragma unroll
      for (int u = 0; u < kTile / 4; ++u) {
        const int c = min(c0 + ly + 4 * u, P.C - 1);
        rv[u] = P.res[b * P.rb + c * P.rc + nr * P.rn];
      }
    }
#pragma unroll
    for (int u = 0; u < kTile / 4; ++u) {
      const int c = c0 + ly + 4 * u;
      if (c < P.C && nok) {
        float v = tile[lx][ly + 4 * u];
        if (P.res) v = fmaf(P.res_scale, rv[u], v);
        P.out[(static_cast<int64_t>(b) * P.C + c) * P.N + n] = v;
      }
    }
  }
}

// ---- bn_bwd_prep ------------------------------------------------------------------------
struct BnBwdPrepParams {
  const float* g;       // (B,C,N) with element strides
  int64_t gb, gc, gn;
  const float* vmax;
  const float* vmin;    // or null
  const float* bnbuf;   // or null (identity: scale 1)
  float* gsel;          // [B,N,C] = g * scale (point-major)
  float* partial;       // [gridDim.x][2][C]: sum g, sum g*sel   (or null)
  int B, N, C;
};

// one workgroup = one sample b x 64-point tile, all channel tiles looped inside so the per-workgroup
// partial sums cover whole channel rows
__global__ __launch_bounds__(kWgThreads) void bn_bwd_prep_kernel(const BnBwdPrepParams P) {
  __shared__ float tile[kTile][kTile + 1];
  __shared__ float red[4][2][kTile];
  const int tiles_n = (P.N + kTile - 1) / kTile;
  const int b = blockIdx.x / tiles_n;
  const int n0 = (blockIdx.x % tiles_n) * kTile;
  const int lx = threadIdx.x % kTile;
  const int ly = threadIdx.x / kTile;
  for (int c0 = 0; c0 < P.C; c0 += kTile) {
    // read g channel-major: rows = channels, fast = points (all 16 loads of a thread in flight: see bn_apply_kernel)
    {
      const int n = n0 + lx;
      const bool nok = n < P.N;
      const int nr = nok ? n : 0;
      float gv[kTile / 4];
#pragma unroll
      for (int u = 0; u < kTile / 4; ++u) {
        const int c = min(c0 + ly + 4 * u, P.C - 1);
        gv[u] = P.g[b * P.gb + c * P.gc + nr * P.gn];
      }
#pragma unroll
      for (int u = 0; u < kTile / 4; ++u) tile[ly + 4 * u][lx] = (nok && c0 + ly + 4 * u < P.C) ? gv[u] : 0.f;
    }
    __syncthreads();
    float s1 = 0.f, s2 = 0.f;  // this thread: channel c0+lx, points n0+ly, +4, ...
    const int c = c0 + lx;
    const bool cok = c < P.C;
    const int cr = cok ? c : 0;
    float scale = 1.f;
    if (P.bnbuf) scale = P.bnbuf[cr];
    {
      const float* __restrict__ src = (scale >= 0.f || !P.vmin) ? P.vmax : P.vmin;
      float sv[kTile / 4];
#pragma unroll
      for (int u = 0; u < kTile / 4; ++u) {
        const int n = min(n0 + ly + 4 * u, P.N - 1);
        sv[u] = src[(static_cast<int64_t>(b) * P.N + n) * P.C + cr];
      }
#pragma unroll
      for (int u = 0; u < kTile / 4; ++u) {
        const int i = ly + 4 * u;
        const int n = n0 + i;
        if (cok && n < P.N) {
          const float g1 = tile[lx][i];
          P.gsel[(static_cast<int64_t>(b) * P.N + n) * P.C + c] = g1 * scale;
          s1 += g1;
          s2 = fmaf(g1, sv[u], s2);
        }
      }
    }
    if (P.partial) {
      red[ly][0][lx] = s1;
      red[ly][1][lx] = s2;
      __syncthreads();
      if (ly == 0 && c < P.C) {
        float* out = P.partial + static_cast<int64_t>(blockIdx.x) * 2 * P.C;
        out[c] = ((red[0][0][lx] + red[1][0][lx]) + red[2][0][lx]) + red[3][0][lx];
        out[P.C + c] = ((red[0][1][lx] + red[1][1][lx]) + red[2][1][lx]) + red[3][1][lx];
      }
    }
    __syncthreads();
  }
}

// ---- bn_bwd_finalize --------------------------------------------------------------------
// y = scale*sel + shift, scale = gamma*r, shift = beta - mean*gamma*r, r = (var+eps)^-1/2,
// var = S2/M - mean^2, mean = S1/M.  With ds = sum g*sel, db = sum g:
//   dgamma = r*(ds - mean*db)        dbeta = db
//   dL/dr = gamma*(ds - mean*db)     dL/dvar = -1/2 r^3 dL/dr
//   dL/dmean = -gamma*r*db - 2*mean*dL/dvar
//   gsum = dL/dS1 = dL/dmean / M     gsq = dL/dS2 = dL/dvar / M
// coef layout: [0]=dgamma [1]=dbeta [2]=gsum [3]=gsq, each [C]
struct BnBwdFinalizeParams {
  const float* partial;
  int nparts, C;
  double count;
  const float* gamma;
  const float* bnbuf;
  int training;   // 0: statistics are constants (eval): gsum = gsq = 0
  float* coef;
};

__global__ __launch_bounds__(kFinThreads) void bn_bwd_finalize_kernel(const BnBwdFinalizeParams P) {
  __shared__ double red[kFinRows][2][kTile];
  const int cc = threadIdx.x % kTile;
  const int r = threadIdx.x / kTile;
  for (int c0 = 0; c0 < P.C; c0 += kTile) {
    const int c = c0 + cc;
    double s1 = 0.0, s2 = 0.0;
    if (c < P.C) {
#pragma unroll 8
      for (int p = r; p < P.nparts; p += kFinRows) {
        s1 += static_cast<double>(P.partial[(static_cast<int64_t>(p) * 2) * P.C + c]);
        s2 += static_cast<double>(P.partial[(static_cast<int64_t>(p) * 2 + 1) * P.C + c]);
      }
    }
    red[r][0][cc] = s1;
    red[r][1][cc] = s2;
    __syncthreads();
    if (r == 0 && c < P.C) {
      double db = 0.0, ds = 0.0;
      for (int q = 0; q < kFinRows; ++q) { db += red[q][0][cc]; ds += red[q][1][cc]; }   // fixed order
      const double gamma = P.gamma ? P.gamma[c] : 1.0;
      const double mean = P.bnbuf[2 * P.C + c];
      const double rstd = P.bnbuf[3 * P.C + c];
      const double centred = ds - mean * db;
      P.coef[c] = static_cast<float>(rstd * centred);
      P.coef[P.C + c] = static_cast<float>(db);
      double gsum = 0.0, gsq = 0.0;
      if (P.training) {
        const double dvar = -0.5 * rstd * rstd * rstd * gamma * centred;
        const double dmean = -gamma * rstd * db - 2.0 * mean * dvar;
        gsum = dmean / P.count;
        gsq = dvar / P.count;
      }
      P.coef[2 * P.C + c] = static_cast<float>(gsum);
      P.coef[3 * P.C + c] = static_cast<float>(gsq);
    }
    __syncthreads();
  }
}

// ---- reduce_parts: dst[row*ld + c] = sum_s parts[s][row][c] -------------------------------
__global__ __launch_bounds__(kWgThreads) void reduce_parts_kernel(const float* __restrict__ parts, int nsplit,
                                                                  int64_t rows, int C, float* __restrict__ dst,
                                                                  int64_t ld) {
  const int c4 = C / 4;
  const int64_t total = rows * c4;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t row = i / c4;
    const int cv = static_cast<int>(i % c4);
    float4 acc = reinterpret_cast<const float4*>(parts)[i];
    for (int s = 1; s < nsplit; ++s) {
      const float4 v = reinterpret_cast<const float4*>(parts + static_cast<int64_t>(s) * rows * C)[i];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4*>(dst + row * ld + cv * 4) = acc;
  }
}

}  // namespace
}  // namespace dgcn

using namespace dgcn;

extern "C" int dgcn_bn_finalize_f32(const float* stats, int32_t nparts, int32_t C, double count,
                                    const float* gamma, const float* beta, float* running_mean,
                                    float* running_var, int64_t* num_batches, int32_t training,
                                    float momentum, float eps, float* bnbuf, void* stream) {
  if (!bnbuf) return DGCN_E_NULL;
  if (C <= 0 || nparts < 0 || count <= 0.0) return DGCN_E_SHAPE;
  if (training && !stats) return DGCN_E_NULL;
  if (!training && (!running_mean || !running_var)) return DGCN_E_NULL;
  if (running_mean && !running_var) return DGCN_E_NULL;
  BnFinalizeParams P{stats, nparts, C, count, gamma, beta, running_mean, running_var, num_batches,
                     training, momentum, eps, bnbuf};
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(kFinThreads), 0, static_cast<hipStream_t>(stream), P);
  return launch_status();
}

extern "C" int dgcn_bn_apply_f32(const float* vmax, const float* vmin, const float* bnbuf, float* out,
                                 int32_t B, int32_t N, int32_t C, void* stream) {
  return dgcn_bn_apply_res_f32(vmax, vmin, bnbuf, nullptr, 0, 0, 0, 0.f, out, B, N, C, stream);
}

extern "C" int dgcn_bn_apply_res_f32(const float* vmax, const float* vmin, const float* bnbuf, const float* res,
                                     int64_t rb, int64_t rc, int64_t rn, float res_scale, float* out, int32_t B,
                                     int32_t N, int32_t C, void* stream) {
  if (!vmax || !out) return DGCN_E_NULL;
  if (B < 0 || N <= 0 || C <= 0) return DGCN_E_SHAPE;
  if (B == 0) return DGCN_OK;
  BnApplyParams P{vmax, vmin, bnbuf, out, B, N, C, res, rb, rc, rn, res_scale};
  const int64_t tiles = static_cast<int64_t>(B) * ((N + kTile - 1) / kTile) * ((C + kTile - 1) / kTile);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(static_cast<unsigned>(tiles)), dim3(kWgThreads), 0,
                     static_cast<hipStream_t>(stream), P);
  return launch_status();
}

extern "C" int32_t dgcn_bn_bwd_num_partials(int32_t B, int32_t N) {
  if (B <= 0 || N <= 0) return 0;
  return B * ((N + kTile - 1) / kTile);
}

extern "C" int dgcn_bn_bwd_prep_f32(const float* g, int64_t gb, int64_t gc, int64_t gn, const float* vmax,
                                    const float* vmin, const float* bnbuf, float* gsel, float* partial,
                                    int32_t B, int32_t N, int32_t C, void* stream) {
  if (!g || !vmax || !gsel) return DGCN_E_NULL;
  if (B < 0 || N <= 0 || C <= 0) return DGCN_E_SHAPE;
  if (B == 0) return DGCN_OK;
  BnBwdPrepParams P{g, gb, gc, gn, vmax, vmin, bnbuf, gsel, partial, B, N, C};
  hipLaunchKernelGGL(bn_bwd_prep_kernel, dim3(dgcn_bn_bwd_num_partials(B, N)), dim3(kWgThreads), 0,
                     static_cast<hipStream_t>(stream), P);
  return launch_status();
}

extern "C" int dgcn_bn_bwd_finalize_f32(const float* partial, int32_t nparts, int32_t C, double count,
                                        const float* gamma, const float* bnbuf, int32_t training,
                                        float* coef, void* stream) {
  if (!partial || !bnbuf || !coef) return DGCN_E_NULL;
  if (C <= 0 || nparts <= 0 || count <= 0.0) return DGCN_E_SHAPE;
  BnBwdFinalizeParams P{partial, nparts, C, count, gamma, bnbuf, training, coef};
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(1), dim3(kFinThreads), 0, static_cast<hipStream_t>(stream), P);
  return launch_status();
}

extern "C" int dgcn_reduce_parts_f32(const float* parts, int32_t nsplit, int64_t rows, int32_t C, float* dst,
                                     int64_t ld, void* stream) {
  if (!parts || !dst) return DGCN_E_NULL;
  if (nsplit < 1 || rows < 0 || C <= 0 || C % 4 != 0 || ld < C || ld % 4 != 0) return DGCN_E_SHAPE;
  if ((reinterpret_cast<uintptr_t>(parts) & 15u) || (reinterpret_cast<uintptr_t>(dst) & 15u)) return DGCN_E_ALIGN;
  if (rows == 0) return DGCN_OK;
  int64_t blocks = (rows * (C / 4) + kWgThreads - 1) / kWgThreads;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(reduce_parts_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kWgThreads), 0,
                     static_cast<hipStream_t>(stream), parts, nsplit, rows, C, dst, ld);
  return launch_status();
}
