// Dense (B x C x N x 1) graph convolution kernels for gfx950.
//
//   dgcn_vertex_gemm_f32           per-VERTEX split of the EdgeConv edge MLP on fp32 MFMA
//   dgcn_dense_edge_reduce_fwd_f32 gather + activation + neighbourhood max/min + BN statistics
//   dgcn_dense_edge_reduce_bwd_f32 its backward
//
// Replaces batched_index_select (gcn_lib/dense/torch_nn.py:75-96), the 1x1 Conv2d + act of BasicConv
// as used INSIDE EdgeConv2d/MRConv2d (torch_nn.py:48-60) and torch.max(..., -1)
// (gcn_lib/dense/torch_vertex.py:16-20, 31-35).  Two identities make the layer cheap:
//   * the edge MLP is linear: W [x_i ; x_j - x_i] + b = ((W1 - W2) x_i + b) + W2 x_j = P_i + Q_j,
//     so the GEMM runs once per vertex (16x fewer flops at k = 16), on v_mfma_f32_16x16x4_f32
//     (exact f32 fma chain) and emits point-major rows that the gather can fetch as whole lines;
//   * BatchNorm is a per-channel affine map, so max_l BN(a_l) needs only max_l a_l and min_l a_l plus
//     the batch statistics sum a, sum a^2: the (B,C',N,k) activation tensor is never materialised.

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "dgcn_common.h"

namespace dgcn {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------
// vertex GEMM:  out[(b*N+n)*M + m] = sum_c x[b,c,n] * W[c*M+m] + bias[m]
// One wave: 16 points x (16*TJ) outputs, K-loop over channels in steps of 4 (one MFMA per step
// per column tile).  A[i][k] = x[b, k0+k, n0+i]  (lanes l&15 walk consecutive points: coalesced),
// B[k][j] = W[k0+k][j0+j], D[row=(l>>4)*4+reg][col=l&15].
// ---------------------------------------------------------------------------------------
struct GemmParams {
  const float* x;
  int64_t sb, sc, sn;
  int B, C, N, M;
  const float* W;
  const float* bias;
  float* out;
  int conv_split;  // 0: W is [C][M].  1: W is the Conv2d weight [M/2][2C] of EdgeConv2d and the kernel forms
                   //    [(W1-W2)^T | W2^T] on the fly; bias [M/2] applies to the first half only
};

constexpr int kGemmKC = 64;    // contraction indices (channels) per LDS tile
constexpr int kGemmMC = 128;   // output columns per LDS tile: eight 16-column MFMA tiles per wave pass
constexpr int kGemmPad = 16;   // row padding (floats): the four lane groups of an MFMA operand read land 16 banks apart

// bits of v where ok, zero bits otherwise: a select that the compiler cannot turn into a branch around the load that
// produced v (a branch per element makes every load of a tile fill wait for its own L2 round trip)
__device__ __forceinline__ float gemm_keep(float v, bool ok) {
  return __uint_as_float(__float_as_uint(v) & (ok ? 0xFFFFFFFFu : 0u));
}

// Persistent workgroups: with C <= 64 one LDS tile holds the whole contraction for 128 output columns, a workgroup
// forms it ONCE (coalesced 16-byte loads along the contiguous axis of the weight, all of a thread's loads in flight
// together) and walks its share of the 16-point tiles under it; the A operands of a tile (16 values per lane) are loaded
// before the tile is formed, so the two latencies overlap.  The result tile leaves through a per-wave LDS buffer as
// whole 512-byte rows of the point-major output.  (Rounds 1 - 4: one workgroup per four point tiles, each re-forming
// the weight tile element by element -- 64 different cache lines per load instruction for the Conv2d layout and a
// dependent branch per element: 28 us per layer at shape D, most of it the tile fill.)
__global__ __launch_bounds__(kWgThreads, 2) void vertex_gemm_kernel(const GemmParams P) {
  __shared__ __attribute__((aligned(16))) float wt[kGemmKC][kGemmMC + kGemmPad];       // 36 KB
  __shared__ __attribute__((aligned(16))) float ot[kWavesPerWg][16][kGemmMC + 4];      // 33 KB: per-wave result rows
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int tiles_n = (P.N + 15) / 16;
  const int total_tiles = P.B * tiles_n;
  const int groups = (total_tiles + kWavesPerWg - 1) / kWavesPerWg;
  const int li = lane & 15, lk = lane >> 4;
  const bool single = P.C <= kGemmKC;
  const bool wvec = P.conv_split && P.C % 4 == 0 && (reinterpret_cast<uintptr_t>(P.W) & 15u) == 0;
  const bool ovec = P.M % 4 == 0 && (reinterpret_cast<uintptr_t>(P.out) & 15u) == 0;
  const int half = P.M / 2;

  for (int m0 = 0; m0 < P.M; m0 += kGemmMC) {
    bool filled = false;
    for (int grp = blockIdx.x; grp < groups; grp += gridDim.x) {
      const int tile = min(grp * kWavesPerWg + wave, total_tiles - 1);    // clamp: all waves hit the barriers
      const bool tile_ok = grp * kWavesPerWg + wave < total_tiles;
      const int b = tile / tiles_n;
      const int n0 = (tile % tiles_n) * 16;
      const bool n_ok = n0 + li < P.N;
      const float* xa = P.x + static_cast<int64_t>(b) * P.sb + static_cast<int64_t>(min(n0 + li, P.N - 1)) * P.sn;
      f32x4 acc[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int kc = 0; kc < P.C; kc += kGemmKC) {
        // A operands of the chunk: lane (point li, kq = lk) holds channels kc + 4 q + kq
        float av[kGemmKC / 4];
#pragma unroll
        for (int q = 0; q < kGemmKC / 4; ++q) {
          const int k = kc + 4 * q + lk;
          av[q] = gemm_keep(xa[static_cast<int64_t>(min(k, P.C - 1)) * P.sc], k < P.C && n_ok);
        }
        if (!(single && filled)) {
          __syncthreads();
          if (wvec) {
            // Conv2d weight [M/2][2C]: column j of the effective operand is row r = j mod M/2, W1[r] - W2[r] (j < M/2)
            // or W2[r]; a wave reads four rows, 64 channels = 256 contiguous bytes each, per pass
            const int k4 = (lane & 15) * 4;
            float4 wa[8], wb[8];
#pragma unroll
            for (int ps = 0; ps < 8; ++ps) {
              const int jj = m0 + ps * 16 + (threadIdx.x >> 4);
              const bool ok = jj < P.M && kc + k4 < P.C;
              const int r = ok ? (jj < half ? jj : jj - half) : 0;
              const float* wr = P.W + static_cast<int64_t>(r) * (2 * P.C) + (ok ? kc + k4 : 0);
              wa[ps] = *reinterpret_cast<const float4*>(wr);
              wb[ps] = *reinterpret_cast<const float4*>(wr + P.C);
            }
#pragma unroll
            for (int ps = 0; ps < 8; ++ps) {
              const int jl = ps * 16 + (threadIdx.x >> 4);
              const int jj = m0 + jl;
              const bool ok = jj < P.M && kc + k4 < P.C;
              const bool top = jj < half;
              wt[k4][jl] = ok ? (top ? wa[ps].x - wb[ps].x : wb[ps].x) : 0.f;
              wt[k4 + 1][jl] = ok ? (top ? wa[ps].y - wb[ps].y : wb[ps].y) : 0.f;
              wt[k4 + 2][jl] = ok ? (top ? wa[ps].z - wb[ps].z : wb[ps].z) : 0.f;
              wt[k4 + 3][jl] = ok ? (top ? wa[ps].w - wb[ps].w : wb[ps].w) : 0.f;
            }
          } else {
            constexpr int kPer = 8;
            for (int e0 = threadIdx.x; e0 < kGemmKC * kGemmMC; e0 += kWgThreads * kPer) {
              float v[kPer];
#pragma unroll
              for (int u = 0; u < kPer; ++u) {
                const int e = e0 + u * kWgThreads;
                const int kk = e / kGemmMC, jl = e % kGemmMC;
                const int k = kc + kk, jj = m0 + jl;
                const bool ok = k < P.C && jj < P.M;
                const int kr = ok ? k : 0, jr = ok ? jj : 0;
                if (P.conv_split) {                              // (uniform over the launch)
                  const bool top = jr < half;
                  const float* wr = P.W + static_cast<int64_t>(top ? jr : jr - half) * (2 * P.C);
                  const float w1 = wr[kr], w2 = wr[P.C + kr];
                  v[u] = gemm_keep(top ? w1 - w2 : w2, ok);
                } else {
                  v[u] = gemm_keep(P.W[static_cast<int64_t>(kr) * P.M + jr], ok);
                }
              }
#pragma unroll
              for (int u = 0; u < kPer; ++u) {
                const int e = e0 + u * kWgThreads;
                wt[e / kGemmMC][e % kGemmMC] = v[u];
              }
            }
          }
          __syncthreads();
          filled = true;
        }
#pragma unroll
        for (int q = 0; q < kGemmKC / 4; ++q) {
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            const float bv = wt[4 * q + lk][t * 16 + li];        // rows past C and columns past M hold zeros
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bv, acc[t], 0, 0, 0);
          }
        }
      }
      // acc[t][r] = out row n0 + 4 lk + r, column m0 + 16 t + li
      if (ovec) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const int j = m0 + t * 16 + li;
          const float bj = (P.bias && j < P.M && (!P.conv_split || j < half)) ? P.bias[j] : 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) ot[wave][lk * 4 + r][t * 16 + li] = acc[t][r] + bj;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
          const int idx = ps * kWave + lane;
          const int row = idx >> 5, c4 = (idx & 31) * 4;
          if (tile_ok && n0 + row < P.N && m0 + c4 < P.M) {
            *reinterpret_cast<float4*>(P.out + (static_cast<int64_t>(b) * P.N + n0 + row) * P.M + m0 + c4) =
                *reinterpret_cast<const float4*>(&ot[wave][row][c4]);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      } else if (tile_ok) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const int j = m0 + t * 16 + li;
          if (j < P.M) {
            const float bj = (P.bias && (!P.conv_split || j < half)) ? P.bias[j] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = n0 + lk * 4 + r;
              if (row < P.N) P.out[(static_cast<int64_t>(b) * P.N + row) * P.M + j] = acc[t][r] + bj;
            }
          }
        }
      }
    }
  }
}

int gemm_grid(int64_t tiles) {
  const int64_t groups = (tiles + kWavesPerWg - 1) / kWavesPerWg;
  return static_cast<int>(groups < 2 * num_cus() ? groups : 2 * num_cus());
}

// ---------------------------------------------------------------------------------------
// edge reduce
// ---------------------------------------------------------------------------------------
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_LEAKY = 2 };

struct EdgeParams {
  const float* P;   // [B,N,C] or null
  const float* Q;   // [B,N,C]
  const int64_t* idx;
  int64_t ib, in_, ik;  // element strides of idx (B,N,k)
  int64_t ldp, ldq;     // row strides (floats) of P/dP and Q/dQ rows; outputs vmax.. are dense [B,N,C]
  int B, N, C, k;
  int act;
  float slope;
  // forward outputs
  float* vmax;      // [B,N,C]
  float* vmin;      // [B,N,C] or null
  uint8_t* amax;    // [B,N,C] or null: neighbour slot l of the maximum (first on ties)
  uint8_t* amin;
  float* stats;     // [gridDim.x][2][C] partial sums of a and a^2, or null
  // backward inputs / outputs
  const float* gmax;  // [B,N,C]
  const float* gmin;  // [B,N,C] or null
  const float* gsum;  // [C] or null
  const float* gsq;   // [C] or null
  const float* selscale;  // [C] or null: BatchNorm scale; gmax is routed to the arg-max slot where scale >= 0
                          // and to the arg-min slot where scale < 0 (gmin is then ignored)
  float* dP;          // [B,N,C] or null
  float* dQ;          // [B,N,C], pre-zeroed, accumulated with hardware fp32 atomics
  float* dz;          // [B,N,k,C] or null: backward writes every edge's dz row here INSTEAD of accumulating dQ (the
                      // inverse-list gather kernel sums them per neighbour afterwards)
  int32_t* inv_cnt;   // with dz: [B,N] zeroed in-degree counters; the edge's arrival rank in its neighbour's list
  int32_t* inv_rank;  //          = atomicAdd(inv_cnt, 1) is stored per edge [B,N,k] (counting sort, first pass)
};

__device__ __forceinline__ float act_apply(float z, int act, float slope) {
  if (act == ACT_RELU) return fmaxf(z, 0.f);
  if (act == ACT_LEAKY) return z > 0.f ? z : z * slope;
  return z;
}
__device__ __forceinline__ float act_grad(float z, int act, float slope) {
  if (act == ACT_RELU) return z > 0.f ? 1.f : 0.f;
  if (act == ACT_LEAKY) return z > 0.f ? 1.f : slope;
  return 1.f;
}

// LPR lanes x float4 cover the C channels of one point; a wave owns G = 64/LPR points at once.
template <int LPR, bool BWD>
__global__ __launch_bounds__(kWgThreads) void dense_edge_kernel(const EdgeParams E) {
  constexpr int G = kWave / LPR;
  constexpr int U = 4;
  __shared__ float red[kWavesPerWg][2][LPR * 4];

  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int g = lane / LPR;
  const int cl = lane % LPR;
  const int C = E.C, k = E.k;
  const int64_t total_pts = static_cast<int64_t>(E.B) * E.N;
  const int64_t groups = (total_pts + G - 1) / G;
  const int64_t wave_stride = static_cast<int64_t>(gridDim.x) * kWavesPerWg;

  for (int cb = 0; cb < C; cb += LPR * 4) {
    const int c0 = cb + cl * 4;
    const bool act_lane = c0 < C;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    float gs[4] = {0.f, 0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (BWD) {
      if (act_lane && E.gsum) load_vec<4>(gs, E.gsum + c0);
      if (act_lane && E.gsq) load_vec<4>(gq, E.gsq + c0);
    }

    for (int64_t grp = static_cast<int64_t>(blockIdx.x) * kWavesPerWg + wave; grp < groups; grp += wave_stride) {
      const int64_t pt = grp * G + g;
      const bool pt_ok = pt < total_pts && act_lane;
      const int b = static_cast<int>(min(pt, total_pts - 1) / E.N);
      const int n = static_cast<int>(min(pt, total_pts - 1) % E.N);
      const int64_t row = (static_cast<int64_t>(b) * E.N + n) * C + c0;
      const int64_t prow = (static_cast<int64_t>(b) * E.N + n) * E.ldp + c0;
      float p[4] = {0.f, 0.f, 0.f, 0.f};
      if (pt_ok && E.P) load_vec<4>(p, E.P + prow);
      const int64_t* irow = E.idx + b * E.ib + n * E.in_;
      const float* Qb = E.Q + static_cast<int64_t>(b) * E.N * E.ldq + c0;

      float vmx[4], vmn[4];
      int amx[4], amn[4];
      float gmx[4] = {0.f, 0.f, 0.f, 0.f}, gmn[4] = {0.f, 0.f, 0.f, 0.f}, dp[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) { vmx[j] = DGCN_NEG_INF; vmn[j] = -DGCN_NEG_INF; amx[j] = 0; amn[j] = 0; }
      if constexpr (BWD) {
        if (pt_ok) {
          load_vec<4>(gmx, E.gmax + row);
          if (E.gmin) load_vec<4>(gmn, E.gmin + row);
          uint32_t pk = *reinterpret_cast<const uint32_t*>(E.amax + row);
#pragma unroll
          for (int j = 0; j < 4; ++j) amx[j] = (pk >> (8 * j)) & 0xFF;
          if (E.amin) {
            pk = *reinterpret_cast<const uint32_t*>(E.amin + row);
#pragma unroll
            for (int j = 0; j < 4; ++j) amn[j] = (pk >> (8 * j)) & 0xFF;
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) amn[j] = -1;
          }
          if (E.selscale) {
            float sc[4];
            load_vec<4>(sc, E.selscale + c0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              if (sc[j] < 0.f) amx[j] = amn[j];
              amn[j] = -1;
              gmn[j] = 0.f;
            }
          }
        }
      }

      for (int l0 = 0; l0 < k; l0 += U) {
        float q[U][4];
        int nb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int l = l0 + u;
          nb[u] = (pt_ok && l < k) ? static_cast<int>(irow[l * E.ik]) : -1;
#pragma unroll
          for (int j = 0; j < 4; ++j) q[u][j] = 0.f;
          if (nb[u] >= 0) load_vec<4>(q[u], Qb + static_cast<int64_t>(nb[u]) * E.ldq);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (nb[u] < 0) continue;
          const int l = l0 + u;
          float dz[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float z = p[j] + q[u][j];
            const float a = act_apply(z, E.act, E.slope);
            if constexpr (!BWD) {
              if (a > vmx[j]) { vmx[j] = a; amx[j] = l; }
              if (a < vmn[j]) { vmn[j] = a; amn[j] = l; }
              s1[j] += a;
              s2[j] = fmaf(a, a, s2[j]);
            } else {
              float da = gs[j] + 2.f * a * gq[j];
              if (l == amx[j]) da += gmx[j];
              if (l == amn[j]) da += gmn[j];
              dz[j] = da * act_grad(z, E.act, E.slope);
              dp[j] += dz[j];
            }
          }
          if constexpr (BWD) {
            if (E.dz) {
              store_vec<4>(E.dz + ((static_cast<int64_t>(b) * E.N + n) * k + l) * C + c0, dz);
              if (cl == 0 && cb == 0) {
                E.inv_rank[(static_cast<int64_t>(b) * E.N + n) * k + l] =
                    atomicAdd(&E.inv_cnt[static_cast<int64_t>(b) * E.N + nb[u]], 1);
              }
            } else {
              float* dq = E.dQ + (static_cast<int64_t>(b) * E.N + nb[u]) * E.ldq + c0;
#pragma unroll
              for (int j = 0; j < 4; ++j) unsafeAtomicAdd(dq + j, dz[j]);
            }
          }
        }
      }

      if (pt_ok) {
        if constexpr (!BWD) {
          store_vec<4>(E.vmax + row, vmx);
          if (E.vmin) store_vec<4>(E.vmin + row, vmn);
          if (E.amax) {
            *reinterpret_cast<uint32_t*>(E.amax + row) =
                (amx[0] & 0xFF) | ((amx[1] & 0xFF) << 8) | ((amx[2] & 0xFF) << 16) | ((amx[3] & 0xFF) << 24);
          }
          if (E.amin) {
            *reinterpret_cast<uint32_t*>(E.amin + row) =
                (amn[0] & 0xFF) | ((amn[1] & 0xFF) << 8) | ((amn[2] & 0xFF) << 16) | ((amn[3] & 0xFF) << 24);
          }
        } else {
          if (E.dP) store_vec<4>(E.dP + prow, dp);
        }
      }
    }

    if constexpr (!BWD) {
      if (E.stats) {
        // combine the G point groups of the wave, then the waves of the workgroup, in a fixed order
#pragma unroll
        for (int off = LPR; off < kWave; off <<= 1) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            s1[j] += __shfl_xor(s1[j], off);
            s2[j] += __shfl_xor(s2[j], off);
          }
        }
        if (g == 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            red[wave][0][cl * 4 + j] = s1[j];
            red[wave][1][cl * 4 + j] = s2[j];
          }
        }
        __syncthreads();
        if (wave == 0 && g == 0 && act_lane) {
          float t1[4], t2[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            t1[j] = ((red[0][0][cl * 4 + j] + red[1][0][cl * 4 + j]) + red[2][0][cl * 4 + j]) + red[3][0][cl * 4 + j];
            t2[j] = ((red[0][1][cl * 4 + j] + red[1][1][cl * 4 + j]) + red[2][1][cl * 4 + j]) + red[3][1][cl * 4 + j];
          }
          float* st = E.stats + static_cast<int64_t>(blockIdx.x) * 2 * C;
          store_vec<4>(st + c0, t1);
          store_vec<4>(st + C + c0, t2);
        }
        __syncthreads();
      }
    }
  }
}


// ---------------------------------------------------------------------------------------
// Backward without global atomics.  dQ[b,j,:] receives contributions from every point that has j as
// a neighbour; global fp32 atomics run at ~60 G adds/s (0.55 ms for one ResGCN layer).  Instead a
// workgroup owns (sample b, 8-channel slice, 1/nsplit of the points) and accumulates its share of
// dQ[b, :, slice] in LDS (N x 8 floats = 128 KB at N = 4096) with ds_add_f32, then writes the slice
// as a dense partial; the partials are summed in a fixed order by the caller: deterministic.
// Two lanes per point (float4 each); dP is summed in registers as before.
// ---------------------------------------------------------------------------------------
constexpr int kBwdLdsThreads = 1024;
constexpr int kBwdSlice = 8;

__global__ __launch_bounds__(kBwdLdsThreads) void dense_edge_bwd_lds_kernel(const EdgeParams E, float* __restrict__ parts,
                                                                            int nsplit) {
  extern __shared__ __attribute__((aligned(16))) float acc[];  // [N][8]
  const int b = blockIdx.x, cs = blockIdx.y, sp = blockIdx.z;
  const int tid = threadIdx.x;
  const int half = tid & 1;
  const int C = E.C, N = E.N, k = E.k;
  const int c0 = cs * kBwdSlice + half * 4;
  const bool cok = c0 < C;
  #ifndef DGCN_EDGE_BWD_U
#define DGCN_EDGE_BWD_U 4
#endif
  constexpr int U = DGCN_EDGE_BWD_U;   // neighbour rows in flight per thread (8 measured the same: the kernel is LDS-atomic bound)

  for (int i = tid; i < N * 2; i += kBwdLdsThreads) reinterpret_cast<float4*>(acc)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();

  float gs[4] = {0.f, 0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};
  if (cok && E.gsum) load_vec<4>(gs, E.gsum + c0);
  if (cok && E.gsq) load_vec<4>(gq, E.gsq + c0);
  const int chunk = (N + nsplit - 1) / nsplit;
  const int n_beg = sp * chunk;
  const int n_end = min(N, n_beg + chunk);
  const float* Qb = E.Q + static_cast<int64_t>(b) * N * E.ldq + c0;

  if (cok) {
    for (int n = n_beg + (tid >> 1); n < n_end; n += kBwdLdsThreads / 2) {
      const int64_t row = (static_cast<int64_t>(b) * N + n) * C + c0;
      const int64_t prow = (static_cast<int64_t>(b) * N + n) * E.ldp + c0;
      float p[4] = {0.f, 0.f, 0.f, 0.f}, gmx[4], gmn[4] = {0.f, 0.f, 0.f, 0.f}, dp[4] = {0.f, 0.f, 0.f, 0.f};
      int amx[4], amn[4];
      if (E.P) load_vec<4>(p, E.P + prow);
      load_vec<4>(gmx, E.gmax + row);
      uint32_t pk = *reinterpret_cast<const uint32_t*>(E.amax + row);
#pragma unroll
      for (int j = 0; j < 4; ++j) { amx[j] = (pk >> (8 * j)) & 0xFF; amn[j] = -1; }
      if (E.gmin) {
        load_vec<4>(gmn, E.gmin + row);
        pk = *reinterpret_cast<const uint32_t*>(E.amin + row);
#pragma unroll
        for (int j = 0; j < 4; ++j) amn[j] = (pk >> (8 * j)) & 0xFF;
      }
      if (E.selscale) {
        float sc[4];
        load_vec<4>(sc, E.selscale + c0);
        if (E.amin) {
          pk = *reinterpret_cast<const uint32_t*>(E.amin + row);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (sc[j] < 0.f) amx[j] = (pk >> (8 * j)) & 0xFF;
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { amn[j] = -1; gmn[j] = 0.f; }
      }
      const int64_t* irow = E.idx + b * E.ib + n * E.in_;
      for (int l0 = 0; l0 < k; l0 += U) {
        float q[U][4];
        int nb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int l = l0 + u;
          nb[u] = (l < k) ? static_cast<int>(irow[l * E.ik]) : -1;
#pragma unroll
          for (int j = 0; j < 4; ++j) q[u][j] = 0.f;
          if (nb[u] >= 0) load_vec<4>(q[u], Qb + static_cast<int64_t>(nb[u]) * E.ldq);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (nb[u] < 0) continue;
          const int l = l0 + u;
          float* a_row = acc + nb[u] * kBwdSlice + half * 4;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float z = p[j] + q[u][j];
            const float a = act_apply(z, E.act, E.slope);
            float da = gs[j] + 2.f * a * gq[j];
            if (l == amx[j]) da += gmx[j];
            if (l == amn[j]) da += gmn[j];
            const float dz = da * act_grad(z, E.act, E.slope);
            dp[j] += dz;
            unsafeAtomicAdd(a_row + j, dz);   // ds_add_f32
          }
        }
      }
      if (E.dP) store_vec<4>(E.dP + prow, dp);
    }
  }
  __syncthreads();
  for (int i = tid; i < N * 2; i += kBwdLdsThreads) {
    const int j = i >> 1, h = i & 1;
    const int cc = cs * kBwdSlice + h * 4;
    if (cc < C) {
      const float4 v = reinterpret_cast<const float4*>(acc)[i];
      *reinterpret_cast<float4*>(parts + ((static_cast<int64_t>(sp) * E.B + b) * N + j) * C + cc) = v;
    }
  }
}


// ---------------------------------------------------------------------------------------
// Backward through inverse neighbour lists (default when the caller provides the workspace).
//   dQ[b,j,:] = sum over the edges (n,l) with idx[b,n,l] = j of dz[b,n,l,:]
// Instead of 33.5 M LDS float atomics per ResGCN layer (the LDS-privatised kernel above runs at ~1 lane/clk/CU),
// the edge kernel writes the dz rows once (B*N*k*C floats, L2/MALL resident) and takes each edge's rank in its
// neighbour's list with one integer L2 atomic; a scan and an atomic-free fill finish the counting sort, and a
// gather kernel sums each neighbour's rows:
// one 16-byte-per-lane row read per incoming edge, no float atomics.  The order inside a list follows the fill
// order, so sums are reproducible only up to fp32 rounding (as with the LDS atomics before).
// ---------------------------------------------------------------------------------------
// ptr[b][0..N] = exclusive scan of cnt[b][0..N) ; one workgroup per sample
__global__ __launch_bounds__(kBwdLdsThreads) void inv_scan_kernel(const int32_t* __restrict__ cnt, int N,
                                                                 int32_t* __restrict__ ptr) {
  __shared__ int32_t wsum[kBwdLdsThreads / kWave];
  __shared__ int32_t carry;
  const int b = blockIdx.x;
  const int lane = lane_id(), wv = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < N; base += kBwdLdsThreads) {
    const int j = base + threadIdx.x;
    const int v = j < N ? cnt[static_cast<int64_t>(b) * N + j] : 0;
    int incl = v;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const int t = __shfl_up(incl, off);
      if (lane >= off) incl += t;
    }
    if (lane == kWave - 1) wsum[wv] = incl;
    __syncthreads();
    int before = carry;
    for (int w = 0; w < wv; ++w) before += wsum[w];
    const int excl = before + incl - v;
    if (j < N) ptr[static_cast<int64_t>(b) * (N + 1) + j] = excl;
    __syncthreads();
    if (threadIdx.x == kBwdLdsThreads - 1) carry = before + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) ptr[static_cast<int64_t>(b) * (N + 1) + N] = carry;
}

__global__ __launch_bounds__(kWgThreads) void inv_fill_kernel(const int64_t* __restrict__ idx, int64_t ib, int64_t in_,
                                                             int64_t ik, int B, int N, int k,
                                                             const int32_t* __restrict__ ptr,
                                                             const int32_t* __restrict__ rank,
                                                             int32_t* __restrict__ inv) {
  const int64_t per_b = static_cast<int64_t>(N) * k;
  const int64_t total = static_cast<int64_t>(B) * per_b;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < total;
       e += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int l = static_cast<int>(e % k);
    const int n = static_cast<int>((e / k) % N);
    const int b = static_cast<int>(e / per_b);
    const int j = static_cast<int>(idx[b * ib + n * in_ + l * ik]);
    inv[b * per_b + ptr[static_cast<int64_t>(b) * (N + 1) + j] + rank[e]] = n * k + l;   // edge id inside the sample
  }
}

// One WAVE per neighbour row: LPR lanes x float4 cover the C channels, the G = 64/LPR lane groups walk G edges of
// the SAME list side by side (kNN in-degrees are heavy-tailed: a hub with hundreds of in-edges must not be summed by
// one lane group), U loads in flight each, partial sums combined with shuffles.
template <int LPR>
__global__ __launch_bounds__(kWgThreads) void inv_gather_kernel(const float* __restrict__ dz,
                                                               const int32_t* __restrict__ ptr,
                                                               const int32_t* __restrict__ inv, int B, int N, int C,
                                                               int k, float* __restrict__ dQ, int64_t ldq) {
  constexpr int G = kWave / LPR;
  constexpr int U = 4;
  const int lane = lane_id();
  const int g = lane / LPR, cl = lane % LPR;
  const int64_t total = static_cast<int64_t>(B) * N;
  const int64_t per_b = static_cast<int64_t>(N) * k;
  const int64_t wave_stride = static_cast<int64_t>(gridDim.x) * kWavesPerWg;
  for (int cb = 0; cb < C; cb += LPR * 4) {
    const int c0 = cb + cl * 4;
    const bool act = c0 < C;
    for (int64_t t = static_cast<int64_t>(blockIdx.x) * kWavesPerWg + (threadIdx.x >> 6); t < total; t += wave_stride) {
      const int b = static_cast<int>(t / N);
      const int j = static_cast<int>(t % N);
      const int beg = uni(ptr[static_cast<int64_t>(b) * (N + 1) + j]);
      const int end = uni(ptr[static_cast<int64_t>(b) * (N + 1) + j + 1]);
      const float* dzb = dz + b * per_b * C + c0;
      const int32_t* invb = inv + b * per_b;
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      for (int blk = beg; blk < end; blk += kWave) {
        const int nb = min(kWave, end - blk);
        const int mye = (lane < nb) ? invb[blk + lane] : 0;      // 64 edge ids with one coalesced load
        for (int s0 = 0; s0 < nb; s0 += G * U) {
          float v[U][4];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int ei = s0 + u * G + g;
            const int e = __shfl(mye, ei & (kWave - 1));
#pragma unroll
            for (int q = 0; q < 4; ++q) v[u][q] = 0.f;
            if (ei < nb && act) load_vec<4>(v[u], dzb + static_cast<int64_t>(e) * C);
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] += v[u][q];
          }
        }
      }
#pragma unroll
      for (int off = LPR; off < kWave; off <<= 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] += __shfl_xor(acc[q], off);
      }
      if (g == 0 && act) store_vec<4>(dQ + (static_cast<int64_t>(b) * N + j) * ldq + c0, acc);
    }
  }
}

int bwd_nsplit(int B, int N, int C) {
  if (static_cast<size_t>(N) * kBwdSlice * 4 > 158u * 1024u) return 0;  // slice does not fit LDS: atomic path
  // one 128 KB workgroup per CU: aim for exactly one round of <= 256 workgroups
  const int base = B * ((C + kBwdSlice - 1) / kBwdSlice);
  int ns = num_cus() / base;
  if (ns < 1) ns = 1;
  if (ns > 16) ns = 16;
  while (ns > 1 && (N + ns - 1) / ns < 64) --ns;
  return ns;
}

int edge_lpr(int C) {
  const int need = (C + 3) / 4;
  int lpr = 4;
  while (lpr < need && lpr < kWave) lpr <<= 1;
  return lpr;
}

int edge_grid(int64_t total_pts, int lpr) {
  const int G = kWave / lpr;
  const int64_t groups = (total_pts + G - 1) / G;
  int64_t wgs = (groups + kWavesPerWg - 1) / kWavesPerWg;
  if (wgs > 2048) wgs = 2048;
  if (wgs < 1) wgs = 1;
  return static_cast<int>(wgs);
}

template <bool BWD>
void launch_edge(const EdgeParams& E, int lpr, int grid, hipStream_t s) {
  switch (lpr) {
    case 4: hipLaunchKernelGGL((dense_edge_kernel<4, BWD>), dim3(grid), dim3(kWgThreads), 0, s, E); break;
    case 8: hipLaunchKernelGGL((dense_edge_kernel<8, BWD>), dim3(grid), dim3(kWgThreads), 0, s, E); break;
    case 16: hipLaunchKernelGGL((dense_edge_kernel<16, BWD>), dim3(grid), dim3(kWgThreads), 0, s, E); break;
    case 32: hipLaunchKernelGGL((dense_edge_kernel<32, BWD>), dim3(grid), dim3(kWgThreads), 0, s, E); break;
    default: hipLaunchKernelGGL((dense_edge_kernel<64, BWD>), dim3(grid), dim3(kWgThreads), 0, s, E); break;
  }
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace
}  // namespace dgcn

using namespace dgcn;

extern "C" int dgcn_vertex_gemm_f32(const float* x, int64_t sb, int64_t sc, int64_t sn, int32_t B,
                                    int32_t C, int32_t N, const float* W, const float* bias,
                                    int32_t M, float* out, void* stream) {
  if (!x || !W || !out) return DGCN_E_NULL;
  if (B < 0 || C <= 0 || N <= 0 || M <= 0) return DGCN_E_SHAPE;
  if (B == 0) return DGCN_OK;
  GemmParams P{x, sb, sc, sn, B, C, N, M, W, bias, out, 0};
  const int grid = gemm_grid(static_cast<int64_t>(B) * ((N + 15) / 16));
  hipLaunchKernelGGL(vertex_gemm_kernel, dim3(grid), dim3(kWgThreads), 0, static_cast<hipStream_t>(stream), P);
  return launch_status();
}

// P/Q producer of EdgeConv2d straight from the Conv2d parameters: conv_w [Cout][2C] (the (Cout,2C,1,1)
// weight), bias [Cout] or NULL;  out [B,N,2*Cout] = [ (W1-W2) x + b | W2 x ].
extern "C" int dgcn_edgeconv_pq_f32(const float* x, int64_t sb, int64_t sc, int64_t sn, int32_t B, int32_t C,
                                    int32_t N, const float* conv_w, const float* bias, int32_t Cout,
                                    float* out, void* stream) {
  if (!x || !conv_w || !out) return DGCN_E_NULL;
  if (B < 0 || C <= 0 || N <= 0 || Cout <= 0) return DGCN_E_SHAPE;
  if (B == 0) return DGCN_OK;
  GemmParams P{x, sb, sc, sn, B, C, N, 2 * Cout, conv_w, bias, out, 1};
  const int grid = gemm_grid(static_cast<int64_t>(B) * ((N + 15) / 16));
  hipLaunchKernelGGL(vertex_gemm_kernel, dim3(grid), dim3(kWgThreads), 0, static_cast<hipStream_t>(stream), P);
  return launch_status();
}

// Number of workgroups the forward uses = rows of the `stats` partial buffer [n][2][C].
extern "C" int32_t dgcn_dense_edge_reduce_num_partials(int32_t B, int32_t N, int32_t C) {
  if (B <= 0 || N <= 0 || C <= 0) return 0;
  return edge_grid(static_cast<int64_t>(B) * N, edge_lpr(C));
}

extern "C" int dgcn_dense_edge_reduce_fwd_f32(const float* P, int64_t ldp, const float* Q, int64_t ldq,
                                              const int64_t* idx, int64_t idx_sb, int64_t idx_sn, int64_t idx_sk,
                                              int32_t B, int32_t N, int32_t C, int32_t k, int32_t act,
                                              float slope, float* vmax, float* vmin, uint8_t* amax,
                                              uint8_t* amin, float* stats, void* stream) {
  if (!Q || !idx || !vmax) return DGCN_E_NULL;
  if (B < 0 || N <= 0 || C <= 0 || k <= 0 || k > 255) return DGCN_E_SHAPE;
  if (C % 4 != 0 || ldq < C || ldq % 4 != 0 || (P && (ldp < C || ldp % 4 != 0))) return DGCN_E_SHAPE;
  if (act < ACT_NONE || act > ACT_LEAKY) return DGCN_E_MODE;
  if (!al16(Q) || (P && !al16(P)) || !al16(vmax) || (vmin && !al16(vmin)) || (stats && !al16(stats)))
    return DGCN_E_ALIGN;
  if (B == 0) return DGCN_OK;
  EdgeParams E{};
  E.P = P; E.Q = Q; E.ldp = ldp; E.ldq = ldq; E.idx = idx; E.ib = idx_sb; E.in_ = idx_sn; E.ik = idx_sk;
  E.B = B; E.N = N; E.C = C; E.k = k; E.act = act; E.slope = slope;
  E.vmax = vmax; E.vmin = vmin; E.amax = amax; E.amin = amin; E.stats = stats;
  const int lpr = edge_lpr(C);
  launch_edge<false>(E, lpr, edge_grid(static_cast<int64_t>(B) * N, lpr), static_cast<hipStream_t>(stream));
  return launch_status();
}

// Number of point splits of the LDS-accumulating backward = leading dimension of `dq_parts`
// [nsplit][B][N][C]; 0 means the slice does not fit LDS and the atomic path (dQ) must be used.
extern "C" int32_t dgcn_dense_edge_reduce_bwd_nsplit(int32_t B, int32_t N, int32_t C) {
  if (B <= 0 || N <= 0 || C <= 0) return 0;
  return bwd_nsplit(B, N, C);
}

// dL/da_e = gmax[b,n,c]*[l==amax] + gmin[b,n,c]*[l==amin] + gsum[c] + 2 a_e gsq[c];  dz = dL/da * act'(z)
// dP[b,n,:] = sum_l dz (overwritten);  dQ[b,j,:] += dz (hardware fp32 atomics; dQ must be zeroed by the caller)
extern "C" int dgcn_dense_edge_reduce_bwd_f32(const float* P, int64_t ldp, const float* Q, int64_t ldq,
                                              const int64_t* idx, int64_t idx_sb, int64_t idx_sn, int64_t idx_sk,
                                              int32_t B, int32_t N, int32_t C, int32_t k, int32_t act,
                                              float slope, const uint8_t* amax, const uint8_t* amin,
                                              const float* gmax, const float* gmin, const float* gsum,
                                              const float* gsq, const float* sel_scale, float* dP, float* dQ,
                                              float* dq_parts, int32_t nsplit, void* stream) {
  if (!Q || !idx || !amax || !gmax || (!dQ && !dq_parts)) return DGCN_E_NULL;
  if (gmin && !amin) return DGCN_E_NULL;
  if (sel_scale && !al16(sel_scale)) return DGCN_E_ALIGN;
  if (B < 0 || N <= 0 || C <= 0 || k <= 0 || k > 255) return DGCN_E_SHAPE;
  if (C % 4 != 0 || ldq < C || ldq % 4 != 0 || (P && (ldp < C || ldp % 4 != 0))) return DGCN_E_SHAPE;
  if (act < ACT_NONE || act > ACT_LEAKY) return DGCN_E_MODE;
  if (!al16(Q) || (P && !al16(P)) || !al16(gmax) || (gmin && !al16(gmin)) || (gsum && !al16(gsum)) ||
      (gsq && !al16(gsq)) || (dP && !al16(dP)) || (dQ && !al16(dQ)) || (dq_parts && !al16(dq_parts)))
    return DGCN_E_ALIGN;
  if (B == 0) return DGCN_OK;
  EdgeParams E{};
  E.P = P; E.Q = Q; E.ldp = ldp; E.ldq = ldq; E.idx = idx; E.ib = idx_sb; E.in_ = idx_sn; E.ik = idx_sk;
  E.B = B; E.N = N; E.C = C; E.k = k; E.act = act; E.slope = slope;
  E.amax = const_cast<uint8_t*>(amax); E.amin = const_cast<uint8_t*>(amin);
  E.gmax = gmax; E.gmin = gmin; E.gsum = gsum; E.gsq = gsq; E.selscale = sel_scale; E.dP = dP; E.dQ = dQ;
  if (dq_parts) {
    if (nsplit < 1 || nsplit != bwd_nsplit(B, N, C)) return DGCN_E_WORKSPACE;
    const size_t lds = static_cast<size_t>(N) * kBwdSlice * sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dense_edge_bwd_lds_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return static_cast<int>(e);
    const dim3 grid(B, (C + kBwdSlice - 1) / kBwdSlice, nsplit);
    hipLaunchKernelGGL(dense_edge_bwd_lds_kernel, grid, dim3(kBwdLdsThreads), lds, static_cast<hipStream_t>(stream),
                       E, dq_parts, static_cast<int>(nsplit));
    return launch_status();
  }
  const int lpr = edge_lpr(C);
  launch_edge<true>(E, lpr, edge_grid(static_cast<int64_t>(B) * N, lpr), static_cast<hipStream_t>(stream));
  return launch_status();
}

namespace {
struct InvWs {
  float* dz;
  int32_t *cnt, *ptr, *rank, *inv;
  size_t bytes;
};

InvWs inv_layout(void* base, int64_t B, int64_t N, int64_t C, int64_t k) {
  auto up = [](size_t v) { return (v + 15u) / 16u * 16u; };
  InvWs w;
  size_t off = 0;
  unsigned char* p = static_cast<unsigned char*>(base);
  w.dz = reinterpret_cast<float*>(p + off); off += up(static_cast<size_t>(B * N * k * C) * 4u);
  w.cnt = reinterpret_cast<int32_t*>(p + off); off += up(static_cast<size_t>(B * N) * 4u);
  w.ptr = reinterpret_cast<int32_t*>(p + off); off += up(static_cast<size_t>(B * (N + 1)) * 4u);
  w.rank = reinterpret_cast<int32_t*>(p + off); off += up(static_cast<size_t>(B * N * k) * 4u);
  w.inv = reinterpret_cast<int32_t*>(p + off); off += up(static_cast<size_t>(B * N * k) * 4u);
  w.bytes = off;
  return w;
}
}  // namespace

extern "C" size_t dgcn_dense_edge_reduce_bwd_inv_workspace_bytes(int32_t B, int32_t N, int32_t C, int32_t k) {
  if (B <= 0 || N <= 0 || C <= 0 || k <= 0) return 0;
  return inv_layout(nullptr, B, N, C, k).bytes;
}

// Same contract as dgcn_dense_edge_reduce_bwd_f32, with dQ produced through inverse neighbour lists: dP and dQ are
// fully overwritten (no pre-zeroing), workspace >= dgcn_dense_edge_reduce_bwd_inv_workspace_bytes.
extern "C" int dgcn_dense_edge_reduce_bwd_inv_f32(const float* P, int64_t ldp, const float* Q, int64_t ldq,
                                                  const int64_t* idx, int64_t idx_sb, int64_t idx_sn, int64_t idx_sk,
                                                  int32_t B, int32_t N, int32_t C, int32_t k, int32_t act,
                                                  float slope, const uint8_t* amax, const uint8_t* amin,
                                                  const float* gmax, const float* gmin, const float* gsum,
                                                  const float* gsq, const float* sel_scale, float* dP, float* dQ,
                                                  void* workspace, size_t workspace_bytes, void* stream) {
  if (!Q || !idx || !amax || !gmax || !dQ || !workspace) return DGCN_E_NULL;
  if (gmin && !amin) return DGCN_E_NULL;
  if (B < 0 || N <= 0 || C <= 0 || k <= 0 || k > 255) return DGCN_E_SHAPE;
  if (C % 4 != 0 || ldq < C || ldq % 4 != 0 || (P && (ldp < C || ldp % 4 != 0))) return DGCN_E_SHAPE;
  if (act < ACT_NONE || act > ACT_LEAKY) return DGCN_E_MODE;
  if (!al16(Q) || (P && !al16(P)) || !al16(gmax) || (gmin && !al16(gmin)) || (gsum && !al16(gsum)) ||
      (gsq && !al16(gsq)) || (dP && !al16(dP)) || !al16(dQ) || !al16(workspace) || (sel_scale && !al16(sel_scale)))
    return DGCN_E_ALIGN;
  if (workspace_bytes < dgcn_dense_edge_reduce_bwd_inv_workspace_bytes(B, N, C, k)) return DGCN_E_WORKSPACE;
  if (B == 0) return DGCN_OK;
  const InvWs W = inv_layout(workspace, B, N, C, k);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (const int zrc = zero_async(W.cnt, static_cast<size_t>(B) * N * sizeof(int32_t), s)) return zrc;   // a kernel, not a memset node: dgcn_common.h
  EdgeParams E{};
  E.P = P; E.Q = Q; E.ldp = ldp; E.ldq = ldq; E.idx = idx; E.ib = idx_sb; E.in_ = idx_sn; E.ik = idx_sk;
  E.B = B; E.N = N; E.C = C; E.k = k; E.act = act; E.slope = slope;
  E.amax = const_cast<uint8_t*>(amax); E.amin = const_cast<uint8_t*>(amin);
  E.gmax = gmax; E.gmin = gmin; E.gsum = gsum; E.gsq = gsq; E.selscale = sel_scale; E.dP = dP; E.dQ = nullptr;
  E.dz = W.dz; E.inv_cnt = W.cnt; E.inv_rank = W.rank;
  const int lpr = edge_lpr(C);
  launch_edge<true>(E, lpr, edge_grid(static_cast<int64_t>(B) * N, lpr), s);
  const int64_t edges = static_cast<int64_t>(B) * N * k;
  int64_t eg = (edges + kWgThreads - 1) / kWgThreads;
  if (eg > 4096) eg = 4096;
  hipLaunchKernelGGL(inv_scan_kernel, dim3(B), dim3(kBwdLdsThreads), 0, s, W.cnt, N, W.ptr);
  hipLaunchKernelGGL(inv_fill_kernel, dim3(static_cast<unsigned>(eg)), dim3(kWgThreads), 0, s, idx, idx_sb, idx_sn,
                     idx_sk, B, N, k, W.ptr, W.rank, W.inv);
  int64_t gw = (static_cast<int64_t>(B) * N + kWavesPerWg - 1) / kWavesPerWg;   // one wave per neighbour row
  if (gw > 8192) gw = 8192;
  const int grid = static_cast<int>(gw);
  switch (lpr) {
    case 4: hipLaunchKernelGGL(inv_gather_kernel<4>, dim3(grid), dim3(kWgThreads), 0, s, W.dz, W.ptr, W.inv, B, N, C, k, dQ, ldq); break;
    case 8: hipLaunchKernelGGL(inv_gather_kernel<8>, dim3(grid), dim3(kWgThreads), 0, s, W.dz, W.ptr, W.inv, B, N, C, k, dQ, ldq); break;
    case 16: hipLaunchKernelGGL(inv_gather_kernel<16>, dim3(grid), dim3(kWgThreads), 0, s, W.dz, W.ptr, W.inv, B, N, C, k, dQ, ldq); break;
    case 32: hipLaunchKernelGGL(inv_gather_kernel<32>, dim3(grid), dim3(kWgThreads), 0, s, W.dz, W.ptr, W.inv, B, N, C, k, dQ, ldq); break;
    default: hipLaunchKernelGGL(inv_gather_kernel<64>, dim3(grid), dim3(kWgThreads), 0, s, W.dz, W.ptr, W.inv, B, N, C, k, dQ, ldq); break;
  }
  return launch_status();
}

