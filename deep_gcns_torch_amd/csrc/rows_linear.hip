// Node-wise Linear layers of the sparse path for gfx950: Y = X W^T + b [+ R] on the bf16 matrix pipe, fp32-faithful
// ("bf16x6", bf16x6.h), with the neighbouring node-wise passes folded into the same sweep over the rows.
//
// Replaces, around GENConv:
//   the Linear stages of MLP                    gcn_lib/sparse/torch_nn.py:50-71
//   mlp(x + m) and the 'res+' residual  h = conv(h2) + h      gcn_lib/sparse/torch_vertex.py:70-76,
//                                                              examples/ogb/ogbn_arxiv/model.py:90-106
//   the BatchNorm statistics pass of the NEXT norm layer (per-workgroup partial sum y, sum y^2: the layout
//   dgcn_bn_finalize_f32 consumes), and in the backward launch (dX = G W, w_trans = 1) the bias gradient
//   (per-workgroup partial column sums of G).
//
// Shape of the work: rows >> K, C (1.7e5 .. 2.4e6 rows, K and C <= 256): HBM streaming, 2*rows*K*C flop.  The stock
// fp32 GEMM runs this at ~60 TF (92 us for 169,343 x 128 x 128); streaming X, R and Y once is ~45 us.
//   * one persistent workgroup (8 waves) per CU; the weight matrix is split ONCE per workgroup into three bf16 planes
//     in LDS (row stride 4*KC+2 sixteen-byte units: conflict-free ds_read_b128 for the fixed lane groups);
//   * a wave owns 16-row batches: lane (m, kq) reads 32 bytes of row m per 32-column block (two dwordx4), splits them
//     exactly into three bf16 fragments in registers, and runs the six-product chain of bf16x6.h against the weight
//     fragments, tile by tile (consecutive MFMAs never share an accumulator); the next batch's rows and residual are
//     requested before the chain starts;
//   * the accumulators start from bias + residual; the D layout (lane (n, q): rows 4q+j, channel 16 ct + n) is stored
//     as 64-byte segments; per-channel sums for the next BatchNorm stay in registers until the workgroup ends.
// Deterministic (fixed summation orders); the library allocates nothing.

#include "bf16x6.h"

namespace dgcn {
namespace {

constexpr int kRlWaves = 8;
constexpr int kRlThreads = kRlWaves * kWave;
constexpr int kRlM = 16;
constexpr int kRlLdsBytes = 160 * 1024;

struct RlParams {
  const float* x;
  int64_t ldx;
  int64_t rows;
  const float* w;
  int64_t ldw;
  int w_trans;        // 0: w[c * ldw + k] (nn.Linear weight, Y = X W^T); 1: w[k * ldw + c] (Y = X W: the input gradient)
  const float* bias;  // [C] or null
  const float* res;   // [rows, C] (row stride ldr) added to the result, or null
  int64_t ldr;
  float* y;
  int64_t ldy;
  int K, C;
  int relu;
  int late;           // the residual joins AFTER the product chain (EPI 3): res + (x W^T + bias) rounded once, as an
                      // elementwise pass behind the GEMM would -- the reversible couplings, whose 112 layers would
                      // otherwise accumulate one rounding at the residual's magnitude per MFMA of every layer
  int negate;         // (late only) y = res - (x W^T + bias): the inverse of the additive coupling, x_i = y_i - F_i(.)
  float* col_stats;   // [gridDim.x][2][C] partial sum y | sum y^2, or null
  float* xcol_sum;    // [gridDim.x][K] partial column sums of X (XSUM kernels), or null
};

// EPI: 0 = bias + residual; 1 = the same + per-channel sum y | sum y^2 partials; 2 = partial column sums of X (the bias
// gradient when X is the upstream gradient), no residual; 3 = residual added / subtracted from behind the product chain
template <int NT, int KC, int EPI>
__global__ __launch_bounds__(kRlThreads) void rows_linear_kernel(const RlParams P) {
  constexpr bool STATS = EPI == 1;
  constexpr bool XSUM = EPI == 2;
  constexpr bool LATE = EPI == 3;
  constexpr bool HALF = NT >= 8;            // weight fragments in one half-and-half buffer instead of two full ones
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int SU = 4 * KC + 2;            // row stride of a weight plane in 16-byte units
  constexpr int PLANE = NT * 16 * SU;
  const int K = P.K, C = P.C;
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int cbase = blockIdx.y * (NT * 16);  // first output channel of this workgroup
  i4v* Wp = reinterpret_cast<i4v*>(smem);    // [3][NT*16][SU] units of 8 bf16
  {
    const bool vec_ok = !P.w_trans && (P.ldw % 4 == 0) && ((reinterpret_cast<uintptr_t>(P.w) & 15u) == 0);
    for (int idx = threadIdx.x; idx < NT * 16 * 4 * KC; idx += kRlThreads) {
      // transposed source: consecutive threads take consecutive output channels (coalesced for a fixed k)
      const int r = P.w_trans ? idx % (NT * 16) : idx / (4 * KC);
      const int g8 = P.w_trans ? idx / (NT * 16) : idx % (4 * KC);
      const int ch = cbase + r;
      f4v f0 = {0.f, 0.f, 0.f, 0.f}, f1 = {0.f, 0.f, 0.f, 0.f};
      if (ch < C) {
        const int k0 = 8 * g8;
        if (vec_ok) {
          const float* wr = P.w + static_cast<int64_t>(ch) * P.ldw + k0;
          if (k0 < K) f0 = *reinterpret_cast<const f4v*>(wr);          // K % 4 == 0
          if (k0 + 4 < K) f1 = *reinterpret_cast<const f4v*>(wr + 4);
        } else {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int k = k0 + e;
            v[e] = 0.f;
            if (k < K) v[e] = P.w_trans ? P.w[static_cast<int64_t>(k) * P.ldw + ch] : P.w[static_cast<int64_t>(ch) * P.ldw + k];
          }
          f0 = f4v{v[0], v[1], v[2], v[3]};
          f1 = f4v{v[4], v[5], v[6], v[7]};
        }
      }
      i4v h, m, l;
      eg_split3(f0, f1, h, m, l);
      Wp[r * SU + g8] = h;
      Wp[PLANE + r * SU + g8] = m;
      Wp[2 * PLANE + r * SU + g8] = l;
    }
  }
  __syncthreads();

  const int n = lane & 15;
  const int kq = lane >> 4;
  const int64_t nbatch = (P.rows + kRlM - 1) / kRlM;
  const int64_t bstride = static_cast<int64_t>(gridDim.x) * kRlWaves;
  const i4v* wb = Wp + n * SU + kq;          // + plane * PLANE + ct * 16 * SU + 4 * sb

  float s1[STATS ? NT : 1], s2[STATS ? NT : 1];
#pragma unroll
  for (int ct = 0; ct < (STATS ? NT : 1); ++ct) { s1[ct] = 0.f; s2[ct] = 0.f; }
  float xs[XSUM ? KC : 1][8];
#pragma unroll
  for (int sb = 0; sb < (XSUM ? KC : 1); ++sb) {
#pragma unroll
    for (int e = 0; e < 8; ++e) xs[sb][e] = 0.f;
  }

  // block sb (32 columns) of the rows of batch bt: lane (m = n, kq) owns floats 32 sb + 8 kq .. + 7 of row m
  auto load_block = [&](int64_t bt, int sb, f4v& f0, f4v& f1) {
    int64_t row = bt * kRlM + n;
    if (row > P.rows - 1) row = P.rows - 1;
    const float* p = P.x + row * P.ldx + 32 * sb + 8 * kq;
    const int o = 32 * sb + 8 * kq;
    f0 = (o < K) ? *reinterpret_cast<const f4v*>(p) : f4v{0.f, 0.f, 0.f, 0.f};          // K % 4 == 0
    f1 = (o + 4 < K) ? *reinterpret_cast<const f4v*>(p + 4) : f4v{0.f, 0.f, 0.f, 0.f};
  };
  // residual rows of a batch in the D layout (lane (n, q): rows 4 q + j, channels 16 ct + n)
  auto load_res = [&](int64_t bt, f4v (&rr)[NT], bool late_call) {
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) rr[ct] = f4v{0.f, 0.f, 0.f, 0.f};
    if (XSUM || !P.res || (LATE && !late_call)) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int64_t row = bt * kRlM + 4 * kq + j;
      if (row > P.rows - 1) row = P.rows - 1;
      const float* rp = P.res + row * P.ldr + cbase + n;
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) {
        if (cbase + ct * 16 + n < C) rr[ct][j] = rp[ct * 16];
      }
    }
  };

  int64_t bt = static_cast<int64_t>(blockIdx.x) * kRlWaves + wave;
  f4v ra[KC][2];       // ring: block sb of the NEXT batch is requested into ra[sb] as soon as this batch has split it
  f4v rc[NT];          // residual of the batch about to be computed; re-requested for the next batch at the loop top
  f4v acc[NT];
  i4v bx[HALF ? 1 : 2][NT];
  if (bt < nbatch) {
#pragma unroll
    for (int sb = 0; sb < KC; ++sb) load_block(bt, sb, ra[sb][0], ra[sb][1]);
    load_res(bt, rc, false);
  }
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) bx[0][ct] = wb[ct * 16 * SU];       // plane 1 of block 0

  while (bt < nbatch) {
    const int64_t btn = bt + bstride;
    const int64_t btl = btn < nbatch ? btn : bt;      // past the end: harmless duplicate of this batch
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
      const int ch = cbase + ct * 16 + n;
      const float bv = (P.bias && ch < C) ? P.bias[ch] : 0.f;      // L1-resident; not worth 8 registers
      acc[ct] = rc[ct] + bv;
    }
    load_res(btl, rc, false);
    f4v rl[NT];                                       // (LATE) this batch's residual rows, met again behind the chain
    if constexpr (LATE) load_res(bt, rl, true);
    const bool xs_ok = bt * kRlM + n < P.rows;
    __builtin_amdgcn_sched_barrier(0);

    // ---- tile = bias + residual + X W^T: six bf16 MFMAs per 16x16x32 block (same chain as gen_aggr_egemm.hip) ----
#pragma unroll
    for (int sb = 0; sb < KC; ++sb) {
      constexpr int NB = HALF ? 1 : 2;
      const int bi = (sb & 1) % NB;            // holds plane 1 of this block
      const int bo = (bi ^ 1) % NB;
      i4v a1, a2, a3;
      eg_split3(ra[sb][0], ra[sb][1], a1, a2, a3);
      if constexpr (XSUM) {
        if (xs_ok) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { xs[sb][e] += ra[sb][0][e]; xs[sb][4 + e] += ra[sb][1][e]; }
        }
      }
      load_block(btl, sb, ra[sb][0], ra[sb][1]);                                                 // next batch
      if constexpr (HALF) {
        // eight tiles: one fragment buffer used as two halves (the register-lean chain of gen_aggr_egemm.hip): a half is
        // re-filled with the next plane right after its MFMAs and consumed after the other half's.  Both halves hold
        // plane 1 of this block on entry.
        constexpr int G0 = NT / 2;
        const int sn = (sb + 1 < KC) ? sb + 1 : 0;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const int g0 = hh * G0, g1 = hh ? NT : G0;
#pragma unroll
            for (int g = g0; g < g1; ++g) acc[g] = eg_mfma_bf16(a1, bx[0][g], acc[g]);
            if (pl < 2) {
#pragma unroll
              for (int g = g0; g < g1; ++g) acc[g] = eg_mfma_bf16(a2, bx[0][g], acc[g]);
            }
            if (pl < 1) {
#pragma unroll
              for (int g = g0; g < g1; ++g) acc[g] = eg_mfma_bf16(a3, bx[0][g], acc[g]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = g0; g < g1; ++g) {
              bx[0][g] = (pl < 2) ? wb[(pl + 1) * PLANE + g * 16 * SU + 4 * sb] : wb[g * 16 * SU + 4 * sn];
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      } else {
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) bx[bo][ct] = wb[PLANE + ct * 16 * SU + 4 * sb];            // plane 2
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) acc[ct] = eg_mfma_bf16(a1, bx[bi][ct], acc[ct]);
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) acc[ct] = eg_mfma_bf16(a2, bx[bi][ct], acc[ct]);
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) acc[ct] = eg_mfma_bf16(a3, bx[bi][ct], acc[ct]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) bx[bi][ct] = wb[2 * PLANE + ct * 16 * SU + 4 * sb];        // plane 3
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) acc[ct] = eg_mfma_bf16(a1, bx[bo][ct], acc[ct]);
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) acc[ct] = eg_mfma_bf16(a2, bx[bo][ct], acc[ct]);
      __builtin_amdgcn_sched_barrier(0);
      {
        const int sn = (sb + 1 < KC) ? sb + 1 : 0;                                              // next plane 1
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) bx[bo][ct] = wb[ct * 16 * SU + 4 * sn];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) acc[ct] = eg_mfma_bf16(a1, bx[bi][ct], acc[ct]);
      __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (!HALF && KC % 2 == 1) {
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) bx[0][ct] = bx[1][ct];
    }

    // ---- epilogue: store the D layout, keep the per-channel sums ----
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t row = bt * kRlM + 4 * kq + j;
      if (row < P.rows) {
        float* yp = P.y + row * P.ldy + cbase + n;
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
          if (cbase + ct * 16 + n < C) {
            float v = acc[ct][j];
            if constexpr (LATE) v = P.negate ? rl[ct][j] - v : rl[ct][j] + v;
            if (P.relu) v = fmaxf(v, 0.f);
            yp[ct * 16] = v;
            if constexpr (STATS) {
              s1[ct] += v;
              s2[ct] = fmaf(v, v, s2[ct]);
            }
          }
        }
      }
    }
    bt = btn;
  }

  // ---- workgroup partials (fixed order: lane groups by xor-shuffle, then the eight waves one after the other) ----
  if constexpr (STATS || XSUM) {
    __syncthreads();                               // every wave is done with the weight planes: reuse the LDS
    float* red = smem;                             // [8][2][NT*16]  |  [8][KC*32]
    if constexpr (STATS) {
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) {
        s1[ct] += __shfl_xor(s1[ct], 16); s1[ct] += __shfl_xor(s1[ct], 32);
        s2[ct] += __shfl_xor(s2[ct], 16); s2[ct] += __shfl_xor(s2[ct], 32);
        if (kq == 0) {
          red[(wave * 2) * (NT * 16) + ct * 16 + n] = s1[ct];
          red[(wave * 2 + 1) * (NT * 16) + ct * 16 + n] = s2[ct];
        }
      }
      __syncthreads();
      for (int i = threadIdx.x; i < 2 * NT * 16; i += kRlThreads) {
        const int which = i / (NT * 16), cc = i % (NT * 16);
        const int ch = cbase + cc;
        if (ch < C) {
          float t = 0.f;
#pragma unroll
          for (int wv = 0; wv < kRlWaves; ++wv) t += red[(wv * 2 + which) * (NT * 16) + cc];
          P.col_stats[(static_cast<int64_t>(blockIdx.x) * 2 + which) * C + ch] = t;
        }
      }
      __syncthreads();
    }
    if constexpr (XSUM) {
      if (blockIdx.y == 0) {
        float* redx = smem;
#pragma unroll
        for (int sb = 0; sb < KC; ++sb) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float v = xs[sb][e];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
            if (n == 0) redx[wave * (KC * 32) + 32 * sb + 8 * kq + e] = v;
          }
        }
        __syncthreads();
        for (int k = threadIdx.x; k < K; k += kRlThreads) {
          float t = 0.f;
#pragma unroll
          for (int wv = 0; wv < kRlWaves; ++wv) t += redx[wv * (KC * 32) + k];
          P.xcol_sum[static_cast<int64_t>(blockIdx.x) * K + k] = t;
        }
      }
    }
  }
}

struct RlShape {
  int nt, kc, ncols;     // channel tiles per workgroup, 32-column blocks, workgroup columns (gridDim.y)
  size_t lds;
};

inline bool rl_shape(int K, int C, RlShape* S) {
  if (K < 16 || K > 256 || K % 4 != 0 || C < 1 || C > 2048) return false;
  const int kc = K <= 64 ? 2 : (K <= 128 ? 4 : 8);
  const int tiles = (C + 15) / 16;
  int nt = (kc == 8 || tiles <= 4) ? 4 : 8;
  S->nt = nt;
  S->kc = kc;
  S->ncols = (tiles + nt - 1) / nt;
  S->lds = static_cast<size_t>(3) * nt * 16 * (4 * kc + 2) * 16;
  return S->lds <= static_cast<size_t>(kRlLdsBytes);
}

inline int rl_grid_x(int64_t rows, int ncols) {
  const int64_t nbatch = (rows + kRlM - 1) / kRlM;
  int64_t gx = (nbatch + kRlWaves - 1) / kRlWaves;
  int64_t cap = num_cus() / ncols;
  if (cap < 1) cap = 1;
  if (gx > cap) gx = cap;
  return static_cast<int>(gx < 1 ? 1 : gx);
}

template <int NT, int KC, int EPI>
int rl_launch_epi(const RlParams& P, const RlShape& S, hipStream_t s) {
  const dim3 grid(rl_grid_x(P.rows, S.ncols), S.ncols);
  const void* fn = reinterpret_cast<const void*>(rows_linear_kernel<NT, KC, EPI>);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(S.lds));
  if (e != hipSuccess) return static_cast<int>(e);
  hipLaunchKernelGGL((rows_linear_kernel<NT, KC, EPI>), grid, dim3(kRlThreads), S.lds, s, P);
  return launch_status();
}

template <int NT, int KC>
int rl_launch(const RlParams& P, const RlShape& S, hipStream_t s) {
  if (P.late) return rl_launch_epi<NT, KC, 3>(P, S, s);
  if (P.xcol_sum) return rl_launch_epi<NT, KC, 2>(P, S, s);
  if (P.col_stats) return rl_launch_epi<NT, KC, 1>(P, S, s);
  return rl_launch_epi<NT, KC, 0>(P, S, s);
}

}  // namespace
}  // namespace dgcn

using namespace dgcn;

extern "C" int32_t dgcn_rows_linear_supported(int32_t K, int32_t C) {
  RlShape S;
  return rl_shape(K, C, &S) ? 1 : 0;
}

extern "C" int32_t dgcn_rows_linear_num_partials(int64_t rows, int32_t K, int32_t C) {
  RlShape S;
  if (rows <= 0 || !rl_shape(K, C, &S)) return 0;
  return rl_grid_x(rows, S.ncols);
}

extern "C" int dgcn_rows_linear_f32(const float* x, int64_t ldx, int64_t rows, const float* w, int64_t ldw,
                                    int32_t w_trans, const float* bias, const float* res, int64_t ldr, float* y,
                                    int64_t ldy, int32_t K, int32_t C, int32_t relu, float* col_stats,
                                    float* xcol_sum, void* stream) {
  if (!x || !w || !y) return DGCN_E_NULL;
  RlShape S;
  if (rows < 0 || !rl_shape(K, C, &S)) return DGCN_E_SHAPE;
  if (xcol_sum && (res || col_stats)) return DGCN_E_MODE;       // the column-sum launch is the plain input-gradient GEMM
  if (relu < 0 || relu > 7 || ((relu & 6) && (xcol_sum || col_stats || (relu & 1) || !res))) return DGCN_E_MODE;
  if (ldx < K || ldy < C || (res && ldr < C) || ldw < (w_trans ? C : K)) return DGCN_E_SHAPE;
  if ((reinterpret_cast<uintptr_t>(x) & 15u) || ldx % 4 != 0) return DGCN_E_ALIGN;
  if (rows == 0) return DGCN_OK;
  RlParams P;
  P.x = x; P.ldx = ldx; P.rows = rows; P.w = w; P.ldw = ldw; P.w_trans = w_trans ? 1 : 0;
  P.bias = bias; P.res = res; P.ldr = ldr; P.y = y; P.ldy = ldy; P.K = K; P.C = C;
  P.relu = (relu & 1) ? 1 : 0; P.negate = (relu & 2) ? 1 : 0; P.late = (relu & 6) ? 1 : 0;
  P.col_stats = col_stats; P.xcol_sum = xcol_sum;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (S.nt == 4) {
    switch (S.kc) {
      case 2: return rl_launch<4, 2>(P, S, s);
      case 4: return rl_launch<4, 4>(P, S, s);
      default: return rl_launch<4, 8>(P, S, s);
    }
  }
  switch (S.kc) {
    case 2: return rl_launch<8, 2>(P, S, s);
    default: return rl_launch<8, 4>(P, S, s);
  }
}
