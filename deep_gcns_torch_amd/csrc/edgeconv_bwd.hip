// Parameter and input gradients of the dense EdgeConv2d layer from dP | dQ, for gfx950 (MI355X).
//
// The layer computes a_{bnl} = act(P_{bn} + Q_{b,j(b,n,l)}) with P = (W1 - W2) x + b, Q = W2 x, W = [W1 | W2] the
// (Cout, 2C, 1, 1) weight of the 1x1 Conv2d inside BasicConv (gcn_lib/dense/torch_vertex.py:31-35,
// gcn_lib/dense/torch_nn.py:48-60; SURVEY.md Appendix A "Dense EdgeConv").  dense_edge.hip's backward leaves
// dPQ [B][N][2 Cout] = [dP | dQ]; what remains is
//     dx  = (W1 - W2)^T dP + W2^T dQ  (+ res_scale * g, the skip connection of ResDynBlock2d, torch_vertex.py:101)
//     dW1 = dP^T x,   dW2 = (dQ - dP)^T x,   db = sum_{b,n} dP
// which rounds 1 - 4 ran as library calls: sub + cat + baddbmm for dx, permute-copy + split-K bmm + sum + sub + cat
// + sum for the parameters (12 launches and ~95 us per layer, 336 of the 813 launches of a ResGCN-28 step).
// Here: one kernel for dx, one for the per-workgroup partials of [dW | db] in the weight's own layout (summed in
// fixed order by dgcn_reduce_partials_f32).  Both run on v_mfma_f32_16x16x4_f32: an exact fp32 fma chain.

#include "dgcn_common.h"

namespace dgcn {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kEbKC = 128;       // contraction indices staged in LDS per step
constexpr int kEbPad = 16;       // row padding (floats) of the LDS tiles read as wt[4 s + kq][16 t + i]: the four lane groups
                                 // kq of one ds_read land 16 banks apart (two lanes per bank: the minimum for 64 lanes)
constexpr int kEbPadT = 4;       // ... of the tile read as xs[16 t + i][4 s + kq]: rows 4 banks apart

struct EbInputParams {
  const float* dpq;     // [B][N][2 Cout]
  const float* w;       // [Cout][2C]
  const float* g;       // upstream gradient (B, C, N) with element strides, or null
  int64_t gsb, gsc, gsn;
  float res_scale;
  float* dx;            // [B][C][N]
  int B, C, N, Cout;
};

// effective operand of dx = Wc^T dPQ:  Wc[k][c] = W1[k][c] - W2[k][c] (k < Cout), W2[k - Cout][c] (k >= Cout)
// (no branch and no select around the loads -- the value is masked with integer bits -- so that the 32 x 2 loads per
// thread of a tile fill are issued together instead of one L2 round trip after the other)
__device__ __forceinline__ float eb_wc(const float* __restrict__ w, int C, int Cout, int k, int c) {
  const bool ok = k < 2 * Cout && c < C;
  const bool top = k < Cout;
  const int kr = ok ? (top ? k : k - Cout) : 0;
  const int cr = ok ? c : 0;
  const float* wr = w + static_cast<int64_t>(kr) * (2 * C);
  const float w1 = wr[cr], w2 = wr[C + cr];
  const float v = top ? w1 - w2 : w2;
  return __uint_as_float(__float_as_uint(v) & (ok ? 0xFFFFFFFFu : 0u));
}

// One wave: 16 points x 64 channels (four 16x16 tiles), D[i = channel][j = point] so that the 16 lanes of a register
// store 16 consecutive points of one channel row (dx is channel-major like x).  A[i][k] = Wc[k][c0 + i] from LDS,
// B[k][j] = dPQ[n0 + j][k]: lane (j, kq) reads the 32 consecutive floats k = 32 kq .. + 31 of its point's row (MFMA step
// s contracts k = 32 kq + s over the four lane groups; LDS row 4 s + kq holds that k).
// The workgroups are persistent: with 2 Cout <= 128 (one LDS tile holds all of Wc for 64 channels) a workgroup forms the
// tile once and walks its share of the point tiles under it; wider layers re-form the tile per point-tile group.
__global__ __launch_bounds__(kWgThreads, 2) void edgeconv_bwd_input_kernel(const EbInputParams Pin) {
  // (the lambdas below capture by reference: a kernel-argument struct whose address is taken moves to scratch memory
  //  and every field access becomes a scratch load; plain locals stay in registers)
  struct {
    const float* __restrict__ dpq; const float* __restrict__ w; const float* __restrict__ g;
    int64_t gsb, gsc, gsn; float res_scale; float* __restrict__ dx; int B, C, N, Cout;
  } const P{Pin.dpq, Pin.w, Pin.g, Pin.gsb, Pin.gsc, Pin.gsn, Pin.res_scale, Pin.dx, Pin.B, Pin.C, Pin.N, Pin.Cout};
  __shared__ __attribute__((aligned(16))) float wt[kEbKC][64 + kEbPad];
  __shared__ __attribute__((aligned(16))) float ot[64][64 + kEbPadT];   // [channel][point] result tile of a group
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int tiles_n = (P.N + 15) / 16;
  const int total_tiles = P.B * tiles_n;
  const int groups = (total_tiles + kWavesPerWg - 1) / kWavesPerWg;
  const int li = lane & 15, lk = lane >> 4;
  const int K = 2 * P.Cout;
  const bool vec = (K % 4 == 0) && ((reinterpret_cast<uintptr_t>(P.dpq) & 15u) == 0);
  const bool single = K <= kEbKC;

  // LDS row of contraction index kk (see the kernel comment): 4 s + kq with kk = 32 kq + s
  auto wt_row = [](int kk) { return (kk & 31) * 4 + (kk >> 5); };
  const bool wvec = single && P.C % 4 == 0 && (reinterpret_cast<uintptr_t>(P.w) & 15u) == 0;
  auto fill = [&](int kc, int c0) {
    __syncthreads();
    if (wvec) {
      // 2 Cout <= 128: a wave reads four rows of W1 and of W2 (64 channels = 256 contiguous bytes each) per pass, all
      // eight 16-byte loads of a thread in flight together, and writes rows r (W1 - W2) and Cout + r (W2) of the tile
      const int c = c0 + (lane & 15) * 4;
      const bool cok = c < P.C;                                  // (C % 4 == 0: four channels in or out together)
      float4 a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = i * 16 + (threadIdx.x >> 4);
        const bool ok = cok && r < P.Cout;
        const float* wr = P.w + static_cast<int64_t>(ok ? r : 0) * (2 * P.C) + (ok ? c : 0);
        a[i] = *reinterpret_cast<const float4*>(wr);
        b[i] = *reinterpret_cast<const float4*>(wr + P.C);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = i * 16 + (threadIdx.x >> 4);
        if (r < P.Cout) {
          const float4 t4 = cok ? make_float4(a[i].x - b[i].x, a[i].y - b[i].y, a[i].z - b[i].z, a[i].w - b[i].w)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
          const float4 b4 = cok ? b[i] : make_float4(0.f, 0.f, 0.f, 0.f);
          *reinterpret_cast<float4*>(&wt[wt_row(r)][(lane & 15) * 4]) = t4;
          *reinterpret_cast<float4*>(&wt[wt_row(P.Cout + r)][(lane & 15) * 4]) = b4;
        }
      }
      // rows 2 Cout .. 127 of the tile: zeros
      for (int e = threadIdx.x; e < (kEbKC - 2 * P.Cout) * 64; e += kWgThreads) wt[wt_row(2 * P.Cout + (e >> 6))][e & 63] = 0.f;
    } else {
      for (int e = threadIdx.x; e < kEbKC * 64; e += kWgThreads) {
        const int kk = e >> 6, cc = e & 63;
        wt[wt_row(kk)][cc] = eb_wc(P.w, P.C, P.Cout, kc + kk, c0 + cc);
      }
    }
    __syncthreads();
  };
  auto load_d = [&](float (&dv)[32], const float* drow, int kc) {
    const int kb = kc + lk * 32;
    if (vec && kb + 32 <= K) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(drow + kb + 4 * q);
        dv[4 * q] = v.x; dv[4 * q + 1] = v.y; dv[4 * q + 2] = v.z; dv[4 * q + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int s = 0; s < 32; ++s) dv[s] = (kb + s < K) ? drow[kb + s] : 0.f;
    }
  };
  auto contract = [&](f32x4 (&acc)[4], const float (&dv)[32]) {
#pragma unroll
    for (int s = 0; s < 32; ++s) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float av = wt[4 * s + lk][t * 16 + li];             // rows past K and columns past C hold zeros
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, dv[s], acc[t], 0, 0, 0);
      }
    }
  };
  auto store = [&](const f32x4 (&acc)[4], int b, int n0, int c0) {
    if (n0 + li >= P.N) return;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = c0 + t * 16 + lk * 4 + r;
        if (c < P.C) {
          float v = acc[t][r];
          if (P.g) v = fmaf(P.res_scale, P.g[b * P.gsb + c * P.gsc + (n0 + li) * P.gsn], v);
          P.dx[(static_cast<int64_t>(b) * P.C + c) * P.N + n0 + li] = v;
        }
      }
    }
  };

  // N % 64 == 0: the four point tiles of a group are 64 consecutive points of one sample -- the 64 x 64 result goes
  // through LDS and leaves as whole 256-byte channel rows (the direct store writes 64-byte pieces 16 KB apart)
  const bool staged = P.N % 64 == 0 && (reinterpret_cast<uintptr_t>(P.dx) & 15u) == 0;
  const bool gvec = P.g && P.gsn == 1 && P.gsb % 4 == 0 && P.gsc % 4 == 0 && (reinterpret_cast<uintptr_t>(P.g) & 15u) == 0;
  auto store_staged = [&](const f32x4 (&acc)[4], int grp, int c0) {
    __syncthreads();                                             // the previous group's rows have left the tile
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) ot[t * 16 + lk * 4 + r][wave * 16 + li] = acc[t][r];
    }
    __syncthreads();
    const int tile0 = grp * kWavesPerWg;
    const int b = tile0 / tiles_n;
    const int n0 = (tile0 % tiles_n) * 16;
    for (int e = threadIdx.x; e < 64 * 16; e += kWgThreads) {
      const int cc = e >> 4, q = (e & 15) * 4;
      const int c = c0 + cc;
      if (c >= P.C) continue;
      float4 v = *reinterpret_cast<const float4*>(&ot[cc][q]);
      if (P.g) {
        const float* gp = P.g + b * P.gsb + c * P.gsc + (n0 + q) * P.gsn;
        float4 gv;
        if (gvec) gv = *reinterpret_cast<const float4*>(gp);
        else gv = make_float4(gp[0], gp[P.gsn], gp[2 * P.gsn], gp[3 * P.gsn]);
        v.x = fmaf(P.res_scale, gv.x, v.x); v.y = fmaf(P.res_scale, gv.y, v.y);
        v.z = fmaf(P.res_scale, gv.z, v.z); v.w = fmaf(P.res_scale, gv.w, v.w);
      }
      *reinterpret_cast<float4*>(P.dx + (static_cast<int64_t>(b) * P.C + c) * P.N + n0 + q) = v;
    }
  };

  for (int c0 = 0; c0 < P.C; c0 += 64) {
    bool filled = false;
    for (int grp = blockIdx.x; grp < groups; grp += gridDim.x) {
      const int tile = min(grp * kWavesPerWg + wave, total_tiles - 1);    // clamp: all waves hit the barriers
      const bool tile_ok = grp * kWavesPerWg + wave < total_tiles;
      const int b = tile / tiles_n;
      const int n0 = (tile % tiles_n) * 16;
      const float* drow = P.dpq + (static_cast<int64_t>(b) * P.N + min(n0 + li, P.N - 1)) * K;
      f32x4 acc[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      float dv[32];
      if (single) {
        load_d(dv, drow, 0);                                     // in flight while the weight tile is formed
        if (!filled) fill(0, c0);
        filled = true;
        contract(acc, dv);
      } else {
        for (int kc = 0; kc < K; kc += kEbKC) {
          load_d(dv, drow, kc);
          fill(kc, c0);
          contract(acc, dv);
        }
      }
      if (staged) store_staged(acc, grp, c0);
      else if (tile_ok) store(acc, b, n0, c0);
    }
  }
}

struct EbWeightParams {
  const float* dpq;     // [B*N][2 Cout]
  const float* x;       // (B, C, N) with element strides
  int64_t sb, sc, sn;
  float* part;          // [gridDim.x][Cout * 2C + Cout]: [dW1 | dW2] rows in the Conv2d weight's layout, then db
  int B, C, N, Cout;
  int chunks;           // 64-point chunks of the B*N points
};

constexpr int kEbPts = 64;       // points per chunk

constexpr int kEbWThreads = 512; // 8 waves: two per row tile, each takes half of a chunk's points

// Workgroup (x, y): y = (rg, cg) selects the 64 x 64 block (rows rg*64.. of Cout, channels cg*64.. of C) of dP^T x and
// dQ^T x, x strides over the 64-point chunks.  Waves w and w + 4 own row tile w of the block (points 0..31 / 32..63 of
// every chunk): accumulators for its dP rows and its dQ rows against the four channel tiles (8 tiles), so that
// dW2 = dQ^T x - dP^T x is formed in registers; the two halves meet through LDS at the end.
// A[i][k] = d[point k][row i], B[k][j] = x[point k][channel j], both from LDS tiles staged with coalesced 16-byte loads.
__global__ __launch_bounds__(kEbWThreads) void edgeconv_bwd_weight_kernel(const EbWeightParams P) {
  __shared__ __attribute__((aligned(16))) float ds[kEbPts][128 + kEbPad];   // [point][dP rows of the block | dQ rows]
  __shared__ __attribute__((aligned(16))) float xs[64][kEbPts + kEbPadT];   // [channel][point]: x is channel-major, the
                                                                            // staging writes stay unit-stride
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int rt = wave & 3, half = wave >> 2;
  const int li = lane & 15, lk = lane >> 4;
  const int CG = (P.C + 63) / 64;
  const int rg = blockIdx.y / CG, cg = blockIdx.y % CG;
  const int j0 = rg * 64, c0 = cg * 64;
  const int K = 2 * P.Cout;
  const int64_t total = static_cast<int64_t>(P.B) * P.N;
  const bool dvec = P.Cout % 4 == 0 && (reinterpret_cast<uintptr_t>(P.dpq) & 15u) == 0;
  const bool xvec = P.sn == 1 && P.N % 4 == 0 && P.sb % 4 == 0 && P.sc % 4 == 0 &&
                    (reinterpret_cast<uintptr_t>(P.x) & 15u) == 0;

  f32x4 accp[4], accq[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    accp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    accq[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float bsum = 0.f;
  for (int ch = blockIdx.x; ch < P.chunks; ch += gridDim.x) {
    const int64_t p0 = static_cast<int64_t>(ch) * kEbPts;
    __syncthreads();
    // d tile: 64 points x (64 dP columns | 64 dQ columns); a row of dPQ is contiguous
    if (dvec) {
      for (int e = threadIdx.x; e < kEbPts * 32; e += kEbWThreads) {
        const int pt = e >> 5, col = (e & 31) * 4;
        const int j = j0 + (col & 63);
        const int64_t p = p0 + pt;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p < total && j < P.Cout) v = *reinterpret_cast<const float4*>(P.dpq + p * K + (col < 64 ? j : P.Cout + j));
        *reinterpret_cast<float4*>(&ds[pt][col]) = v;
      }
    } else {
      for (int e = threadIdx.x; e < kEbPts * 128; e += kEbWThreads) {
        const int pt = e >> 7, col = e & 127;
        const int j = j0 + (col & 63);
        const int64_t p = p0 + pt;
        float v = 0.f;
        if (p < total && j < P.Cout) v = P.dpq[p * K + (col < 64 ? j : P.Cout + j)];
        ds[pt][col] = v;
      }
    }
    // x tile: consecutive threads walk consecutive points of one channel row (x is channel-major)
    if (xvec) {
      const int pt = (threadIdx.x & 15) * 4;                     // four consecutive points: one sample (N % 4 == 0)
      const int64_t p = p0 + pt;
      const int64_t b = p / P.N, n = p - b * P.N;
      const float* xp = P.x + b * P.sb + n;
      for (int cc = threadIdx.x >> 4; cc < 64; cc += kEbWThreads / 16) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p < total && c0 + cc < P.C) v = *reinterpret_cast<const float4*>(xp + (c0 + cc) * P.sc);
        *reinterpret_cast<float4*>(&xs[cc][pt]) = v;
      }
    } else {
      const int pt = threadIdx.x & 63;
      const int64_t p = p0 + pt;
      const int64_t b = p / P.N, n = p - b * P.N;
      const float* xp = P.x + b * P.sb + n * P.sn;
      for (int cc = threadIdx.x >> 6; cc < 64; cc += kEbWThreads / 64) {
        xs[cc][pt] = (p < total && c0 + cc < P.C) ? xp[(c0 + cc) * P.sc] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < kEbPts / 8; ++s) {
      const int k = 32 * half + 4 * s + lk;
      const float ap = ds[k][rt * 16 + li];
      const float aq = ds[k][64 + rt * 16 + li];
      bsum += ap;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float bv = xs[t * 16 + li][k];
        accp[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap, bv, accp[t], 0, 0, 0);
        accq[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq, bv, accq[t], 0, 0, 0);
      }
    }
  }
  // the second half of the points joins the first through LDS (fixed order: first half + second half)
  __syncthreads();
  float* scratch = &ds[0][0];                                    // 4 waves x 33 x 64 floats
  if (half == 1) {
    float* sw = scratch + rt * 33 * kWave + lane;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sw[(t * 4 + r) * kWave] = accp[t][r];
        sw[(16 + t * 4 + r) * kWave] = accq[t][r];
      }
    }
    sw[32 * kWave] = bsum;
  }
  __syncthreads();
  if (half == 1) return;
  {
    const float* sw = scratch + rt * 33 * kWave + lane;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        accp[t][r] += sw[(t * 4 + r) * kWave];
        accq[t][r] += sw[(16 + t * 4 + r) * kWave];
      }
    }
    bsum += sw[32 * kWave];
  }
  // partial in the weight's layout: W[j][c] <- dP^T x, W[j][C + c] <- dQ^T x - dP^T x
  float* part = P.part + static_cast<int64_t>(blockIdx.x) * (static_cast<int64_t>(P.Cout) * 2 * P.C + P.Cout);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int c = c0 + t * 16 + li;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = j0 + rt * 16 + lk * 4 + r;
      if (j < P.Cout && c < P.C) {
        part[static_cast<int64_t>(j) * 2 * P.C + c] = accp[t][r];
        part[static_cast<int64_t>(j) * 2 * P.C + P.C + c] = accq[t][r] - accp[t][r];
      }
    }
  }
  if (cg == 0) {
    bsum += __shfl_xor(bsum, 16);
    bsum += __shfl_xor(bsum, 32);
    const int j = j0 + rt * 16 + li;
    if (lk == 0 && j < P.Cout) part[static_cast<int64_t>(P.Cout) * 2 * P.C + j] = bsum;
  }
}

int eb_weight_grid(int64_t points) {
  const int64_t chunks = (points + kEbPts - 1) / kEbPts;
  return static_cast<int>(chunks < 256 ? (chunks > 0 ? chunks : 1) : 256);
}

}  // namespace
}  // namespace dgcn

using namespace dgcn;

extern "C" int dgcn_edgeconv_bwd_input_f32(const float* dpq, const float* conv_w, const float* g, int64_t gsb,
                                           int64_t gsc, int64_t gsn, float res_scale, int32_t B, int32_t C, int32_t N,
                                           int32_t Cout, float* dx, void* stream) {
  if (!dpq || !conv_w || !dx) return DGCN_E_NULL;
  if (B < 0 || C <= 0 || N <= 0 || Cout <= 0) return DGCN_E_SHAPE;
  if (B == 0) return DGCN_OK;
  EbInputParams P{dpq, conv_w, g, gsb, gsc, gsn, res_scale, dx, B, C, N, Cout};
  const int64_t tiles = static_cast<int64_t>(B) * ((N + 15) / 16);
  const int64_t groups = (tiles + kWavesPerWg - 1) / kWavesPerWg;
  const int grid = static_cast<int>(groups < 2 * num_cus() ? groups : 2 * num_cus());     // persistent: see the kernel
  hipLaunchKernelGGL(edgeconv_bwd_input_kernel, dim3(grid), dim3(kWgThreads), 0, static_cast<hipStream_t>(stream), P);
  return launch_status();
}

extern "C" int32_t dgcn_edgeconv_bwd_weight_num_partials(int32_t B, int32_t N) {
  if (B <= 0 || N <= 0) return 0;
  return eb_weight_grid(static_cast<int64_t>(B) * N);
}

extern "C" int dgcn_edgeconv_bwd_weight_f32(const float* dpq, const float* x, int64_t sb, int64_t sc, int64_t sn,
                                            int32_t B, int32_t C, int32_t N, int32_t Cout, float* partials,
                                            void* stream) {
  if (!dpq || !x || !partials) return DGCN_E_NULL;
  if (B <= 0 || C <= 0 || N <= 0 || Cout <= 0) return DGCN_E_SHAPE;
  const int64_t points = static_cast<int64_t>(B) * N;
  EbWeightParams P{dpq, x, sb, sc, sn, partials, B, C, N, Cout, static_cast<int>((points + kEbPts - 1) / kEbPts)};
  const int blocks = ((Cout + 63) / 64) * ((C + 63) / 64);
  hipLaunchKernelGGL(edgeconv_bwd_weight_kernel, dim3(eb_weight_grid(points), blocks), dim3(kEbWThreads), 0,
                     static_cast<hipStream_t>(stream), P);
  return launch_status();
}
