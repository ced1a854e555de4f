// Weight gradients of the row-wise Linear layers for gfx950:  dW[c][k] = sum_r G[r][c] * X[r][k]   (G^T X, r = 10^4 .. 10^6
// rows, C <= 128, K <= 256) on the bf16 matrix pipe, fp32-faithful (six-product split, bf16x6.h).
//
// Replaces the backward GEMMs with a 10^5 .. 10^6-long reduction that the stock library runs at the memory system's
// pace or worse:
//   grad of edge_encoder.weight  (dz^T F, E x C x hidden)   gcn_lib/sparse/torch_vertex.py:63-66 under
//                                                            eff_gcn_modules/rev/gcn_revop.py:121-133
//   grad of the MLP Linear weights (g^T a, N x C x K)        gcn_lib/sparse/torch_nn.py:50-71
// (round 2: a batched split-K product + partial sum + ragged-tail GEMM: 324 us at E = 791 k x 112 x 224, 83 us at
// N = 169 k x 128 x 128.)
//
// The reduction index is the ROW, so both MFMA operands need 8 consecutive rows of one column per lane -- the transpose
// of how the matrices lie in memory.  A persistent 768-thread workgroup streams its slab of rows 32 at a time with the
// two halves of the work in DIFFERENT waves (one wave doing both ran them one after the other: a wave that waits for
// room in the memory queue to issue its loads cannot issue its MFMAs either -- measured 0.41 ms of loads + 0.25 ms of
// MFMAs = 0.65 ms at 2.4 M x 128 x 128):
//   * waves 8..11 load: a thread fetches 8 consecutive rows of ONE column (for a fixed row the lanes read consecutive
//     columns: coalesced), of 2 column tiles of G and 2 NTW of X, two steps ahead; splits the 8 floats exactly into three
//     bf16 fragments ONCE and parks them in LDS in fragment order (unit = (column, 8-row group), 16 bytes; a wave's 64
//     units are one contiguous kilobyte in the order the MFMA lanes read them back: no bank conflicts either way);
//   * waves 0..7 multiply: wave w owns (all channel tiles) x (column tiles NTW w .. NTW w + NTW - 1): per step it reads
//     its B fragments once and walks the channel tiles, six MFMAs per tile pair, accumulators alternate;
//   * two LDS buffers, one barrier per step: the loaders fill buffer b ^ 1 while the multipliers read buffer b;
//   * each workgroup leaves one (C, K) partial; tn_reduce_kernel sums them in a fixed order: deterministic.
// Bound: HBM (G and X are read exactly once: 1.06 GB at the RevGCN shape).  Measured (round 3, 2.4 M x 128 x 128): the loads
// alone stream at 5.8 TB/s (0.43 ms), the MFMAs + fragment reads alone take 0.27 ms, split + LDS stores 0.12 ms; together
// 0.58 - 0.66 ms -- split and stores add to the load time instead of hiding under it and the per-step barrier makes
// every step as slow as the slower side; a third staging set (three steps ahead) changed nothing.  Next: three LDS
// buffers with full / empty counters instead of the barrier (the K <= 128 shapes have the LDS for it).

#include "bf16x6.h"

namespace dgcn {
namespace {

constexpr int kTnMmaWaves = 8;
constexpr int kTnLoadWaves = 4;
constexpr int kTnThreads = (kTnMmaWaves + kTnLoadWaves) * kWave;
constexpr int kTnStep = 32;            // rows per step = the k extent of v_mfma_f32_16x16x32_bf16

struct TnParams {
  const float* g;
  int64_t ldg;
  const float* x;
  int64_t ldx;
  int64_t rows;
  int C, K;
  float* part;        // [gridDim.x][part_stride]: the (C, K) partial, then (colsum_of != 0) the column sums of G or X
  int64_t part_stride;
  int colsum_of;      // 0: none; 1: column sums of G (C values); 2: of X (K values) -- the bias gradient of the Linear
  int64_t steps_per_wg;
};

struct TnRange {
  int64_t s_begin, s_end, nfull;
  bool has_tail;
};

// Position of the fragment unit (8-row group kq, column n) inside its tile's 64 units.  The multipliers read a tile as one
// contiguous kilobyte whatever the order inside it; the order is chosen for the loaders, whose lanes hold FOUR adjacent
// columns each: columns 4 apart go to adjacent units and odd tiles swap the halves of every 8-unit run, so that the
// eight lanes of a ds_write_b128 group (two tiles x four columns) cover all banks.
__device__ __forceinline__ int tn_unit(int tile_parity, int kq, int n) {
  return kq * 16 + ((((n & 3) << 2) | (n >> 2)) ^ (tile_parity << 2));
}

// ---------------------------------------------------------------------------------------------------- loading waves
// A wave-task is 32 rows x 64 columns of one matrix: lane (q, gj) loads rows 8 gj .. 8 gj + 7 of columns 4 q .. 4 q + 3
// with eight 16-byte loads (for a fixed row the 16 lanes of a row group read 256 contiguous bytes) and then owns four
// whole fragment units -- 8 consecutive rows of one column each -- without any exchange between lanes.  (One column per
// lane and dword loads needed 4x the load instructions: a wave can have 63 outstanding, the chip then starved.)
// G has 2 column blocks, X 2 NTW: wave pw takes the tasks pw, pw + 4.  Loads use the scalar base of the step's first
// row (+ the row inside the group) + one per-thread byte offset that never changes (no predicate: a predicated global load is a branch around it).
// Column groups past C / K inside a live block are clamped to the last one: they produce duplicates in units whose
// results the epilogue does not store; blocks past C / K are skipped.
template <int MT, int NTW>
__device__ __forceinline__ void tn_load_waves(const TnParams& P, const TnRange& R, i4v* Ab, i4v* Bb, int pw, int lane) {
  constexpr int AU = 8 * 64, BU = kTnMmaWaves * NTW * 64;
  constexpr int GB = 2, XB = 2 * NTW, TI = (GB + XB + kTnLoadWaves - 1) / kTnLoadWaves;
  const int q = lane & 15, gj = lane >> 4;
  const float* base[TI];
  int64_t ld[TI];
  bool live[TI];
  uint32_t off0[TI];           // byte offset of (row 8 gj, column 64 b + 4 q) from the step's first row
  i4v* region[TI];             // A or B planes
  int unit0[TI];               // the thread's unit of column 4 q (j = 0) inside a plane
  int pstride[TI];             // units between planes
  // column sums (P.colsum_of): the thread already holds 8 rows x 4 columns of its task per step -- the sums cost 8 adds
  // per column and step here instead of a pass of their own over the matrix
  bool want[TI];
  int ngroups[TI], col0[TI];
  f4v cs[TI];
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    const int t = pw + kTnLoadWaves * i;
    const bool is_g = t < GB;
    const int b = is_g ? t : t - GB;
    const int ncol = is_g ? P.C : P.K;
    live[i] = t < GB + XB && 64 * b < ncol && (!is_g || 4 * b < MT);
    base[i] = is_g ? P.g : P.x;
    ld[i] = is_g ? P.ldg : P.ldx;
    const int groups = (ncol - 64 * b) / 4;                    // 4-column groups of the block that exist
    const int qc = q < groups ? q : groups - 1;
    off0[i] = static_cast<uint32_t>((8 * gj * ld[i] + 64 * b + 4 * qc) * 4);
    const int tile = 4 * b + (q >> 2);
    pstride[i] = is_g ? AU : BU;
    region[i] = is_g ? Ab : Bb;
    unit0[i] = tile * 64 + tn_unit(tile & 1, gj, 4 * (q & 3));
    want[i] = live[i] && P.colsum_of == (is_g ? 1 : 2);
    ngroups[i] = groups;
    col0[i] = 64 * b;
    cs[i] = f4v{0.f, 0.f, 0.f, 0.f};
  }
  struct Stage { f4v v[8]; };
  auto fetch = [&](int i, int64_t step, Stage& S) {
    if (live[i]) {
      const char* sb = reinterpret_cast<const char*>(base[i] + step * kTnStep * ld[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) S.v[e] = *reinterpret_cast<const f4v*>(sb + e * ld[i] * 4 + off0[i]);   // scalar + lane
    }
  };
  auto fetch_tail = [&](int i, Stage& S) {          // the rows % 32 left over: zero-filled
    const int left = static_cast<int>(P.rows - R.nfull * kTnStep);      // 1 .. 31
    const char* sb = reinterpret_cast<const char*>(base[i] + R.nfull * kTnStep * ld[i]);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      S.v[e] = (live[i] && 8 * gj + e < left) ? *reinterpret_cast<const f4v*>(sb + e * ld[i] * 4 + off0[i])
                                              : f4v{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto park = [&](int i, int buf, const Stage& S) {
    if (live[i]) {
      if (want[i]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[i] = cs[i] + S.v[e];
      }
      i4v* u = region[i] + buf * 3 * pstride[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        i4v h, m, l;
        eg_split3(f4v{S.v[0][j], S.v[1][j], S.v[2][j], S.v[3][j]}, f4v{S.v[4][j], S.v[5][j], S.v[6][j], S.v[7][j]}, h, m, l);
        // tn_unit(parity, gj, 4 (q & 3) + j) = tn_unit(parity, gj, 4 (q & 3)) ^ (4 j): column j of the lane
        i4v* w = u + (unit0[i] ^ (4 * j));
        w[0] = h; w[pstride[i]] = m; w[2 * pstride[i]] = l;
        __builtin_amdgcn_sched_barrier(0);       // one column at a time: four interleaved splits need 100+ registers
      }
    }
  };
  // Step s is multiplied out of buffer (s - s_begin) & 1.  The wave's first task is staged two steps ahead (SA holds
  // step s + 1 on entry of an even step, SB on an odd one); its second task (K > 128 only: 6 tasks on 4 waves) one step
  // ahead in a single set of registers, re-requested as soon as it is parked -- three stages of 32 registers fit the
  // 168 a wave may have at 12 waves per CU, four do not.
  Stage SA, SB, S1;
  if (R.s_begin < R.s_end) {
    fetch(0, R.s_begin, SA);
    if constexpr (TI > 1) fetch(1, R.s_begin, S1);
    park(0, 0, SA);
    if (R.s_begin + 1 < R.s_end) fetch(0, R.s_begin + 1, SA);
    if constexpr (TI > 1) {
      park(1, 0, S1);
      if (R.s_begin + 1 < R.s_end) fetch(1, R.s_begin + 1, S1);
    }
    __syncthreads();
    for (int64_t s = R.s_begin; s < R.s_end; s += 2) {
      if (s + 2 < R.s_end) fetch(0, s + 2, SB);
      if (s + 1 < R.s_end) {
        park(0, 1, SA);
        if constexpr (TI > 1) {
          park(1, 1, S1);
          if (s + 2 < R.s_end) fetch(1, s + 2, S1);
        }
      }
      __syncthreads();
      if (s + 1 >= R.s_end) break;
      if (s + 3 < R.s_end) fetch(0, s + 3, SA);
      if (s + 2 < R.s_end) {
        park(0, 0, SB);
        if constexpr (TI > 1) {
          park(1, 0, S1);
          if (s + 3 < R.s_end) fetch(1, s + 3, S1);
        }
      }
      __syncthreads();
    }
  }
  if (R.has_tail) {
    fetch_tail(0, SA);
    park(0, 0, SA);
    if constexpr (TI > 1) {
      fetch_tail(1, S1);
      park(1, 0, S1);
    }
    __syncthreads();
  }
  if (P.colsum_of) {
    // the four row groups of the wave meet by shuffle (fixed order), row group 0 writes its existing column groups
    float* cdst = P.part + static_cast<int64_t>(blockIdx.x) * P.part_stride + static_cast<int64_t>(P.C) * P.K;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = cs[i][j];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        cs[i][j] = v;
      }
      if (want[i] && gj == 0 && q < ngroups[i]) *reinterpret_cast<f4v*>(cdst + col0[i] + 4 * q) = cs[i];
    }
  }
}

// ------------------------------------------------------------------------------------------------ multiplying waves
template <int MT, int NTW>
__device__ __forceinline__ void tn_mma_waves(const TnParams& P, const TnRange& R, const i4v* Ab, const i4v* Bb, int wave,
                                           int lane) {
  constexpr int AT = 8, BT = kTnMmaWaves * NTW;
  constexpr int AU = AT * 64, BU = BT * 64;
  const int C = P.C, K = P.K;
  const int n = lane & 15, kq = lane >> 4;
  const int lu[2] = {tn_unit(0, kq, n), tn_unit(1, kq, n)};      // the lane's unit in an even / odd tile
  f4v acc[MT][NTW];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = f4v{0.f, 0.f, 0.f, 0.f};
  }
  auto compute = [&](int buf) {
    const i4v* A = Ab + buf * 3 * AU;                         // + plane * AU + mt * 64 + unit of the lane
    const i4v* B = Bb + buf * 3 * BU + wave * NTW * 64;      // + plane * BU + nt * 64 + unit of the lane
    i4v bf[NTW][3];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
#pragma unroll
      for (int p = 0; p < 3; ++p) bf[nt][p] = B[p * BU + nt * 64 + lu[(wave * NTW + nt) & 1]];
    }
    constexpr int pa[6] = {0, 0, 1, 0, 2, 1};     // a1 b1, a1 b2, a2 b1, a1 b3, a3 b1, a2 b2
    constexpr int pb[6] = {0, 1, 0, 2, 0, 1};
    if constexpr (NTW >= 2) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        i4v af[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) af[p] = A[p * AU + mt * 64 + lu[mt & 1]];
#pragma unroll
        for (int s6 = 0; s6 < 6; ++s6) {
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = eg_mfma_bf16(af[pa[s6]], bf[nt][pb[s6]], acc[mt][nt]);
        }
      }
    } else {
      // one column tile per wave: pair the channel tiles so that consecutive MFMAs use different accumulators
#pragma unroll
      for (int mt = 0; mt < MT; mt += 2) {
        i4v af[2][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          af[0][p] = A[p * AU + mt * 64 + lu[mt & 1]];
          af[1][p] = (mt + 1 < MT) ? A[p * AU + (mt + 1) * 64 + lu[(mt + 1) & 1]] : af[0][p];
        }
#pragma unroll
        for (int s6 = 0; s6 < 6; ++s6) {
          acc[mt][0] = eg_mfma_bf16(af[0][pa[s6]], bf[0][pb[s6]], acc[mt][0]);
          if (mt + 1 < MT) acc[mt + 1][0] = eg_mfma_bf16(af[1][pa[s6]], bf[0][pb[s6]], acc[mt + 1][0]);
        }
      }
    }
  };
  const bool wave_live = wave * NTW * 16 < K;       // (a dead wave would read LDS nobody wrote)
  if (R.s_begin < R.s_end) {
    __syncthreads();
    for (int64_t s = R.s_begin; s < R.s_end; s += 2) {
      if (wave_live) compute(0);
      __syncthreads();
      if (s + 1 >= R.s_end) break;
      if (wave_live) compute(1);
      __syncthreads();
    }
  }
  if (R.has_tail) {
    __syncthreads();
    if (wave_live) compute(0);
  }
  // ---- this workgroup's partial: D layout, lane (n, q) holds channels 16 mt + 4 q + j, column 16 nt + n ----
  float* part = P.part + static_cast<int64_t>(blockIdx.x) * P.part_stride;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const int k = (wave * NTW + nt) * 16 + n;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = mt * 16 + 4 * kq + j;
        if (c < C && k < K) part[static_cast<int64_t>(c) * K + k] = acc[mt][nt][j];
      }
    }
  }
}

// MT = channel tiles the multipliers walk (C <= 16 MT), NTW = column tiles per multiplying wave (K <= 128 NTW)
template <int MT, int NTW>
__global__ __launch_bounds__(kTnThreads) void rows_tn_kernel(const TnParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  i4v* Ab = reinterpret_cast<i4v*>(smem_raw);      // [2][3][8 tiles * 64 units of 16 bytes]
  i4v* Bb = Ab + 2 * 3 * 8 * 64;                   // [2][3][8 NTW tiles * 64]
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));   // uniform: scalar registers
  TnRange R;
  R.s_begin = static_cast<int64_t>(blockIdx.x) * P.steps_per_wg;
  R.nfull = P.rows / kTnStep;                      // whole steps; the last workgroup adds the rows left over
  R.s_end = R.s_begin + P.steps_per_wg;
  if (R.s_end > R.nfull) R.s_end = R.nfull;
  R.has_tail = blockIdx.x == gridDim.x - 1 && R.nfull * kTnStep < P.rows;
  if (wave >= kTnMmaWaves) {
    tn_load_waves<MT, NTW>(P, R, Ab, Bb, wave - kTnMmaWaves, lane);
  } else {
    tn_mma_waves<MT, NTW>(P, R, Ab, Bb, wave, lane);
  }
}

// dst[c][k] = sum_s part[s][c][k], s ascending within 16 interleaved groups, the groups in a fixed tree: deterministic.
// 16 lanes x 16 groups per block so that a few hundred partials of a 16 k-element matrix still fill the chip (one thread
// per output walking 256 partials one after the other took 62 us).  Elements past the matrix (mat_total .. total) are the
// column sums the loaders left behind the partial: they go to `tail`.  `transposed`: dst[k][c] (row stride ldo) -- the
// caller asked for (X^T G)^T.
template <typename V>
__global__ __launch_bounds__(256) void tn_reduce_kernel(const V* __restrict__ parts, int nparts, int64_t total,
                                                         int64_t mat_total, int64_t row_len, float* __restrict__ dst,
                                                         int64_t ldo, int transposed, float* __restrict__ tail) {
  __shared__ V red[16][16];
  const int o = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * 16 + o;          // output index in units of V
  V acc{};
  if (idx < total) {
    V v[4];
    int s = grp;
    for (; s + 48 < nparts; s += 64) {
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = parts[static_cast<int64_t>(s + 16 * u) * total + idx];
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = acc + v[u];
    }
    for (; s < nparts; s += 16) acc = acc + parts[static_cast<int64_t>(s) * total + idx];
  }
  red[grp][o] = acc;
  __syncthreads();
  if (grp == 0 && idx < total) {
    V r = red[0][o];
#pragma unroll
    for (int q = 1; q < 16; ++q) r = r + red[q][o];
    constexpr int W = sizeof(V) / 4;
    const int64_t e = idx * W;                       // element index in the (C, K) matrix, row_len = K
    if (e >= mat_total) {
      *reinterpret_cast<V*>(tail + (e - mat_total)) = r;
    } else if (!transposed) {
      *reinterpret_cast<V*>(dst + (e / row_len) * ldo + (e % row_len)) = r;
    } else {
      const float* rf = reinterpret_cast<const float*>(&r);
#pragma unroll
      for (int w = 0; w < W; ++w) dst[((e + w) % row_len) * ldo + (e + w) / row_len] = rf[w];
    }
  }
}

struct TnShape {
  int mt, ntw;
  size_t lds;
};

inline bool tn_shape(int C, int K, TnShape* S) {
  if (C < 4 || C > 128 || K < 4 || K > 256 || C % 4 != 0 || K % 4 != 0) return false;      // 16-byte loads
  S->mt = (C + 15) / 16;
  if (S->mt < 7) S->mt = (S->mt <= 4) ? 4 : 7;      // instantiated: 4, 7, 8 channel tiles
  S->ntw = K <= 128 ? 1 : 2;
  S->lds = static_cast<size_t>(2) * 3 * (8 * 64 + kTnMmaWaves * S->ntw * 64) * 16;
  return true;
}

inline int tn_grid(int64_t rows) {
  const int64_t steps = rows / kTnStep;               // whole steps; the last workgroup takes the rows % 32 too
  int64_t g = (steps + 3) / 4;                        // at least four steps per workgroup (13 k rows: 20.9 -> 15.7 us,
                                                      // its partial sum 4.7 -> 5.3 us; two steps: 14.9 / 6.3)
  if (g > num_cus()) g = num_cus();
  return static_cast<int>(g < 1 ? 1 : g);
}

template <int MT, int NTW>
int tn_launch(const TnParams& P, const TnShape& S, int grid, hipStream_t s) {
  const void* fn = reinterpret_cast<const void*>(rows_tn_kernel<MT, NTW>);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(S.lds));
  if (e != hipSuccess) return static_cast<int>(e);
  hipLaunchKernelGGL((rows_tn_kernel<MT, NTW>), dim3(grid), dim3(kTnThreads), S.lds, s, P);
  return launch_status();
}

}  // namespace
}  // namespace dgcn

using namespace dgcn;

extern "C" int32_t dgcn_rows_tn_supported(int32_t C, int32_t K) {
  TnShape S;
  return tn_shape(C, K, &S) ? 1 : 0;
}

extern "C" int32_t dgcn_rows_tn_num_partials(int64_t rows, int32_t C, int32_t K) {
  TnShape S;
  if (rows <= 0 || !tn_shape(C, K, &S)) return 0;
  return tn_grid(rows);
}

// out = G^T X (C, K) [or its transpose (K, C) with `transposed`], and with colsum_of = 1 / 2 the column sums of G (C) /
// of X (K) into `colsum` from the same pass over the rows (the bias gradient of the Linear whose weight gradient this is).
// `partials`: dgcn_rows_tn_num_partials(rows, C, K) x (C K + (colsum_of == 1 ? C : colsum_of == 2 ? K : 0)) floats.
extern "C" int dgcn_rows_tn_colsum_f32(const float* g, int64_t ldg, const float* x, int64_t ldx, int64_t rows, int32_t C,
                                       int32_t K, float* partials, float* out, int64_t ldo, int32_t transposed,
                                       int32_t colsum_of, float* colsum, void* stream) {
  if (!g || !x || !partials || !out) return DGCN_E_NULL;
  if (colsum_of < 0 || colsum_of > 2) return DGCN_E_MODE;
  if (colsum_of && !colsum) return DGCN_E_NULL;
  TnShape S;
  if (rows <= 0 || !tn_shape(C, K, &S) || ldg < C || ldx < K || ldo < (transposed ? C : K) || ldg % 4 != 0 || ldx % 4 != 0)
    return DGCN_E_SHAPE;
  if ((reinterpret_cast<uintptr_t>(g) & 15u) || (reinterpret_cast<uintptr_t>(x) & 15u)) return DGCN_E_ALIGN;
  if (colsum_of && ((reinterpret_cast<uintptr_t>(partials) & 15u) || (reinterpret_cast<uintptr_t>(colsum) & 15u)))
    return DGCN_E_ALIGN;
  // the loaders address a step with 32-bit byte offsets from its first row
  if (ldg * 4 * kTnStep > 0x7fffffffLL || ldx * 4 * kTnStep > 0x7fffffffLL) return DGCN_E_SHAPE;
  const int64_t mat_total = static_cast<int64_t>(C) * K;
  const int64_t total = mat_total + (colsum_of == 1 ? C : colsum_of == 2 ? K : 0);
  TnParams P;
  P.g = g; P.ldg = ldg; P.x = x; P.ldx = ldx; P.rows = rows; P.C = C; P.K = K; P.part = partials;
  P.part_stride = total; P.colsum_of = colsum_of;
  const int grid = tn_grid(rows);
  const int64_t steps = rows / kTnStep;
  P.steps_per_wg = (steps + grid - 1) / grid;
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc;
  if (S.ntw == 1) {
    switch (S.mt) {
      case 4: rc = tn_launch<4, 1>(P, S, grid, s); break;
      case 7: rc = tn_launch<7, 1>(P, S, grid, s); break;
      default: rc = tn_launch<8, 1>(P, S, grid, s); break;
    }
  } else {
    switch (S.mt) {
      case 4: rc = tn_launch<4, 2>(P, S, grid, s); break;
      case 7: rc = tn_launch<7, 2>(P, S, grid, s); break;
      default: rc = tn_launch<8, 2>(P, S, grid, s); break;
    }
  }
  if (rc != DGCN_OK) return rc;
  const bool wide = K % 4 == 0 && ldo % 4 == 0 && !(reinterpret_cast<uintptr_t>(partials) & 15u) &&
                    !(reinterpret_cast<uintptr_t>(out) & 15u) && (!colsum_of || !(reinterpret_cast<uintptr_t>(colsum) & 15u));
  if (wide) {
    const int64_t t4 = total / 4;
    hipLaunchKernelGGL(tn_reduce_kernel<f4v>, dim3(static_cast<unsigned>((t4 + 15) / 16)), dim3(256), 0, s,
                       reinterpret_cast<const f4v*>(partials), grid, t4, mat_total, static_cast<int64_t>(K), out, ldo,
                       transposed ? 1 : 0, colsum);
  } else {
    hipLaunchKernelGGL(tn_reduce_kernel<float>, dim3(static_cast<unsigned>((total + 15) / 16)), dim3(256), 0, s,
                       partials, grid, total, mat_total, static_cast<int64_t>(K), out, ldo, transposed ? 1 : 0, colsum);
  }
  return launch_status();
}

extern "C" int dgcn_rows_tn_f32(const float* g, int64_t ldg, const float* x, int64_t ldx, int64_t rows, int32_t C,
                                int32_t K, float* partials, float* out, int64_t ldo, void* stream) {
  return dgcn_rows_tn_colsum_f32(g, ldg, x, ldx, rows, C, K, partials, out, ldo, 0, 0, nullptr, stream);
}
