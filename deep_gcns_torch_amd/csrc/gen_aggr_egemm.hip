// GENConv with a per-layer edge encoder on WIDE edge features, forward, for gfx950 (MI355X).
//
// Reference call chain (one GENConv of RevGCN / DeeperGCN on ogbn-proteins, ogbg-ppa, ...):
//   edge_emb = self.edge_encoder(edge_attr)      Linear(edge_feat_dim -> C) over ALL E rows, edge_feat_dim = hidden
//                                                (gcn_lib/sparse/torch_vertex.py:56-66; eff_gcn_modules/rev/
//                                                 rev_layer.py:53-75; examples/ogb_eff/ogbn_proteins/model_rev.py:45-55)
//   m_e = relu(x[src] + edge_emb) + eps ; out_i = AGGR_{e -> i} m_e      (torch_vertex.py:68,78-85, torch_message.py:44-85)
// i.e. an E x K x C GEMM (39.7 GFLOP at E = 791 k, K = 224, C = 112) whose (E, C) result is written, re-read by the
// gather/scatter chain and thrown away.  Here the GEMM tile never leaves the chip:
//
//   * a work item = 64 consecutive CSR positions (edges sorted by destination), one wave per item, 16 edges per
//     batch.  Lane (m, kb) holds the feature row of edge m as the A operand of v_mfma_f32_16x16x4_f32 (exact fp32,
//     an fma chain) and streams it 128 bytes (one cache line per edge) at a time; the encoder weight lives in LDS
//     for the lifetime of the workgroup and is read with conflict-free ds_read_b128 (k indices permuted so that a
//     lane's four consecutive B operands are one 16-byte word); the accumulators start from x[src] + bias, so the
//     tile comes out as z = x_j + W f_e + b;
//   * the tile goes through a per-wave LDS buffer into the row-walk layout (lanes over channels) and is folded
//     edge by edge into the running aggregation state of the current destination row (online softmax / power sums
//     / first arg-max), exactly the state algebra of gen_aggr_fwd.hip;
//   * rows that straddle item boundaries leave partial states (at most two per item) that a tiny second kernel
//     merges in item order: deterministic, no atomics, any degree distribution is perfectly balanced because items
//     are cut by EDGE count, not by row;
//   * with z_save the pre-activation rows are written once (original edge order) for the backward, which then needs
//     neither the features nor another GEMM to rebuild them.
//
// Bound: fp32 MFMA (2*E*K*C flop at 157 TF) against E*K*4 bytes of features from HBM; both ~0.2 ms at the
// ogbn-proteins cluster shape.  x rows and the index arrays are L2-resident.

#include "gen_aggr_common.h"
#include "gen_aggr_state.h"

namespace dgcn {
namespace {

typedef float f4v __attribute__((ext_vector_type(4)));

constexpr int kEgM = 16;        // edges per MFMA batch (M of the 16x16x4 tile)
constexpr int kEgItem = 64;     // consecutive CSR positions per work item
constexpr int kEgChunk = 32;    // feature floats per k-chunk = one 128-byte line per edge
constexpr int kEgWPad = 8;      // LDS row stride of W = Kpad + 8 floats: (stride/4) % 4 == 2 makes the B-operand
                                // ds_read_b128 of the four fixed lane groups conflict-free
constexpr int kEgMaxWaves = 8;
constexpr int kEgLdsBytes = 160 * 1024;
constexpr int kEgInfo = 4;      // int32 per item: head_row, tail_row, head_continues, unused

struct EgParams {
  int n_rows, n_edges, n_items;
  const int32_t* rowptr;
  const int32_t* col;
  const int32_t* eperm;
  const int32_t* erow;      // [E] destination row of every CSR position
  const float* x;
  int64_t x_stride;
  const float* feat;
  int64_t feat_stride;
  const float* w;           // [C][K]
  const float* b;           // [C] or null
  int C, K, Kpad, ZS;
  int mode, msg, with_d;
  float t, p, eps;
  const float* t_dev;
  const float* p_dev;
  float* out;
  void* aux1;
  float* aux2;
  int32_t* range_flag;
  int add_root;
  float* z_save;            // [E][C] original edge order, or null
  float* part;              // [n_items][2][4][C]
  int32_t* info;            // [n_items][kEgInfo]
};

// ---- state -> result ------------------------------------------------------------------------------------------
// (same formulas as the epilogue of gen_aggr_fwd_kernel; the softmax sums arrive already corrected for eps)
template <int MODE, int VEC>
__device__ __forceinline__ void eg_finalize(const State<VEC>& st, float deg, float p, float (&res)[VEC],
                                            float (&x1)[VEC], float (&x2)[VEC], int (&xi)[VEC], bool& out_of_range) {
  out_of_range = false;
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    x1[j] = 0.f; x2[j] = 0.f; xi[j] = -1;
    if constexpr (MODE == DGCN_AGGR_SOFTMAX) {
      const bool any = st.b[j] > 0.f;
      const float inv = any ? 1.f / st.b[j] : 0.f;
      res[j] = st.c[j] * inv;
      x1[j] = any ? (st.a[j] + fast_log2(st.b[j])) * 0.6931471805599453f : 0.f;
      x2[j] = st.d[j] * inv;
      out_of_range = out_of_range || !(fabsf(x1[j]) < kShiftSafe);
    } else if constexpr (MODE == DGCN_AGGR_POWER) {
      const float q = st.b[j] / fmaxf(deg, 1.f);
      const float r = fminf(fmaxf(q, kPowLo), kPowHi);
      res[j] = fast_pow(r, 1.f / p);
      x1[j] = q;
      x2[j] = st.d[j];
    } else if constexpr (MODE == DGCN_AGGR_MAX) {
      res[j] = st.idx[j] >= 0 ? st.a[j] : 0.f;
      xi[j] = st.idx[j];
    } else if constexpr (MODE == DGCN_AGGR_MEAN) {
      res[j] = st.b[j] / fmaxf(deg, 1.f);
    } else {
      res[j] = st.b[j];
    }
  }
}

template <int MODE, int VEC>
__device__ __forceinline__ void eg_write_row(const EgParams& P, int row, int c0, const State<VEC>& st, float deg,
                                             float p) {
  float res[VEC], x1[VEC], x2[VEC];
  int xi[VEC];
  bool oor;
  eg_finalize<MODE, VEC>(st, deg, p, res, x1, x2, xi, oor);
  const int64_t o = static_cast<int64_t>(row) * P.C + c0;
  if (P.add_root) {
    float xr[VEC];
    load_vec<VEC>(xr, P.x + static_cast<int64_t>(row) * P.x_stride + c0);
#pragma unroll
    for (int j = 0; j < VEC; ++j) res[j] += xr[j];
  }
  store_vec<VEC>(P.out + o, res);
  if constexpr (MODE == DGCN_AGGR_MAX) {
    if (P.aux1) store_vec_i<VEC>(static_cast<int32_t*>(P.aux1) + o, xi);
  } else if constexpr (MODE == DGCN_AGGR_SOFTMAX || MODE == DGCN_AGGR_POWER) {
    if (P.aux1) store_vec<VEC>(static_cast<float*>(P.aux1) + o, x1);
    if (P.aux2) store_vec<VEC>(P.aux2 + o, x2);
    if constexpr (MODE == DGCN_AGGR_SOFTMAX) {
      if (P.range_flag && oor) atomicOr(P.range_flag, 1);   // rare
    }
  }
}

// Softmax sums are kept over r = relu(z) inside the fold; back to sums over m = r + eps before a state leaves
// the registers:  sum e m = A + eps D,  sum e m^2 = A2 + 2 eps A + eps^2 D.
template <int MODE, int VEC>
__device__ __forceinline__ void eg_fix_eps(State<VEC>& st, float eps_r, bool with_d) {
  if constexpr (MODE == DGCN_AGGR_SOFTMAX) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      if (with_d) st.d[j] = fmaf(eps_r, fmaf(eps_r, st.b[j], 2.f * st.c[j]), st.d[j]);
      st.c[j] = fmaf(eps_r, st.b[j], st.c[j]);
    }
  }
}

template <int MODE, int VEC>
__device__ __forceinline__ void eg_store_partial(const EgParams& P, int item, int which, int c0,
                                                 const State<VEC>& st) {
  float* ws = P.part + (static_cast<int64_t>(item) * 2 + which) * 4 * P.C + c0;
  if constexpr (MODE == DGCN_AGGR_MAX) {
    float fi[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) fi[j] = __int_as_float(st.idx[j]);
    store_vec<VEC>(ws, st.a);
    store_vec<VEC>(ws + P.C, fi);
  } else {
    store_vec<VEC>(ws, st.a);
    store_vec<VEC>(ws + P.C, st.b);
    store_vec<VEC>(ws + 2 * P.C, st.c);
    store_vec<VEC>(ws + 3 * P.C, st.d);
  }
}

// Per-wave walk bookkeeping (all wave-uniform).
struct EgWalk {
  int cur_row;     // destination row whose state is in the registers
  int cnt;         // edges folded into it inside this item
  int head;        // 1: the row started before this item (its state here is a partial)
  int head_row;    // what goes into info[]: row of the head partial or -1
  int head_cont;   // the head partial's row also continues past this item
  int tail_row;    // row of the tail partial or -1
};

// The current row is complete inside this item or is the item's head partial: flush it.
template <int MODE>
__device__ __forceinline__ void eg_flush(const EgParams& P, EgWalk& wk, int item, State<4>& st, int c0, bool act,
                                         float eps_r, float p) {
  eg_fix_eps<MODE, 4>(st, eps_r, P.with_d != 0);
  if (wk.head) {
    if (act) eg_store_partial<MODE, 4>(P, item, 0, c0, st);
    wk.head_row = wk.cur_row;
    wk.head = 0;
  } else if (act) {
    eg_write_row<MODE, 4>(P, wk.cur_row, c0, st, static_cast<float>(wk.cnt), p);
  }
}

template <int MODE>
__device__ __forceinline__ void eg_empty_rows(const EgParams& P, int r0, int r1, int c0, bool act, float p) {
  State<4> e;
  state_init<MODE, 4>(e);
  for (int r = r0; r < r1; ++r) {
    if (act) eg_write_row<MODE, 4>(P, r, c0, e, 0.f, p);
  }
}

// Fold the nb edges of one batch (rows of the LDS tile zt) into the running state.
template <int MODE, bool RELU, bool WITH_D>
__device__ __forceinline__ void eg_walk_batch(const EgParams& P, EgWalk& wk, int item, State<4>& st,
                                              const float* __restrict__ zt, int nb, int rowv, int eidv, int cl,
                                              bool act, float eps, float eps_r, float t2, float c0s, float p) {
  const int c0 = cl * 4;
#pragma unroll 2
  for (int e = 0; e < kEgM; ++e) {
    if (e < nb) {   // wave-uniform
      const int row = __builtin_amdgcn_readlane(rowv, e);
      const int eid = __builtin_amdgcn_readlane(eidv, e);
      if (row != wk.cur_row) {
        eg_flush<MODE>(P, wk, item, st, c0, act, eps_r, p);
        eg_empty_rows<MODE>(P, wk.cur_row + 1, row, c0, act, p);
        state_init<MODE, 4>(st);
        wk.cur_row = row;
        wk.cnt = 0;
      }
      float v[1][4];
      bool ok[1] = {true};
      int eids[1] = {eid};
#pragma unroll
      for (int j = 0; j < 4; ++j) v[0][j] = 0.f;
      if (act) {
        const f4v zz = *reinterpret_cast<const f4v*>(zt + e * P.ZS + c0);
        v[0][0] = zz.x; v[0][1] = zz.y; v[0][2] = zz.z; v[0][3] = zz.w;
        if (P.z_save) {
          *reinterpret_cast<f4v*>(P.z_save + static_cast<int64_t>(eid) * P.C + c0) = zz;
        }
      }
      accumulate<MODE, 4, 1, RELU, WITH_D, true>(st, v, ok, eids, eps, t2, c0s, p);
      wk.cnt += 1;
    }
  }
}

template <int MODE>
__device__ __forceinline__ void eg_walk_dispatch(const EgParams& P, EgWalk& wk, int item, State<4>& st,
                                                 const float* __restrict__ zt, int nb, int rowv, int eidv, int cl,
                                                 bool act, float eps, float eps_r, float t2, float c0s, float p) {
  const bool relu = P.msg == DGCN_MSG_RELU_EPS;
  constexpr bool CAN_D = MODE == DGCN_AGGR_SOFTMAX || MODE == DGCN_AGGR_POWER;
  if constexpr (CAN_D) {
    if (P.with_d) {
      if (relu) eg_walk_batch<MODE, true, true>(P, wk, item, st, zt, nb, rowv, eidv, cl, act, eps, eps_r, t2, c0s, p);
      else eg_walk_batch<MODE, false, true>(P, wk, item, st, zt, nb, rowv, eidv, cl, act, eps, eps_r, t2, c0s, p);
      return;
    }
  }
  if (relu) eg_walk_batch<MODE, true, false>(P, wk, item, st, zt, nb, rowv, eidv, cl, act, eps, eps_r, t2, c0s, p);
  else eg_walk_batch<MODE, false, false>(P, wk, item, st, zt, nb, rowv, eidv, cl, act, eps, eps_r, t2, c0s, p);
}

// End of an item: the row in the registers either ends here (complete, or a head partial that ends) or continues
// into the next item (tail partial; a row covering the whole item stays a head partial that "continues").
template <int MODE>
__device__ __forceinline__ void eg_finish_item(const EgParams& P, EgWalk& wk, int item, int ie, State<4>& st, int c0,
                                               bool act, float eps_r, float p) {
  const int next_row = (ie < P.n_edges) ? uni(P.erow[ie]) : P.n_rows;   // n_rows: "no more edges"
  const bool continues = next_row == wk.cur_row;
  if (continues && !wk.head) {
    eg_fix_eps<MODE, 4>(st, eps_r, P.with_d != 0);
    if (act) eg_store_partial<MODE, 4>(P, item, 1, c0, st);
    wk.tail_row = wk.cur_row;
  } else {
    wk.head_cont = (continues && wk.head) ? 1 : 0;
    eg_flush<MODE>(P, wk, item, st, c0, act, eps_r, p);
  }
  if (ie >= P.n_edges) eg_empty_rows<MODE>(P, wk.cur_row + 1, P.n_rows, c0, act, p);   // trailing empty rows
}

// KC > 0: the feature width is known at compile time (KC 32-float chunks): a lane keeps its edge's whole feature
// row in registers, all of its loads are in flight at once and the MFMA chain of a batch runs without waiting on
// memory.  KC == 0: any width, chunk loads software-pipelined one chunk ahead.
template <int NT, int KC>
__global__ __launch_bounds__(kEgMaxWaves * kWave) void egemm_fwd_kernel(const EgParams P) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int C = P.C, K = P.K, Kpad = P.Kpad;
  const int WS = Kpad + kEgWPad;
  const int ZS = P.ZS;
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int nwaves = blockDim.x >> 6;
  float* Wl = smem;                                        // [NT*16][WS], rows >= C and columns >= K are zero
  float* zt = smem + NT * 16 * WS + wave * (kEgM * ZS);    // this wave's z tile [16][ZS]

  {
    const int q = Kpad / 4;
    for (int idx = threadIdx.x; idx < NT * 16 * q; idx += blockDim.x) {
      const int r = idx / q, c4 = (idx - r * q) * 4;
      f4v v = {0.f, 0.f, 0.f, 0.f};
      if (r < C && c4 < K) v = *reinterpret_cast<const f4v*>(P.w + static_cast<int64_t>(r) * K + c4);
      *reinterpret_cast<f4v*>(Wl + r * WS + c4) = v;
    }
  }
  __syncthreads();

  const int n = lane & 15;        // column of the MFMA tile: channel (B / D operand), edge (A operand)
  const int kb = lane >> 4;       // k slot of the A / B operand, row block of D
  const int cl = lane;            // walk layout: lane -> channels 4*cl .. 4*cl+3
  const bool act = cl * 4 < C;
  const float t = P.t_dev ? *P.t_dev : P.t;
  const float p = P.p_dev ? *P.p_dev : P.p;
  const float eps = P.eps;
  const float eps_r = (P.msg == DGCN_MSG_RELU_EPS) ? eps : 0.f;
  const float t2 = t * 1.4426950408889634f;
  const float c0s = t2 * eps_r;
  const int E = P.n_edges;
  const uint32_t xs32 = static_cast<uint32_t>(P.x_stride);
  const int nchunks = Kpad / kEgChunk;

  float bias[NT];
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) bias[ct] = (P.b && ct * 16 + n < C) ? P.b[ct * 16 + n] : 0.f;

  for (int item = blockIdx.x * nwaves + wave; item < P.n_items; item += gridDim.x * nwaves) {
    const int ib = item * kEgItem;
    const int ie = min(ib + kEgItem, E);
    EgWalk wk;
    wk.head_row = -1; wk.tail_row = -1; wk.head_cont = 0; wk.cnt = 0;
    {
      const int prev_row = ib > 0 ? uni(P.erow[ib - 1]) : -1;
      const int first_row = uni(P.erow[ib]);
      wk.head = (prev_row == first_row) ? 1 : 0;
      wk.cur_row = first_row;
      // rows without edges between the previous item's last row and this item's first row belong to this item
      if (!wk.head) {
        switch (P.mode) {
          case DGCN_AGGR_ADD: eg_empty_rows<DGCN_AGGR_ADD>(P, prev_row + 1, first_row, cl * 4, act, p); break;
          case DGCN_AGGR_MEAN: eg_empty_rows<DGCN_AGGR_MEAN>(P, prev_row + 1, first_row, cl * 4, act, p); break;
          case DGCN_AGGR_MAX: eg_empty_rows<DGCN_AGGR_MAX>(P, prev_row + 1, first_row, cl * 4, act, p); break;
          case DGCN_AGGR_SOFTMAX: eg_empty_rows<DGCN_AGGR_SOFTMAX>(P, prev_row + 1, first_row, cl * 4, act, p); break;
          default: eg_empty_rows<DGCN_AGGR_POWER>(P, prev_row + 1, first_row, cl * 4, act, p); break;
        }
      }
    }
    State<4> st;
    state_init<DGCN_AGGR_SOFTMAX, 4>(st);   // the initial state is the same for every mode

    // metadata of the first batch: lane m < 16 describes edge ib + m
    int pos = min(ib + n, ie - 1);
    int srcv = P.col[pos];
    int eidv = P.eperm ? P.eperm[pos] : pos;
    int rowv = P.erow[pos];

    for (int b = ib; b < ie; b += kEgM) {
      const int nb = min(kEgM, ie - b);
      // ---- A operand: lane (m = n, kb) reads 16 bytes at float offset 32 c + 16 i2 + 4 kb of its edge's row ----
      const float* arow = P.feat + static_cast<int64_t>(eidv) * P.feat_stride + 4 * kb;
      constexpr int KA = KC > 0 ? KC : 1;
      f4v a[KA][2];
      if constexpr (KC > 0) {
#pragma unroll
        for (int c = 0; c < KC; ++c) {
          a[c][0] = *reinterpret_cast<const f4v*>(arow + c * kEgChunk);
          if (c + 1 < KC || c * kEgChunk + 16 < K) {
            a[c][1] = *reinterpret_cast<const f4v*>(arow + c * kEgChunk + 16);
          } else {
            a[c][1] = f4v{0.f, 0.f, 0.f, 0.f};
          }
        }
      } else {
        a[0][0] = *reinterpret_cast<const f4v*>(arow);                     // K >= 16
        a[0][1] = (16 < K) ? *reinterpret_cast<const f4v*>(arow + 16) : f4v{0.f, 0.f, 0.f, 0.f};
      }

      // ---- accumulators start from x[src] + bias (D layout: lane (n, kb) owns edges 4 kb + j, channels 16 ct + n);
      //      these L2-resident gathers wait together with the feature loads ----
      f4v acc[NT];
      {
        int srcj[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) srcj[j] = __shfl(srcv, 4 * kb + j);
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
          const int ch = ct * 16 + n;
          const bool chok = ch < C;
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[ct][j] = (chok ? row_ptr(P.x, srcj[j], xs32)[ch] : 0.f) + bias[ct];
        }
      }

      // metadata of the next batch while this one computes
      const int posn = min(b + kEgM + n, ie - 1);
      const int srcn = P.col[posn];
      const int eidn = P.eperm ? P.eperm[posn] : posn;
      const int rown = P.erow[posn];

      // ---- tile += F W^T on the matrix cores ----
      auto chunk_mfma = [&](const f4v& a0, const f4v& a1, int c) {
        const float* wl = Wl + n * WS + c * kEgChunk + 4 * kb;
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
          const f4v bw = *reinterpret_cast<const f4v*>(wl + ct * 16 * WS);
          acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, bw.x, acc[ct], 0, 0, 0);
          acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, bw.y, acc[ct], 0, 0, 0);
          acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, bw.z, acc[ct], 0, 0, 0);
          acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, bw.w, acc[ct], 0, 0, 0);
        }
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
          const f4v bw = *reinterpret_cast<const f4v*>(wl + ct * 16 * WS + 16);
          acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, bw.x, acc[ct], 0, 0, 0);
          acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, bw.y, acc[ct], 0, 0, 0);
          acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, bw.z, acc[ct], 0, 0, 0);
          acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, bw.w, acc[ct], 0, 0, 0);
        }
      };
      if constexpr (KC > 0) {
#pragma unroll
        for (int c = 0; c < KC; ++c) chunk_mfma(a[c][0], a[c][1], c);
      } else {
        f4v an0 = a[0][0], an1 = a[0][1];
        for (int c = 0; c < nchunks; ++c) {
          const f4v a0 = an0, a1 = an1;
          if (c + 1 < nchunks) {
            const int o = (c + 1) * kEgChunk;
            an0 = *reinterpret_cast<const f4v*>(arow + o);              // o < K always holds for the first half
            an1 = (o + 16 < K) ? *reinterpret_cast<const f4v*>(arow + o + 16) : f4v{0.f, 0.f, 0.f, 0.f};
          }
          chunk_mfma(a0, a1, c);
        }
      }
      // ---- D layout (row = 4 kb + j, col = n) -> LDS tile [edge][channel] ----
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) {
#pragma unroll
        for (int j = 0; j < 4; ++j) zt[(4 * kb + j) * ZS + ct * 16 + n] = acc[ct][j];
      }
      __builtin_amdgcn_wave_barrier();

      // ---- fold the tile edge by edge into the state of the current destination row ----
      switch (P.mode) {
        case DGCN_AGGR_ADD:
          eg_walk_dispatch<DGCN_AGGR_ADD>(P, wk, item, st, zt, nb, rowv, eidv, cl, act, eps, eps_r, t2, c0s, p); break;
        case DGCN_AGGR_MEAN:
          eg_walk_dispatch<DGCN_AGGR_MEAN>(P, wk, item, st, zt, nb, rowv, eidv, cl, act, eps, eps_r, t2, c0s, p); break;
        case DGCN_AGGR_MAX:
          eg_walk_dispatch<DGCN_AGGR_MAX>(P, wk, item, st, zt, nb, rowv, eidv, cl, act, eps, eps_r, t2, c0s, p); break;
        case DGCN_AGGR_SOFTMAX:
          eg_walk_dispatch<DGCN_AGGR_SOFTMAX>(P, wk, item, st, zt, nb, rowv, eidv, cl, act, eps, eps_r, t2, c0s, p); break;
        default:
          eg_walk_dispatch<DGCN_AGGR_POWER>(P, wk, item, st, zt, nb, rowv, eidv, cl, act, eps, eps_r, t2, c0s, p); break;
      }
      __builtin_amdgcn_wave_barrier();
      srcv = srcn; eidv = eidn; rowv = rown;
    }

    switch (P.mode) {
      case DGCN_AGGR_ADD: eg_finish_item<DGCN_AGGR_ADD>(P, wk, item, ie, st, cl * 4, act, eps_r, p); break;
      case DGCN_AGGR_MEAN: eg_finish_item<DGCN_AGGR_MEAN>(P, wk, item, ie, st, cl * 4, act, eps_r, p); break;
      case DGCN_AGGR_MAX: eg_finish_item<DGCN_AGGR_MAX>(P, wk, item, ie, st, cl * 4, act, eps_r, p); break;
      case DGCN_AGGR_SOFTMAX: eg_finish_item<DGCN_AGGR_SOFTMAX>(P, wk, item, ie, st, cl * 4, act, eps_r, p); break;
      default: eg_finish_item<DGCN_AGGR_POWER>(P, wk, item, ie, st, cl * 4, act, eps_r, p); break;
    }
    if (lane == 0) {
      int32_t* info = P.info + static_cast<int64_t>(item) * kEgInfo;
      info[0] = wk.head_row;
      info[1] = wk.tail_row;
      info[2] = wk.head_cont;
      info[3] = 0;
    }
  }
}

// Rows that straddle item boundaries: the item where such a row STARTS (its tail partial) owns the merge; the
// following items' head partials are folded in item order until one does not continue.  One wave per item.
template <int MODE>
__device__ __forceinline__ void eg_fixup_body(const EgParams& P, int item) {
  const int lane = lane_id();
  const int C = P.C;
  const int row = uni(P.info[static_cast<int64_t>(item) * kEgInfo + 1]);
  if (row < 0) return;
  const float p = P.p_dev ? *P.p_dev : P.p;
  const float deg = static_cast<float>(P.rowptr[row + 1] - P.rowptr[row]);
  for (int c = lane; c < C; c += kWave) {
    State<1> st;
    state_init<MODE, 1>(st);
    int it = item, which = 1;
    while (true) {
      const float* ws = P.part + (static_cast<int64_t>(it) * 2 + which) * 4 * C + c;
      State<1> o;
      state_init<MODE, 1>(o);
      if constexpr (MODE == DGCN_AGGR_MAX) {
        o.a[0] = ws[0];
        o.idx[0] = __float_as_int(ws[C]);
      } else {
        o.a[0] = ws[0]; o.b[0] = ws[C]; o.c[0] = ws[2 * C]; o.d[0] = ws[3 * C];
      }
      state_merge<MODE, 1>(st, o);
      const bool more = (which == 1) || (uni(P.info[static_cast<int64_t>(it) * kEgInfo + 2]) != 0);
      if (!more) break;
      ++it;
      which = 0;
      if (it >= P.n_items || uni(P.info[static_cast<int64_t>(it) * kEgInfo + 0]) != row) break;
    }
    eg_write_row<MODE, 1>(P, row, c, st, deg, p);
  }
}

__global__ __launch_bounds__(kWgThreads) void egemm_fixup_kernel(const EgParams P) {
  const int item = blockIdx.x * kWavesPerWg + (threadIdx.x >> 6);
  if (item >= P.n_items) return;
  switch (P.mode) {
    case DGCN_AGGR_ADD: eg_fixup_body<DGCN_AGGR_ADD>(P, item); break;
    case DGCN_AGGR_MEAN: eg_fixup_body<DGCN_AGGR_MEAN>(P, item); break;
    case DGCN_AGGR_MAX: eg_fixup_body<DGCN_AGGR_MAX>(P, item); break;
    case DGCN_AGGR_SOFTMAX: eg_fixup_body<DGCN_AGGR_SOFTMAX>(P, item); break;
    default: eg_fixup_body<DGCN_AGGR_POWER>(P, item); break;
  }
}

struct EgLayout {
  int nt, kpad, zs, nwaves;
  size_t lds_bytes;
};

inline bool eg_layout(int n_feat, int channels, EgLayout* L) {
  if (channels <= 0 || channels % 4 != 0 || channels > 128) return false;
  if (n_feat < 16 || n_feat % 16 != 0 || n_feat > 256) return false;
  L->nt = (channels + 15) / 16;
  L->kpad = (n_feat + kEgChunk - 1) / kEgChunk * kEgChunk;
  L->zs = L->nt * 16 + 4;                               // % 8 == 4: the D-layout ds_write_b32 is conflict-free
  const size_t wbytes = static_cast<size_t>(L->nt) * 16 * (L->kpad + kEgWPad) * sizeof(float);
  const size_t zbytes = static_cast<size_t>(kEgM) * L->zs * sizeof(float);
  if (wbytes + 2 * zbytes > static_cast<size_t>(kEgLdsBytes)) return false;
  int nw = static_cast<int>((kEgLdsBytes - wbytes) / zbytes);
  if (nw > kEgMaxWaves) nw = kEgMaxWaves;
  L->nwaves = nw;
  L->lds_bytes = wbytes + nw * zbytes;
  return true;
}

template <int NT, int KC>
int launch_egemm(const EgParams& P, const EgLayout& L, hipStream_t s) {
  const void* fn = reinterpret_cast<const void*>(egemm_fwd_kernel<NT, KC>);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(L.lds_bytes));
  if (e != hipSuccess) return static_cast<int>(e);
  int per_cu = static_cast<int>(kEgLdsBytes / L.lds_bytes);
  if (per_cu > 2) per_cu = 2;
  int grid = (P.n_items + L.nwaves - 1) / L.nwaves;
  if (grid > kNumCU * per_cu) grid = kNumCU * per_cu;
  hipLaunchKernelGGL((egemm_fwd_kernel<NT, KC>), dim3(grid), dim3(L.nwaves * kWave), L.lds_bytes, s, P);
  return DGCN_OK;
}

// (channel tiles, feature chunks) of the reference's models get the register-resident feature row; every other
// supported shape takes the generic kernel.  hidden/group: 224/2, 64/2, 80/2 (RevGNN-Deep), 128/2; ungrouped
// hidden 64 and 128 (examples/ogb/ogbn_proteins/model.py, ogbg_ppa/model.py).
int launch_egemm_any(const EgParams& P, const EgLayout& L, hipStream_t s) {
  const int kc = L.kpad / kEgChunk;
#define DGCN_EG_CASE(NTV, KCV) if (L.nt == NTV && kc == KCV) return launch_egemm<NTV, KCV>(P, L, s);
  DGCN_EG_CASE(7, 7)
  DGCN_EG_CASE(2, 2)
  DGCN_EG_CASE(3, 3)
  DGCN_EG_CASE(4, 4)
  DGCN_EG_CASE(4, 2)
  DGCN_EG_CASE(8, 4)
#undef DGCN_EG_CASE
  switch (L.nt) {
    case 1: return launch_egemm<1, 0>(P, L, s);
    case 2: return launch_egemm<2, 0>(P, L, s);
    case 3: return launch_egemm<3, 0>(P, L, s);
    case 4: return launch_egemm<4, 0>(P, L, s);
    case 5: return launch_egemm<5, 0>(P, L, s);
    case 6: return launch_egemm<6, 0>(P, L, s);
    case 7: return launch_egemm<7, 0>(P, L, s);
    default: return launch_egemm<8, 0>(P, L, s);
  }
}

}  // namespace
}  // namespace dgcn

using namespace dgcn;

extern "C" int32_t dgcn_gen_aggr_egemm_supported(int32_t n_feat, int32_t channels) {
  EgLayout L;
  return eg_layout(n_feat, channels, &L) ? 1 : 0;
}

extern "C" size_t dgcn_gen_aggr_egemm_fwd_workspace_bytes(int32_t n_edges, int32_t channels) {
  if (n_edges <= 0 || channels <= 0) return 0;
  const size_t n_items = (static_cast<size_t>(n_edges) + kEgItem - 1) / kEgItem;
  return n_items * (2u * 4u * static_cast<size_t>(channels) * sizeof(float) + kEgInfo * sizeof(int32_t));
}

extern "C" int dgcn_gen_aggr_egemm_fwd_f32(const dgcn_graph* g, const int32_t* erow, const float* x, int64_t x_stride,
                                           const float* edge_feat, int64_t feat_stride, const float* enc_weight,
                                           const float* enc_bias, int32_t n_feat, int32_t channels, int32_t mode,
                                           int32_t msg, int32_t flags, float t, float p, float eps,
                                           const float* t_dev, const float* p_dev, float* out, void* aux1,
                                           float* aux2, int32_t* range_flag, float* z_save, void* workspace,
                                           size_t workspace_bytes, void* stream) {
  if (!g || !x || !out || !edge_feat || !enc_weight) return DGCN_E_NULL;
  EgLayout L;
  if (!eg_layout(n_feat, channels, &L)) return DGCN_E_SHAPE;
  if (g->n_dst <= 0 || g->n_edges <= 0) return DGCN_E_SHAPE;   // an edge-free graph takes dgcn_gen_aggr_fwd_f32
  if ((flags & DGCN_FLAG_ADD_ROOT) && g->n_dst > g->n_src) return DGCN_E_SHAPE;
  if (x_stride < channels || x_stride > 0x7fffffffLL || feat_stride < n_feat) return DGCN_E_SHAPE;
  if (mode < DGCN_AGGR_ADD || mode > DGCN_AGGR_POWER) return DGCN_E_MODE;
  if (msg != DGCN_MSG_IDENTITY && msg != DGCN_MSG_RELU_EPS) return DGCN_E_MODE;
  if (!g->rowptr || !g->col || !erow) return DGCN_E_NULL;
  if (!aligned16(edge_feat) || feat_stride % 4 != 0 || !aligned16(enc_weight) || !aligned16(out) ||
      (aux1 && !aligned16(aux1)) || (aux2 && !aligned16(aux2)) || (z_save && !aligned16(z_save)) ||
      !aligned16(x) || x_stride % 4 != 0 || !aligned16(workspace)) {
    return DGCN_E_ALIGN;
  }
  if (!workspace || workspace_bytes < dgcn_gen_aggr_egemm_fwd_workspace_bytes(g->n_edges, channels)) {
    return DGCN_E_WORKSPACE;
  }
  EgParams P;
  P.n_rows = g->n_dst; P.n_edges = g->n_edges;
  P.n_items = (g->n_edges + kEgItem - 1) / kEgItem;
  P.rowptr = g->rowptr; P.col = g->col; P.eperm = g->eperm; P.erow = erow;
  P.x = x; P.x_stride = x_stride; P.feat = edge_feat; P.feat_stride = feat_stride;
  P.w = enc_weight; P.b = enc_bias;
  P.C = channels; P.K = n_feat; P.Kpad = L.kpad; P.ZS = L.zs;
  P.mode = mode; P.msg = msg;
  P.with_d = (aux2 != nullptr && (mode == DGCN_AGGR_SOFTMAX || mode == DGCN_AGGR_POWER)) ? 1 : 0;
  P.t = t; P.p = p; P.eps = eps; P.t_dev = t_dev; P.p_dev = p_dev;
  P.out = out; P.aux1 = aux1; P.aux2 = aux2;
  P.range_flag = (mode == DGCN_AGGR_SOFTMAX) ? range_flag : nullptr;
  P.add_root = (flags & DGCN_FLAG_ADD_ROOT) ? 1 : 0;
  P.z_save = z_save;
  P.part = static_cast<float*>(workspace);
  P.info = reinterpret_cast<int32_t*>(static_cast<char*>(workspace) +
                                      static_cast<size_t>(P.n_items) * 2u * 4u * channels * sizeof(float));
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int rc = launch_egemm_any(P, L, s);
  if (rc != DGCN_OK) return rc;
  const int fg = (P.n_items + kWavesPerWg - 1) / kWavesPerWg;
  hipLaunchKernelGGL(egemm_fixup_kernel, dim3(fg), dim3(kWgThreads), 0, s, P);
  return launch_status();
}
