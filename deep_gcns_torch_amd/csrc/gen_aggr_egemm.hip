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
//   * a work item = item_len consecutive CSR positions (edges sorted by destination; one item per wave slot of the
//     chip, equally long: exact static balance for any degree distribution), one wave per item, 16 edges per
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

#include "bf16x6.h"
#include "gen_aggr_common.h"
#include "gen_aggr_state.h"

namespace dgcn {
namespace {

constexpr int kEgM = 16;        // edges per MFMA batch (M of the 16x16x4 tile)
constexpr int kEgMinItem = 64;  // a work item = item_len consecutive CSR positions (a multiple of 16, >= 64): the edge
                                // list is cut into at most one item per wave slot of the chip (256 CUs x 8 waves), all
                                // items equally long -> static, exact load balance for any degree distribution, and
                                // only two partial row states per WAVE leave the registers
constexpr int kEgChunk = 32;    // feature floats per k-chunk = one 128-byte line per edge
constexpr int kEgWPad = 8;      // LDS row stride of W = Kpad + 8 floats: (stride/4) % 4 == 2 makes the B-operand
                                // ds_read_b128 of the four fixed lane groups conflict-free
constexpr int kEgMaxWaves = 8;
constexpr int kEgLdsBytes = 160 * 1024;
constexpr int kEgInfo = 4;      // int32 per item: head_row, tail_row, head_continues, unused

// Tuning switches (DGCN_EG_GENERIC / _WPS / _WAVES / _DEBUG) exist only in builds with -DDGCN_EG_TUNING, where the
// environment is read ONCE (thread-safe static); the product build has no mutable or environment-dependent state.
struct EgTuning {
  bool generic;   // force the generic fp32-MFMA kernel
  int wps;        // 2: force the two-waves-per-SIMD register layout
  int waves;      // waves per workgroup (0 = default)
  int dbg;        // phase switches: 1 skip the walk, 2 skip the MFMA chain, 4 skip feature loads (results are wrong)
};
#ifdef DGCN_EG_TUNING
inline const EgTuning& eg_tuning() {
  static const EgTuning t = [] {
    auto num = [](const char* n) { const char* e = getenv(n); return e ? atoi(e) : 0; };
    return EgTuning{getenv("DGCN_EG_GENERIC") != nullptr, num("DGCN_EG_WPS"), num("DGCN_EG_WAVES"), num("DGCN_EG_DEBUG")};
  }();
  return t;
}
#else
inline EgTuning eg_tuning() { return EgTuning{false, 0, 0, 0}; }
#endif

struct EgParams {
  int n_rows, n_edges, n_items, item_len;
  const int32_t* rowptr;
  const int32_t* col;
  const int32_t* eperm;
  const int32_t* erow;      // [E] destination row of every CSR position
  const float* x;
  int64_t x_stride;
  const float* xg;          // gather source of the pipelined kernels: x, or x + bias (prepared in the workspace)
  int64_t xg_stride;
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
  int dbg;                  // tuning builds only (EgTuning::dbg); 0 in the product build
};

// ---- state -> result ------------------------------------------------------------------------------------------
// (same formulas as the epilogue of gen_aggr_fwd_kernel; the softmax sums arrive already corrected for eps)
__device__ __forceinline__ float eg_dead_at(const EgParams& P) {
  return P.msg == DGCN_MSG_RELU_EPS ? P.eps : DGCN_NEG_INF;
}

template <int MODE, int VEC>
__device__ __forceinline__ void eg_finalize(const State<VEC>& st, float deg, float p, float dead_at, float (&res)[VEC],
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
      // m = relu(z) + eps equals eps exactly where no neighbour has z > 0: no edge receives a gradient there, and
      // saying so in the arg-max id lets the backward run without the pre-activations (z_save is not needed for max)
      xi[j] = st.a[j] > dead_at ? st.idx[j] : -1;
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
  eg_finalize<MODE, VEC>(st, deg, p, eg_dead_at(P), res, x1, x2, xi, oor);
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
                                              bool act, float eps, float eps_r, float t2, float c0s, float p,
                                              const f4v& bias4) {
  const int c0 = cl * 4;
  // Common case (average degree >> 16): the whole batch continues the current row -> two edges per fold, their LDS
  // rows requested together, no per-edge row bookkeeping.
  if (nb == kEgM && __builtin_amdgcn_readlane(rowv, 0) == wk.cur_row &&
      __builtin_amdgcn_readlane(rowv, kEgM - 1) == wk.cur_row) {
#pragma unroll 2
    for (int e0 = 0; e0 < kEgM; e0 += 2) {
      float v[2][4];
      bool ok[2] = {true, true};
      int eids[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        eids[u] = __builtin_amdgcn_readlane(eidv, e0 + u);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[u][j] = 0.f;
        if (act) {
          const f4v zz = *reinterpret_cast<const f4v*>(zt + (e0 + u) * P.ZS + c0) + bias4;
          v[u][0] = zz.x; v[u][1] = zz.y; v[u][2] = zz.z; v[u][3] = zz.w;
          if (P.z_save) *reinterpret_cast<f4v*>(P.z_save + static_cast<int64_t>(eids[u]) * P.C + c0) = zz;
        }
      }
      accumulate<MODE, 4, 2, RELU, WITH_D, true>(st, v, ok, eids, eps, t2, c0s, p);
    }
    wk.cnt += kEgM;
    return;
  }
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
        const f4v zz = *reinterpret_cast<const f4v*>(zt + e * P.ZS + c0) + bias4;
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
                                                 bool act, float eps, float eps_r, float t2, float c0s, float p,
                                                 const f4v& bias4) {
  const bool relu = P.msg == DGCN_MSG_RELU_EPS;
  constexpr bool CAN_D = MODE == DGCN_AGGR_SOFTMAX || MODE == DGCN_AGGR_POWER;
  if constexpr (CAN_D) {
    if (P.with_d) {
      if (relu) eg_walk_batch<MODE, true, true>(P, wk, item, st, zt, nb, rowv, eidv, cl, act, eps, eps_r, t2, c0s, p, bias4);
      else eg_walk_batch<MODE, false, true>(P, wk, item, st, zt, nb, rowv, eidv, cl, act, eps, eps_r, t2, c0s, p, bias4);
      return;
    }
  }
  if (relu) eg_walk_batch<MODE, true, false>(P, wk, item, st, zt, nb, rowv, eidv, cl, act, eps, eps_r, t2, c0s, p, bias4);
  else eg_walk_batch<MODE, false, false>(P, wk, item, st, zt, nb, rowv, eidv, cl, act, eps, eps_r, t2, c0s, p, bias4);
}

// End of an item: the row in the registers either ends here (complete, or a head partial that ends) or continues
// into the next item (tail partial; a row covering the whole item stays a head partial that "continues").
template <int MODE>
__device__ __forceinline__ void eg_finish_item(const EgParams& P, EgWalk& wk, int item, int ie, int next_row,
                                               State<4>& st, int c0, bool act, float eps_r, float p) {
  // next_row = row of CSR position ie, or n_rows when there are no more edges
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
  const f4v kZero4 = {0.f, 0.f, 0.f, 0.f};           // the accumulators of this variant already contain the bias

  for (int item = blockIdx.x * nwaves + wave; item < P.n_items; item += gridDim.x * nwaves) {
    const int ib = item * P.item_len;
    const int ie = min(ib + P.item_len, E);
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
          eg_walk_dispatch<DGCN_AGGR_ADD>(P, wk, item, st, zt, nb, rowv, eidv, cl, act, eps, eps_r, t2, c0s, p, kZero4); break;
        case DGCN_AGGR_MEAN:
          eg_walk_dispatch<DGCN_AGGR_MEAN>(P, wk, item, st, zt, nb, rowv, eidv, cl, act, eps, eps_r, t2, c0s, p, kZero4); break;
        case DGCN_AGGR_MAX:
          eg_walk_dispatch<DGCN_AGGR_MAX>(P, wk, item, st, zt, nb, rowv, eidv, cl, act, eps, eps_r, t2, c0s, p, kZero4); break;
        case DGCN_AGGR_SOFTMAX:
          eg_walk_dispatch<DGCN_AGGR_SOFTMAX>(P, wk, item, st, zt, nb, rowv, eidv, cl, act, eps, eps_r, t2, c0s, p, kZero4); break;
        default:
          eg_walk_dispatch<DGCN_AGGR_POWER>(P, wk, item, st, zt, nb, rowv, eidv, cl, act, eps, eps_r, t2, c0s, p, kZero4); break;
      }
      __builtin_amdgcn_wave_barrier();
      srcv = srcn; eidv = eidn; rowv = rown;
    }

    const int next_row = (ie < E) ? uni(P.erow[ie]) : P.n_rows;
    switch (P.mode) {
      case DGCN_AGGR_ADD: eg_finish_item<DGCN_AGGR_ADD>(P, wk, item, ie, next_row, st, cl * 4, act, eps_r, p); break;
      case DGCN_AGGR_MEAN: eg_finish_item<DGCN_AGGR_MEAN>(P, wk, item, ie, next_row, st, cl * 4, act, eps_r, p); break;
      case DGCN_AGGR_MAX: eg_finish_item<DGCN_AGGR_MAX>(P, wk, item, ie, next_row, st, cl * 4, act, eps_r, p); break;
      case DGCN_AGGR_SOFTMAX: eg_finish_item<DGCN_AGGR_SOFTMAX>(P, wk, item, ie, next_row, st, cl * 4, act, eps_r, p); break;
      default: eg_finish_item<DGCN_AGGR_POWER>(P, wk, item, ie, next_row, st, cl * 4, act, eps_r, p); break;
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

// ---- software-pipelined, tile-free variant for compile-time feature widths -----------------------------------------
// Measured on the kernel above (K = 224, C = 112, E = 791 k; DGCN_EG_DEBUG phase switches): MFMA chain 0.31 ms,
// tile walk 0.12 ms, feature-load stalls 0.07 ms, and they ADD UP -- two waves per SIMD, s_setprio and staggering
// change nothing, because v_mfma_f32_16x16x4_f32 runs at the fp32 VECTOR rate on the vector ALU: while one wave
// streams fp32 MFMAs the other wave's VALU work does not issue.  So every instruction counts:
//   * the B fragments of the NT channel tiles are read in two groups; a group's reads are issued before the other
//     group's MFMAs and consumed after them, and inside a group the MFMAs go tile by tile for each k (consecutive
//     MFMAs never touch the same accumulator): no LDS wait, no dependent-accumulator bubble in the chain;
//   * the feature row streams through a register ring kEgAhead 16-float halves ahead, across batch boundaries;
//     the gathered x rows of the next batch are requested into the (just drained) accumulators before the current
//     tile is folded, the index metadata two batches ahead; item bookkeeping comes from the same metadata loads;
//   * the tile is folded where it is: in the MFMA D layout lane (n, q) holds z[edge 4 q + j][channel 16 ct + n],
//     i.e. four CONSECUTIVE edges of NT channels.  A batch that continues the current row (the common case) is
//     folded four edges at a time per lane, all 64 lanes busy (the LDS-tile walk above uses C/4 of the 64 lanes), the
//     running state stays split over the four lane groups and is merged with two xor-shuffles only when the row
//     ends.  No LDS tile, no ds_write / ds_read / barrier per batch, and the LDS holds nothing but the weights.
constexpr int kEgAhead = 4;

struct EgCoord {
  int item, b, ie;
};

struct EgMeta {
  int src, eid;   // of CSR position b + (lane & 15), clamped to the item: source row, original edge id
  int row;        // lanes 0..15: destination row of that position; lanes 16..31: row of position b - 1 (the edge
                  // before the batch); lanes 32..63: row of position ie (the edge after the item)
};
constexpr int kEgLanePrev = 16, kEgLaneNext = 32;

__device__ __forceinline__ EgMeta eg_load_meta(const EgParams& P, const EgCoord& c, int lane) {
  const int pos = min(c.b + (lane & 15), c.ie - 1);
  EgMeta m;
  m.src = P.col[pos];
  m.eid = P.eperm ? P.eperm[pos] : pos;
  int rpos = pos;
  if (lane >= kEgLaneNext) rpos = min(c.ie, P.n_edges - 1);
  else if (lane >= kEgLanePrev) rpos = max(c.b - 1, 0);
  m.row = P.erow[rpos];
  return m;
}

// -- D-layout row output: after the merge every lane holds the full state of channels 16 ct + n; lanes 0..15 write --
template <int MODE, int NT>
__device__ __forceinline__ void egd_write_row(const EgParams& P, int row, int n, bool writer, const State<NT>& st,
                                              float deg, float p) {
  float res[NT], x1[NT], x2[NT];
  int xi[NT];
  bool oor;
  eg_finalize<MODE, NT>(st, deg, p, eg_dead_at(P), res, x1, x2, xi, oor);
  if (!writer) return;
  bool flag = false;
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) {
    const int ch = ct * 16 + n;
    if (ch < P.C) {
      const int64_t o = static_cast<int64_t>(row) * P.C + ch;
      float r = res[ct];
      if (P.add_root) r += P.x[static_cast<int64_t>(row) * P.x_stride + ch];
      P.out[o] = r;
      if constexpr (MODE == DGCN_AGGR_MAX) {
        if (P.aux1) static_cast<int32_t*>(P.aux1)[o] = xi[ct];
      } else if constexpr (MODE == DGCN_AGGR_SOFTMAX || MODE == DGCN_AGGR_POWER) {
        if (P.aux1) static_cast<float*>(P.aux1)[o] = x1[ct];
        if (P.aux2) P.aux2[o] = x2[ct];
        if constexpr (MODE == DGCN_AGGR_SOFTMAX) flag = flag || !(fabsf(x1[ct]) < kShiftSafe);
      }
    }
  }
  if constexpr (MODE == DGCN_AGGR_SOFTMAX) {
    if (P.range_flag && flag) atomicOr(P.range_flag, 1);   // rare
  }
}

template <int MODE, int NT>
__device__ __forceinline__ void egd_store_partial(const EgParams& P, int item, int which, int n, bool writer,
                                                  const State<NT>& st) {
  if (!writer) return;
  float* ws = P.part + (static_cast<int64_t>(item) * 2 + which) * 4 * P.C;
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) {
    const int ch = ct * 16 + n;
    if (ch < P.C) {
      ws[ch] = st.a[ct];
      if constexpr (MODE == DGCN_AGGR_MAX) {
        ws[P.C + ch] = __int_as_float(st.idx[ct]);
      } else {
        ws[P.C + ch] = st.b[ct];
        ws[2 * P.C + ch] = st.c[ct];
        ws[3 * P.C + ch] = st.d[ct];
      }
    }
  }
}

// merge the four lane groups' partial states of the current row (every lane ends up with the full state)
template <int MODE, int NT>
__device__ __forceinline__ void egd_merge_groups(State<NT>& st) {
#pragma unroll
  for (int off = 16; off < kWave; off <<= 1) {
    const State<NT> o = state_shfl_xor<MODE, NT>(st, off);
    state_merge<MODE, NT>(st, o);
  }
}

template <int MODE, int NT>
__device__ __forceinline__ void egd_flush(const EgParams& P, EgWalk& wk, int item, State<NT>& st, int n, bool writer,
                                          float eps_r, float p) {
  egd_merge_groups<MODE, NT>(st);
  eg_fix_eps<MODE, NT>(st, eps_r, P.with_d != 0);
  if (wk.head) {
    egd_store_partial<MODE, NT>(P, item, 0, n, writer, st);
    wk.head_row = wk.cur_row;
    wk.head = 0;
  } else {
    egd_write_row<MODE, NT>(P, wk.cur_row, n, writer, st, static_cast<float>(wk.cnt), p);
  }
}

template <int MODE, int NT>
__device__ __forceinline__ void egd_empty_rows(const EgParams& P, int r0, int r1, int n, bool writer, float p) {
  State<NT> e;
  state_init<MODE, NT>(e);
  for (int r = r0; r < r1; ++r) egd_write_row<MODE, NT>(P, r, n, writer, e, 0.f, p);
}

// Fold one batch held in the D layout (acc[ct][j] = z of edge 4 q + j, channel 16 ct + n) into the running state.
template <int MODE, int NT, bool WITH_D>
__device__ __forceinline__ void egd_walk_batch(const EgParams& P, EgWalk& wk, int item, State<NT>& st,
                                               const f4v (&acc)[NT], int nb, int rowv, int eidv, int n, int q,
                                               bool writer, float eps, float eps_r, float t2, float c0s, float p) {
  constexpr bool NEED_EID = MODE == DGCN_AGGR_MAX;
  int eidq[4] = {0, 0, 0, 0};
  if (NEED_EID || P.z_save) {
#pragma unroll
    for (int j = 0; j < 4; ++j) eidq[j] = __shfl(eidv, 4 * q + j);
  }
  if (P.z_save) {      // pre-activations for the backward, original edge order (64-byte segments per 16 lanes)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (4 * q + j < nb) {
        float* zr = P.z_save + static_cast<int64_t>(eidq[j]) * P.C + n;
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
          if (ct * 16 + n < P.C) zr[ct * 16] = acc[ct][j];
        }
      }
    }
  }
  if (nb == kEgM && __builtin_amdgcn_readlane(rowv, 0) == wk.cur_row &&
      __builtin_amdgcn_readlane(rowv, kEgM - 1) == wk.cur_row) {
    // the whole batch continues the current row: every lane folds its four edges, two at a time
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      float v[2][NT];
      const bool ok[2] = {true, true};
      const int eids[2] = {eidq[2 * h2], eidq[2 * h2 + 1]};
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) v[u][ct] = acc[ct][2 * h2 + u];
      }
      accumulate<MODE, NT, 2, true, WITH_D, true>(st, v, ok, eids, eps, t2, c0s, p);
    }
    wk.cnt += kEgM;
    return;
  }
  // row boundaries inside the batch: edge by edge in CSR order (wave-uniform control), the owning lane group folds
#pragma unroll
  for (int e = 0; e < kEgM; ++e) {
    if (e < nb) {
      const int row = __builtin_amdgcn_readlane(rowv, e);
      if (row != wk.cur_row) {
        egd_flush<MODE, NT>(P, wk, item, st, n, writer, eps_r, p);
        egd_empty_rows<MODE, NT>(P, wk.cur_row + 1, row, n, writer, p);
        state_init<MODE, NT>(st);
        wk.cur_row = row;
        wk.cnt = 0;
      }
      if (q == (e >> 2)) {
        float v[1][NT];
        const bool ok[1] = {true};
        const int eids[1] = {eidq[e & 3]};
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) v[0][ct] = acc[ct][e & 3];
        accumulate<MODE, NT, 1, true, WITH_D, true>(st, v, ok, eids, eps, t2, c0s, p);
      }
      wk.cnt += 1;
    }
  }
}

template <int MODE, int NT>
__device__ __forceinline__ void egd_finish_item(const EgParams& P, EgWalk& wk, int item, int ie, int next_row,
                                                State<NT>& st, int n, bool writer, float eps_r, float p) {
  const bool continues = next_row == wk.cur_row;
  if (continues && !wk.head) {
    egd_merge_groups<MODE, NT>(st);
    eg_fix_eps<MODE, NT>(st, eps_r, P.with_d != 0);
    egd_store_partial<MODE, NT>(P, item, 1, n, writer, st);
    wk.tail_row = wk.cur_row;
  } else {
    wk.head_cont = (continues && wk.head) ? 1 : 0;
    egd_flush<MODE, NT>(P, wk, item, st, n, writer, eps_r, p);
  }
  if (ie >= P.n_edges) egd_empty_rows<MODE, NT>(P, wk.cur_row + 1, P.n_rows, n, writer, p);
}


// ---- the same kernel with the GEMM on the bf16 matrix cores, fp32-faithful ("bf16x6") ----------------------------
// fp32 MFMA runs on the vector ALU at the vector rate; v_mfma_f32_16x16x32_bf16 runs on the matrix pipe at 16x that
// rate and overlaps with VALU work of the other wave.  Every fp32 operand is split EXACTLY into three bf16 values by
// truncation (hi = top 16 bits, mid = top 16 bits of the remainder, lo = top 16 bits of what is left: 3 x 8 = 24
// significand bits, f == hi + mid + lo bit for bit) and the product a*b is accumulated in fp32 from the six largest
// cross terms  a1 b1 + a1 b2 + a2 b1 + a1 b3 + a3 b1 + a2 b2  (the dropped ones are <= 3 * 2^-24 |a b|, i.e. at the
// level of fp32 rounding; measured max error / sum|a||b| = 1.7e-7 against 3.3e-7 for a plain fp32 GEMM).
// 6 MFMAs of 16 cycles replace 8 fp32 MFMAs of 32 cycles per 16x16x32 block: 2.7x less matrix time, and it is
// hidden time.  The weight matrix is split once per workgroup into three bf16 planes in LDS (161 KB at C = 112,
// K = 224: possible because the tile-free fold needs no LDS); features are split in registers as they arrive.
constexpr int kEgAheadB = 2;     // 32-float feature blocks requested ahead of the one being multiplied

// WPS = waves per SIMD the register budget is written for.  2: two full weight-fragment buffers (a plane's fragments
// are requested a whole MFMA phase ahead), feature ring two blocks deep, 256 VGPRs.  3: the wide shapes (NT >= 5) are
// latency-bound at two waves (SQ_WAIT_ANY 43 %, profiles/r02_egemm_analysis.md), so the same chain is written for
// <= 168 VGPRs and 12 waves per CU: the fragments of the two tile halves alternate in two half-size buffers (a
// half is re-filled right after its MFMAs and consumed after the other half's), feature ring one block deep.
template <int NT, int KC, int MODE, int WPS>
__device__ __forceinline__ void egemm_bf16_body(const EgParams& P) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int DBW = WPS == 3 ? 1 : kEgAheadB;
  // the next batch's x rows wait in the idle fragment buffer during the fold -- except where the fold itself needs
  // those registers (wide tiles with the four-float softmax / power state)
  constexpr bool PARK = WPS == 2 && !(NT >= 5 && MODE == DGCN_AGGR_SOFTMAX);
  constexpr int DB = DBW < KC ? DBW : KC;
  constexpr int SU = 4 * KC + 2;            // row stride of a weight plane in 16-byte units: % 4 == 2, conflict-free
  constexpr int PLANE = NT * 16 * SU;       // units per plane
  const int C = P.C, K = P.K;
  const int lane = lane_id();
  const int wave = uni(static_cast<int>(threadIdx.x >> 6));   // scalar: the batch coordinates below stay in SGPRs
  const int nwaves = blockDim.x >> 6;
  i4v* Wp = reinterpret_cast<i4v*>(smem);   // [3][NT*16][SU] units of 8 bf16
  {
    // split the weights once: unit (row r, group g8) holds W[r][8 g8 .. 8 g8 + 7]; rows >= C and columns >= K are zero
    for (int idx = threadIdx.x; idx < NT * 16 * 4 * KC; idx += blockDim.x) {
      const int r = idx / (4 * KC), g8 = idx - r * (4 * KC);
      f4v f0 = {0.f, 0.f, 0.f, 0.f}, f1 = {0.f, 0.f, 0.f, 0.f};
      if (r < C) {
        const float* wr = P.w + static_cast<int64_t>(r) * K + 8 * g8;
        if (8 * g8 < K) f0 = *reinterpret_cast<const f4v*>(wr);
        if (8 * g8 + 4 < K) f1 = *reinterpret_cast<const f4v*>(wr + 4);
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
  const int kq = lane >> 4;                 // k octet of the A / B fragments = row block q of the D tile
  const bool writer = lane < 16;
  const float t = P.t_dev ? *P.t_dev : P.t;
  const float p = P.p_dev ? *P.p_dev : P.p;
  const float eps = P.eps;
  const float eps_r = eps;
  const float t2 = t * 1.4426950408889634f;
  const float c0s = t2 * eps_r;
  const int E = P.n_edges;
  const uint32_t xg32 = static_cast<uint32_t>(P.xg_stride);   // gather source: x, or x + bias prepared by the entry point
  const int istride = gridDim.x * nwaves;

  EgCoord cur;
  cur.item = blockIdx.x * nwaves + wave;
  if (cur.item >= P.n_items) return;
  cur.b = cur.item * P.item_len;
  cur.ie = min(cur.b + P.item_len, E);
  auto next_coord = [&](const EgCoord& c, bool& valid) {
    EgCoord x = c;
    valid = true;
    if (c.b + kEgM < c.ie) {
      x.b = c.b + kEgM;
    } else if (c.item + istride < P.n_items) {
      x.item = c.item + istride;
      x.b = x.item * P.item_len;
      x.ie = min(x.b + P.item_len, E);
    } else {
      valid = false;
    }
    return x;
  };
  // x rows of a batch in the D layout (lane (n, q): edges 4 q + j, channels 16 ct + n), parked in a weight-fragment
  // buffer that is idle between two MFMA chains: requested before the current tile is folded, they are in registers
  // when the next chain starts
  auto gather_x = [&](i4v (&dst)[NT], int srcv) {
    int srcj[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) srcj[j] = __shfl(srcv, 4 * kq + j);
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
      const int ch = ct * 16 + n;
      const bool chok = ch < C;
#pragma unroll
      for (int j = 0; j < 4; ++j) dst[ct][j] = chok ? __float_as_int(row_ptr(P.xg, srcj[j], xg32)[ch]) : 0;
    }
  };
  // raw feature block s of a row: lane (m = n, kq) owns floats 32 s + 8 kq .. + 7 (two 16-byte loads)
  auto load_raw = [&](const float* arow, int sblk, f4v& f0, f4v& f1) {
    f0 = f4v{0.f, 0.f, 0.f, 0.f};
    f1 = f4v{0.f, 0.f, 0.f, 0.f};
    if (P.dbg & 4) return;
    const int o = sblk * kEgChunk + 8 * kq;
    // streamed once: non-temporal, so the 709 MB of features do not evict the gathered x rows (5.9 MB) from L2
    if (sblk + 1 < KC || o < K) f0 = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(arow + o));   // K % 16 == 0
    if (sblk + 1 < KC || o + 4 < K) f1 = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(arow + o + 4));
  };
  const i4v* wb = Wp + n * SU + kq;          // + plane * PLANE + ct * 16 * SU + 4 * s

  bool vn, vnn;
  EgCoord nxt = next_coord(cur, vn);
  EgMeta mc = eg_load_meta(P, cur, lane);
  EgMeta mn = eg_load_meta(P, nxt, lane);
  const float* arow_c = P.feat + static_cast<int64_t>(mc.eid) * P.feat_stride;
  f4v ra[KC][2];
#pragma unroll
  for (int sb = 0; sb < KC; ++sb) { ra[sb][0] = f4v{0.f, 0.f, 0.f, 0.f}; ra[sb][1] = f4v{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int sb = 0; sb < DB; ++sb) load_raw(arow_c, sb, ra[sb][0], ra[sb][1]);
  f4v acc[NT];
  constexpr int G0 = (NT + 1) / 2, G1 = NT - G0;      // WPS == 3: tile halves of the two half-size buffers
  i4v bx[WPS == 3 ? 1 : 2][NT];               // WPS == 2: two weight-plane fragment buffers (bx[1] also parks the next
                                              // batch's x rows between two chains); WPS == 3: one, used as two halves
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) bx[0][ct] = wb[ct * 16 * SU];       // plane 1 of block 0
  if constexpr (PARK) gather_x(bx[1], mc.src);

  EgWalk wk;
  State<NT> st;
  wk.cur_row = -1; wk.cnt = 0; wk.head = 0; wk.head_row = -1; wk.head_cont = 0; wk.tail_row = -1;
  state_init<MODE, NT>(st);

  while (true) {
    const EgCoord nn = next_coord(nxt, vnn);
    EgMeta mnn;
    if constexpr (WPS != 3) mnn = eg_load_meta(P, nn, lane);
    const float* arow_n = P.feat + static_cast<int64_t>(mn.eid) * P.feat_stride;
    if constexpr (PARK) {
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[ct][j] = __int_as_float(bx[1][ct][j]);
      }
    } else {
      // three waves per SIMD cover this gather's (L2) latency; no register left to park it earlier
      i4v (&ai)[NT] = *reinterpret_cast<i4v(*)[NT]>(&acc[0]);
      gather_x(ai, mc.src);
    }

    // ---- tile = x + bias + F W^T, six bf16 MFMAs per 16x16x32 block ----
    if constexpr (WPS == 3) {
      if (!(P.dbg & 2))
#pragma unroll
      for (int sb = 0; sb < KC; ++sb) {
        if (sb + DB < KC) {
          load_raw(arow_c, sb + DB, ra[sb + DB][0], ra[sb + DB][1]);
        } else if (KC > DB) {
          load_raw(arow_n, sb + DB - KC, ra[sb + DB - KC][0], ra[sb + DB - KC][1]);
        }
        i4v a1, a2, a3;
        eg_split3(ra[sb][0], ra[sb][1], a1, a2, a3);
        const int sn = (sb + 1 < KC) ? sb + 1 : 0;
        // bx[0][0 .. G0) = half X, bx[0][G0 .. NT) = half Y; both hold plane 1 of this block on entry
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          // --- half X: MFMAs of this plane, then refill with the next plane (or plane 1 of the next block) ---
#pragma unroll
          for (int g = 0; g < G0; ++g) acc[g] = eg_mfma_bf16(a1, bx[0][g], acc[g]);
          if (pl < 2) {
#pragma unroll
            for (int g = 0; g < G0; ++g) acc[g] = eg_mfma_bf16(a2, bx[0][g], acc[g]);
          }
          if (pl < 1) {
#pragma unroll
            for (int g = 0; g < G0; ++g) acc[g] = eg_mfma_bf16(a3, bx[0][g], acc[g]);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int g = 0; g < G0; ++g) {
            bx[0][g] = (pl < 2) ? wb[(pl + 1) * PLANE + g * 16 * SU + 4 * sb] : wb[g * 16 * SU + 4 * sn];
          }
          __builtin_amdgcn_sched_barrier(0);
          // --- half Y ---
#pragma unroll
          for (int g = G0; g < NT; ++g) acc[g] = eg_mfma_bf16(a1, bx[0][g], acc[g]);
          if (pl < 2) {
#pragma unroll
            for (int g = G0; g < NT; ++g) acc[g] = eg_mfma_bf16(a2, bx[0][g], acc[g]);
          }
          if (pl < 1) {
#pragma unroll
            for (int g = G0; g < NT; ++g) acc[g] = eg_mfma_bf16(a3, bx[0][g], acc[g]);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int g = G0; g < NT; ++g) {
            bx[0][g] = (pl < 2) ? wb[(pl + 1) * PLANE + g * 16 * SU + 4 * sb] : wb[g * 16 * SU + 4 * sn];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr (KC <= DB) {
#pragma unroll
        for (int sb = 0; sb < KC; ++sb) load_raw(arow_n, sb, ra[sb][0], ra[sb][1]);
      }
    } else {
    if (!(P.dbg & 2))
#pragma unroll
      for (int sb = 0; sb < KC; ++sb) {
        constexpr int X = 0;                     // buffer roles alternate with the block parity
        const int bi = sb & 1;                   // holds plane 1 of this block
        const int bo = bi ^ 1;
        (void)X;
        if (sb + DB < KC) {
          load_raw(arow_c, sb + DB, ra[sb + DB][0], ra[sb + DB][1]);
        } else if (KC > DB) {
          load_raw(arow_n, sb + DB - KC, ra[sb + DB - KC][0], ra[sb + DB - KC][1]);
        }
        i4v a1, a2, a3;
        eg_split3(ra[sb][0], ra[sb][1], a1, a2, a3);
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
      if constexpr (KC <= DB) {
#pragma unroll
        for (int sb = 0; sb < KC; ++sb) load_raw(arow_n, sb, ra[sb][0], ra[sb][1]);
      }
      if constexpr (KC % 2 == 1) {
        // an odd number of blocks leaves the next plane 1 in buffer 1: move it where block 0 expects it
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) bx[0][ct] = bx[1][ct];
      }
    }

    if constexpr (PARK) gather_x(bx[1], mn.src);       // next batch's x rows; bx[1] is idle until the next chain
    if constexpr (WPS == 3) mnn = eg_load_meta(P, nn, lane);   // (register-lean: requested here, the fold covers it)

    // ---- fold the tile (it stays in the accumulators) ----
    const int nb = (P.dbg & 1) ? 0 : min(kEgM, cur.ie - cur.b);
    const int item = cur.item;
    if (cur.b == cur.item * P.item_len) {
      const int first_row = __builtin_amdgcn_readlane(mc.row, 0);
      const int prev_row = cur.b > 0 ? __builtin_amdgcn_readlane(mc.row, kEgLanePrev) : -1;
      wk.head_row = -1; wk.tail_row = -1; wk.head_cont = 0; wk.cnt = 0;
      wk.head = (prev_row == first_row) ? 1 : 0;
      wk.cur_row = first_row;
      state_init<MODE, NT>(st);
      if (!wk.head) egd_empty_rows<MODE, NT>(P, prev_row + 1, first_row, n, writer, p);
    }
    constexpr bool CAN_D = MODE == DGCN_AGGR_SOFTMAX || MODE == DGCN_AGGR_POWER;
    if (CAN_D && P.with_d) {
      egd_walk_batch<MODE, NT, CAN_D>(P, wk, item, st, acc, nb, mc.row, mc.eid, n, kq, writer, eps, eps_r, t2, c0s, p);
    } else {
      egd_walk_batch<MODE, NT, false>(P, wk, item, st, acc, nb, mc.row, mc.eid, n, kq, writer, eps, eps_r, t2, c0s, p);
    }
    if (cur.b + kEgM >= cur.ie) {
      const int ie = cur.ie;
      const int next_row = (ie < E) ? __builtin_amdgcn_readlane(mc.row, kEgLaneNext) : P.n_rows;
      egd_finish_item<MODE, NT>(P, wk, item, ie, next_row, st, n, writer, eps_r, p);
      if (lane == 0) {
        int32_t* info = P.info + static_cast<int64_t>(item) * kEgInfo;
        info[0] = wk.head_row;
        info[1] = wk.tail_row;
        info[2] = wk.head_cont;
        info[3] = 0;
      }
    }
    if (!vn) break;
    cur = nxt; mc = mn; nxt = nn; mn = mnn; vn = vnn;
    arow_c = arow_n;
  }
}

template <int NT, int KC, int MODE>
__global__ __launch_bounds__(kEgMaxWaves * kWave) void egemm_fwd_bf16_kernel(const EgParams P) {
  egemm_bf16_body<NT, KC, MODE, 2>(P);
}

template <int NT, int KC, int MODE>
__global__ __launch_bounds__(12 * kWave) void egemm_fwd_bf16_w3_kernel(const EgParams P) {
  egemm_bf16_body<NT, KC, MODE, 3>(P);
}

// Rows that straddle item boundaries: the item where such a row STARTS (it holds the row's tail partial) owns the
// merge, one wave per item.  The row's extent comes from rowptr, so the partials of a long chain (a hub row of
// 11 k edges spans ~30 items) are independent loads: lanes 0..31 / 32..63 hold the channels as float4 and take the
// even / odd items of the chain, the two halves are merged at the end -- a fixed order, deterministic.
template <int MODE>
__device__ __forceinline__ void eg_fixup_body(const EgParams& P, int item) {
  const int lane = lane_id();
  const int C = P.C;
  const int row = uni(P.info[static_cast<int64_t>(item) * kEgInfo + 1]);
  if (row < 0) return;
  const float p = P.p_dev ? *P.p_dev : P.p;
  const int rb = uni(P.rowptr[row]), re = uni(P.rowptr[row + 1]);
  const float deg = static_cast<float>(re - rb);
  const int last = (re - 1) / P.item_len;         // the chain: items item .. last (item == rb / item_len)
  const int half = lane >> 5;
  const int c0 = (lane & 31) * 4;
  const bool act = c0 < C;
  State<4> st;
  state_init<MODE, 4>(st);
  for (int it = item + half; it <= last; it += 2) {
    const int which = (it == item) ? 1 : 0;       // the first item holds the tail partial, the others head partials
    State<4> o;
    state_init<MODE, 4>(o);
    if (act) {
      const float* ws = P.part + (static_cast<int64_t>(it) * 2 + which) * 4 * C + c0;
      load_vec<4>(o.a, ws);
      if constexpr (MODE == DGCN_AGGR_MAX) {
        float fi[4];
        load_vec<4>(fi, ws + C);
#pragma unroll
        for (int q = 0; q < 4; ++q) o.idx[q] = __float_as_int(fi[q]);
      } else {
        load_vec<4>(o.b, ws + C);
        load_vec<4>(o.c, ws + 2 * C);
        load_vec<4>(o.d, ws + 3 * C);
      }
    }
    state_merge<MODE, 4>(st, o);
  }
  {
    const State<4> o = state_shfl_xor<MODE, 4>(st, 32);
    if (half == 0) {
      state_merge<MODE, 4>(st, o);
      if (act) eg_write_row<MODE, 4>(P, row, c0, st, deg, p);
    }
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

// item length: one item per wave slot of the chip, a multiple of the 16-edge batch, at least kEgMinItem
inline int eg_item_len(int n_edges, int waves_per_cu) {
  const int64_t slots = static_cast<int64_t>(num_cus()) * waves_per_cu;
  int64_t len = (n_edges + slots - 1) / slots;
  len = (len + kEgM - 1) / kEgM * kEgM;
  return static_cast<int>(len < kEgMinItem ? kEgMinItem : len);
}
inline int eg_num_items(int n_edges, int waves_per_cu) {
  const int len = eg_item_len(n_edges, waves_per_cu);
  return (n_edges + len - 1) / len;
}

// xg[r][c] = x[r][c] + bias[c]: the pipelined kernels gather from it, so the bias costs no register and no add
__global__ __launch_bounds__(kWgThreads) void eg_bias_rows_kernel(const float* __restrict__ x, int64_t x_stride,
                                                                  const float* __restrict__ b, float* __restrict__ xg,
                                                                  int64_t n4, int c4) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const int64_t r = i / c4;
    const int cc = static_cast<int>(i - r * c4) * 4;
    const f4v xv = *reinterpret_cast<const f4v*>(x + r * x_stride + cc);
    const f4v bv = *reinterpret_cast<const f4v*>(b + cc);
    *reinterpret_cast<f4v*>(xg + r * (static_cast<int64_t>(c4) * 4) + cc) = xv + bv;
  }
}

// waves per CU of the kernel that serves (n_feat, channels): 12 for the wide pipelined shapes (register-lean variant),
// 8 for the other pipelined shapes, the LDS-limited count for the generic kernel
inline bool eg_pipelined_shape(int nt, int kc) {
  return (nt == 7 && kc == 7) || (nt == 2 && kc == 2) || (nt == 3 && kc == 3) || (nt == 4 && kc == 4) ||
         (nt == 4 && kc == 2);
}
inline int eg_wps(int nt, int mode) {
  if (eg_tuning().wps == 2) return 2;
  // the register-lean variant pays off where the fold state is small (max: 0.378 -> 0.347 ms at K = 224, C = 112);
  // the softmax / power folds spill in it and stay on the two-waves layout
  const bool small_state = mode == DGCN_AGGR_MAX || mode == DGCN_AGGR_ADD || mode == DGCN_AGGR_MEAN;
  return (nt >= 5 && small_state) ? 3 : 2;
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

inline int eg_waves_per_cu(const EgLayout& L, int msg, int mode) {
  const int kc = L.kpad / kEgChunk;
  if (msg == DGCN_MSG_RELU_EPS && eg_pipelined_shape(L.nt, kc) && !eg_tuning().generic) {
    return eg_wps(L.nt, mode) == 3 ? 12 : kEgMaxWaves;
  }
  return L.nwaves;
}

template <int NT, int KC>
int launch_egemm(const EgParams& P, const EgLayout& L, hipStream_t s) {
  const void* fn = reinterpret_cast<const void*>(egemm_fwd_kernel<NT, KC>);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(L.lds_bytes));
  if (e != hipSuccess) return static_cast<int>(e);
  int per_cu = static_cast<int>(kEgLdsBytes / L.lds_bytes);
  if (per_cu > 2) per_cu = 2;
  int grid = (P.n_items + L.nwaves - 1) / L.nwaves;
  if (grid > num_cus() * per_cu) grid = num_cus() * per_cu;
  hipLaunchKernelGGL((egemm_fwd_kernel<NT, KC>), dim3(grid), dim3(L.nwaves * kWave), L.lds_bytes, s, P);
  return DGCN_OK;
}


template <int NT, int KC, int MODE>
int launch_egemm_bf16_mode(const EgParams& P, hipStream_t s) {
  const size_t lds = static_cast<size_t>(3) * NT * 16 * (4 * KC + 2) * 16;                  // three bf16 weight planes
  // the register-lean variant exists only where eg_wps can choose it (wide shapes, small fold state)
  constexpr bool kHasW3 = NT >= 5 && (MODE == DGCN_AGGR_MAX || MODE == DGCN_AGGR_ADD || MODE == DGCN_AGGR_MEAN);
  bool w3 = false;
  if constexpr (kHasW3) w3 = eg_wps(NT, MODE) == 3;
  const void* fn = reinterpret_cast<const void*>(egemm_fwd_bf16_kernel<NT, KC, MODE>);
  if constexpr (kHasW3) {
    if (w3) fn = reinterpret_cast<const void*>(egemm_fwd_bf16_w3_kernel<NT, KC, MODE>);
  }
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
  if (e != hipSuccess) return static_cast<int>(e);
  int nwaves = w3 ? 12 : kEgMaxWaves;
  if (const int w = eg_tuning().waves; w >= 1 && w <= nwaves) nwaves = w;
  int grid = (P.n_items + nwaves - 1) / nwaves;
  if (grid > num_cus()) grid = num_cus();                 // the weight planes fill the LDS: one workgroup per CU
  if constexpr (kHasW3) {
    if (w3) {
      hipLaunchKernelGGL((egemm_fwd_bf16_w3_kernel<NT, KC, MODE>), dim3(grid), dim3(nwaves * kWave), lds, s, P);
      return DGCN_OK;
    }
  }
  hipLaunchKernelGGL((egemm_fwd_bf16_kernel<NT, KC, MODE>), dim3(grid), dim3(nwaves * kWave), lds, s, P);
  return DGCN_OK;
}

template <int NT, int KC>
int launch_egemm_bf16(const EgParams& P, hipStream_t s) {
  switch (P.mode) {
    case DGCN_AGGR_ADD: return launch_egemm_bf16_mode<NT, KC, DGCN_AGGR_ADD>(P, s);
    case DGCN_AGGR_MEAN: return launch_egemm_bf16_mode<NT, KC, DGCN_AGGR_MEAN>(P, s);
    case DGCN_AGGR_MAX: return launch_egemm_bf16_mode<NT, KC, DGCN_AGGR_MAX>(P, s);
    case DGCN_AGGR_SOFTMAX: return launch_egemm_bf16_mode<NT, KC, DGCN_AGGR_SOFTMAX>(P, s);
    default: return launch_egemm_bf16_mode<NT, KC, DGCN_AGGR_POWER>(P, s);
  }
}


// (channel tiles, feature chunks) of the reference's models get the pipelined kernel; every other supported shape
// (and the identity message) takes the generic one.  hidden/group: 224/2, 64/2, 80/2 (RevGNN-Deep), 128/2; ungrouped
// hidden 64 and 128 (examples/ogb/ogbn_proteins/model.py, ogbg_ppa/model.py).
int launch_egemm_any(const EgParams& P, const EgLayout& L, hipStream_t s) {
  const int kc = L.kpad / kEgChunk;
  const bool pipe_ok = P.msg == DGCN_MSG_RELU_EPS && !eg_tuning().generic && eg_pipelined_shape(L.nt, kc);
#define DGCN_EG_CASE(NTV, KCV) if (pipe_ok && L.nt == NTV && kc == KCV) return launch_egemm_bf16<NTV, KCV>(P, s);
  DGCN_EG_CASE(7, 7)
  DGCN_EG_CASE(2, 2)
  DGCN_EG_CASE(3, 3)
  DGCN_EG_CASE(4, 4)
  DGCN_EG_CASE(4, 2)
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

extern "C" size_t dgcn_gen_aggr_egemm_fwd_workspace_bytes(int32_t n_edges, int32_t n_src, int32_t n_feat,
                                                          int32_t channels) {
  EgLayout L;
  if (n_edges <= 0 || n_src <= 0 || !eg_layout(n_feat, channels, &L)) return 0;
  // the message kind is not known here: size for the finer of the two possible item cuts
  const size_t n_items = static_cast<size_t>(eg_num_items(n_edges, 12));
  const size_t part = n_items * (2u * 4u * static_cast<size_t>(channels) * sizeof(float) + kEgInfo * sizeof(int32_t));
  const size_t xg = static_cast<size_t>(n_src) * static_cast<size_t>(channels) * sizeof(float);
  return ((part + 255) & ~static_cast<size_t>(255)) + xg;
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
      (enc_bias && !aligned16(enc_bias)) ||
      !aligned16(x) || x_stride % 4 != 0 || !aligned16(workspace)) {
    return DGCN_E_ALIGN;
  }
  if (!workspace || workspace_bytes < dgcn_gen_aggr_egemm_fwd_workspace_bytes(g->n_edges, g->n_src, n_feat, channels)) {
    return DGCN_E_WORKSPACE;
  }
  EgParams P;
  P.n_rows = g->n_dst; P.n_edges = g->n_edges;
  const int wpc = eg_waves_per_cu(L, msg, mode);
  P.item_len = eg_item_len(g->n_edges, wpc);
  P.n_items = eg_num_items(g->n_edges, wpc);
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
  P.dbg = eg_tuning().dbg;
  P.part = static_cast<float*>(workspace);
  P.info = reinterpret_cast<int32_t*>(static_cast<char*>(workspace) +
                                      static_cast<size_t>(P.n_items) * 2u * 4u * channels * sizeof(float));
  hipStream_t s = static_cast<hipStream_t>(stream);
  P.xg = x; P.xg_stride = x_stride;
  if (enc_bias && msg == DGCN_MSG_RELU_EPS && eg_pipelined_shape(L.nt, L.kpad / kEgChunk) && !eg_tuning().generic) {
    // gather source with the bias folded in, behind the partial-state area of the workspace
    const size_t n_items12 = static_cast<size_t>(eg_num_items(g->n_edges, 12));
    const size_t part = n_items12 * (2u * 4u * static_cast<size_t>(channels) * sizeof(float) + kEgInfo * sizeof(int32_t));
    float* xg = reinterpret_cast<float*>(static_cast<char*>(workspace) + ((part + 255) & ~static_cast<size_t>(255)));
    const int64_t n4 = static_cast<int64_t>(g->n_src) * (channels / 4);
    int64_t blocks = (n4 + kWgThreads - 1) / kWgThreads;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(eg_bias_rows_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kWgThreads), 0, s, x, x_stride,
                       enc_bias, xg, n4, channels / 4);
    P.xg = xg; P.xg_stride = channels;
  }
  const int rc = launch_egemm_any(P, L, s);
  if (rc != DGCN_OK) return rc;
  const int fg = (P.n_items + kWavesPerWg - 1) / kWavesPerWg;
  hipLaunchKernelGGL(egemm_fixup_kernel, dim3(fg), dim3(kWgThreads), 0, s, P);
  return launch_status();
}
