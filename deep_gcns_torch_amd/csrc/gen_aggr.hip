// Sparse generalized aggregation for gfx950 (MI355X): forward and backward.
//
// Replaces, in ONE pass per direction and without any (E,C) temporary,
//   GENConv.propagate/message          gcn_lib/sparse/torch_vertex.py:68,78-85
//   GenMessagePassing.aggregate        gcn_lib/sparse/torch_message.py:44-85
//   torch_scatter scatter/scatter_softmax/scatter_max underneath them.
//
// Execution shape (wave = 64 lanes):
//   * one wave owns one work item = one destination row (or a <=chunk slice of a hub row);
//   * a row of C fp32 channels is covered by LPR = C/4 lanes holding a float4 each, so a
//     wave walks G = 64/LPR edges of the SAME row at once: every global_load_dwordx4 of the
//     wave fetches G full, 16B-aligned source rows (C=128: 2 rows = 1 KiB per instruction);
//   * column indices are read 64 at a time with one coalesced load and handed to the edge
//     groups with ds_bpermute, so the index fetch is off the gather's critical path;
//   * U batches of loads are issued back to back before any is consumed (memory-level
//     parallelism), the reduction state lives in registers (online softmax: running
//     max / denominator / weighted sum per channel), and the G partial states are combined
//     with wave shuffles at the end.  No atomics anywhere: results are bit-reproducible.
//
// The bound is HBM: algorithmic bytes per launch = E*(4C+4) + N*4C + 4(N+1)  (DESIGN.md).

#include "dgcn_common.h"

namespace dgcn {
namespace {

constexpr float kShiftSafe = 80.f;  // |L| below this keeps exp(-L) and exp(t*m) inside the fp32 range
constexpr float kPowLo = 1e-7f;  // torch_message.py:69
constexpr float kPowHi = 1e1f;

struct WalkGraph {
  int n_rows;
  int n_work;
  const int32_t* rowptr;
  const int32_t* col;
  const int32_t* eperm;
  const int32_t* work_row;
  const int32_t* work_beg;
  const int32_t* work_end;
  const int32_t* work_slot;
  int n_split;
  const int32_t* split_item;
};

struct FwdParams {
  WalkGraph g;
  const float* x;
  int64_t x_stride;
  const float* ea;
  int C;
  int msg;
  float t, p, eps;
  const float* t_dev;
  const float* p_dev;
  float* out;
  void* aux1;
  float* aux2;
  int32_t* range_flag;  // softmax: set to 1 when some |L_i| >= kShiftSafe (the backward then gathers two rows)
  float* ws;  // partial slots: [slot][4][C]
};

struct BwdParams {
  WalkGraph g;      // transposed walk: rows = sources, col = destinations
  const float* x;
  int64_t x_stride;
  const float* ea;
  int C;
  int msg;
  int learn_t;
  float t, p, eps;
  const float* t_dev;
  const float* p_dev;
  const float* gcoef;
  const void* aux1;
  const float* out;
  const float* gshift;    // [n_dst, C] g_i * exp(kshift_c - L_i)  (single-gather softmax backward) or null
  const float* kshift;    // [C] per-channel shift
  const int32_t* shift_ok;  // device flag: 1 = the shifted form is numerically safe for this call
  float* grad_x;
  float* grad_ea;
  float* ws;  // partial slots: [slot][C]
};

struct Work {
  int row, beg, end, slot;
};

__device__ __forceinline__ Work fetch_work(const WalkGraph& g, int item) {
  Work w;
  if (g.n_work) {
    w.row = uni(g.work_row[item]);
    w.beg = uni(g.work_beg[item]);
    w.end = uni(g.work_end[item]);
    w.slot = uni(g.work_slot[item]);
  } else {
    w.row = item;
    w.beg = uni(g.rowptr[item]);
    w.end = uni(g.rowptr[item + 1]);
    w.slot = -1;
  }
  return w;
}

// Block b runs on XCD b % 8 (observed dispatch order; used for L2 affinity only).  Remap so
// that, within one grid-stride sweep, each XCD covers a contiguous range of rows: rows that
// are adjacent in a locality-ordered graph then share their neighbours' lines in one L2.
__device__ __forceinline__ int virtual_block() {
  const int per = gridDim.x / kNumXCD;  // gridDim.x is a multiple of 8
  return (blockIdx.x % kNumXCD) * per + blockIdx.x / kNumXCD;
}

__device__ __forceinline__ float msg_apply(float z, int msg, float eps) {
  return msg == DGCN_MSG_RELU_EPS ? fmaxf(z, 0.f) + eps : z;
}

__device__ __forceinline__ float fast_pow(float u, float p) {  // u > 0
  return fast_exp2(p * fast_log2(u));
}

// ---------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------
// Per-channel reduction state.  Meaning by mode:
//   SOFTMAX: a = running max M of t*m, b = sum exp(s-M), c = sum exp(s-M)*m, d = sum exp(s-M)*m^2
//   POWER  : b = sum u^p, d = sum u^p ln u
//   ADD/MEAN: b = sum m
//   MAX    : a = best m, idx = original edge id of the first maximal edge
template <int VEC>
struct State {
  float a[VEC], b[VEC], c[VEC], d[VEC];
  int idx[VEC];
};

template <int MODE, int VEC>
__device__ __forceinline__ void state_init(State<VEC>& s) {
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    s.a[j] = DGCN_NEG_INF;
    s.b[j] = 0.f;
    s.c[j] = 0.f;
    s.d[j] = 0.f;
    s.idx[j] = -1;
  }
}

// merge `o` (another partial of the same row) into `s`
template <int MODE, int VEC>
__device__ __forceinline__ void state_merge(State<VEC>& s, const State<VEC>& o) {
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    if constexpr (MODE == DGCN_AGGR_SOFTMAX) {
      const float nm = fmaxf(s.a[j], o.a[j]);
      const float s1 = (s.a[j] == DGCN_NEG_INF) ? 0.f : fast_exp(s.a[j] - nm);
      const float s2 = (o.a[j] == DGCN_NEG_INF) ? 0.f : fast_exp(o.a[j] - nm);
      s.b[j] = s.b[j] * s1 + o.b[j] * s2;
      s.c[j] = s.c[j] * s1 + o.c[j] * s2;
      s.d[j] = s.d[j] * s1 + o.d[j] * s2;
      s.a[j] = nm;
    } else if constexpr (MODE == DGCN_AGGR_MAX) {
      const bool take = (o.a[j] > s.a[j]) ||
                        (o.a[j] == s.a[j] && o.idx[j] >= 0 && (s.idx[j] < 0 || o.idx[j] < s.idx[j]));
      if (take) {
        s.a[j] = o.a[j];
        s.idx[j] = o.idx[j];
      }
    } else {
      s.b[j] += o.b[j];
      s.d[j] += o.d[j];
    }
  }
}

template <int MODE, int VEC>
__device__ __forceinline__ State<VEC> state_shfl_xor(const State<VEC>& s, int off) {
  State<VEC> o;
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    o.a[j] = o.b[j] = o.c[j] = o.d[j] = 0.f;
    o.idx[j] = -1;
    if constexpr (MODE == DGCN_AGGR_SOFTMAX) {
      o.a[j] = __shfl_xor(s.a[j], off);
      o.b[j] = __shfl_xor(s.b[j], off);
      o.c[j] = __shfl_xor(s.c[j], off);
      o.d[j] = __shfl_xor(s.d[j], off);
    } else if constexpr (MODE == DGCN_AGGR_MAX) {
      o.a[j] = __shfl_xor(s.a[j], off);
      o.idx[j] = __shfl_xor(s.idx[j], off);
    } else {
      o.b[j] = __shfl_xor(s.b[j], off);
      o.d[j] = __shfl_xor(s.d[j], off);
    }
  }
  return o;
}

// Fold U gathered rows (this lane's VEC channels of each) into the state.
template <int MODE, int VEC, int U>
__device__ __forceinline__ void accumulate(State<VEC>& st, const float (&v)[U][VEC],
                                           const bool (&ok)[U], const int (&eid)[U], int msg,
                                           float eps, float t, float p) {
  if (!ok[0]) return;  // ok[] is monotone in u: nothing valid for this edge group
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    if constexpr (MODE == DGCN_AGGR_SOFTMAX) {
      float m[U], s[U];
      float bm = DGCN_NEG_INF;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        m[u] = msg_apply(v[u][j], msg, eps);
        s[u] = ok[u] ? t * m[u] : DGCN_NEG_INF;
        bm = fmaxf(bm, s[u]);
      }
      const float nm = fmaxf(st.a[j], bm);
      const float sc = fast_exp(st.a[j] - nm);  // exp(-inf) = 0 on the first batch
      float D = st.b[j] * sc, A = st.c[j] * sc, A2 = st.d[j] * sc;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float e = fast_exp(s[u] - nm);  // 0 for masked edges
        const float em = e * m[u];
        D += e;
        A += em;
        A2 = fmaf(em, m[u], A2);
      }
      st.a[j] = nm;
      st.b[j] = D;
      st.c[j] = A;
      st.d[j] = A2;
    } else if constexpr (MODE == DGCN_AGGR_POWER) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (ok[u]) {
          const float m = msg_apply(v[u][j], msg, eps);
          const float uu = fminf(fmaxf(m, kPowLo), kPowHi);
          const float l2 = fast_log2(uu);
          const float up = fast_exp2(p * l2);
          st.b[j] += up;
          st.d[j] = fmaf(up, l2 * 0.6931471805599453f, st.d[j]);
        }
      }
    } else if constexpr (MODE == DGCN_AGGR_MAX) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (ok[u]) {
          const float m = msg_apply(v[u][j], msg, eps);
          // strict '>' keeps the FIRST maximal edge (edges arrive in increasing id per group)
          if (m > st.a[j] || st.idx[j] < 0) {
            st.a[j] = m;
            st.idx[j] = eid[u];
          }
        }
      }
    } else {  // ADD / MEAN
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (ok[u]) st.b[j] += msg_apply(v[u][j], msg, eps);
      }
    }
  }
}

template <int MODE, int VEC, int LPR, bool HAS_EA>
__global__ __launch_bounds__(kWgThreads) void gen_aggr_fwd_kernel(const FwdParams P) {
  constexpr int G = kWave / LPR;            // edges walked in parallel by one wave
#ifdef DGCN_FWD_U
  constexpr int U = DGCN_FWD_U;
#else
  constexpr int U = (VEC == 4) ? 4 : 8;     // load batches in flight per lane
#endif
  constexpr bool NEED_EID = HAS_EA || MODE == DGCN_AGGR_MAX;

  const int lane = lane_id();
  const int g = lane / LPR;
  const int cl = lane % LPR;
  const int C = P.C;
  const int n_items = P.g.n_work ? P.g.n_work : P.g.n_rows;
  const int total_waves = gridDim.x * kWavesPerWg;
  const int wave0 = virtual_block() * kWavesPerWg + (threadIdx.x >> 6);
  const float t = P.t_dev ? *P.t_dev : P.t;
  const float p = P.p_dev ? *P.p_dev : P.p;
  const float eps = P.eps;
  const int msg = P.msg;

  for (int item = wave0; item < n_items; item += total_waves) {
    const Work w = fetch_work(P.g, item);
    for (int cb = 0; cb < C; cb += LPR * VEC) {
      const int c0 = cb + cl * VEC;
      const bool act = c0 < C;
      State<VEC> st;
      state_init<MODE, VEC>(st);

      for (int blk = w.beg; blk < w.end; blk += kWave) {
        const int nb = min(kWave, w.end - blk);
        int mycol = 0, myeid = 0;
        if (lane < nb) {
          mycol = P.g.col[blk + lane];
          if constexpr (NEED_EID) myeid = P.g.eperm ? P.g.eperm[blk + lane] : blk + lane;
        }
        for (int s0 = 0; s0 < nb; s0 += G * U) {
          float v[U][VEC];
          bool ok[U];
          int eid[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int ei = s0 + u * G + g;
            ok[u] = ei < nb;
            const int src = __shfl(mycol, ei & (kWave - 1));
            eid[u] = 0;
            if constexpr (NEED_EID) eid[u] = __shfl(myeid, ei & (kWave - 1));
#pragma unroll
            for (int j = 0; j < VEC; ++j) v[u][j] = 0.f;
            if (ok[u] && act) {
              load_vec<VEC>(v[u], P.x + static_cast<int64_t>(src) * P.x_stride + c0);
              if constexpr (HAS_EA) {
                float a[VEC];
                load_vec<VEC>(a, P.ea + static_cast<int64_t>(eid[u]) * C + c0);
#pragma unroll
                for (int j = 0; j < VEC; ++j) v[u][j] += a[j];
              }
            }
          }
          accumulate<MODE, VEC, U>(st, v, ok, eid, msg, eps, t, p);
        }
      }

      // combine the G edge groups of this wave (all lanes participate)
#pragma unroll
      for (int off = LPR; off < kWave; off <<= 1) {
        const State<VEC> o = state_shfl_xor<MODE, VEC>(st, off);
        state_merge<MODE, VEC>(st, o);
      }

      if (g == 0 && act) {
        if (w.slot >= 0) {
          float* ws = P.ws + (static_cast<int64_t>(w.slot) * 4) * C + c0;
          if constexpr (MODE == DGCN_AGGR_MAX) {
            float fi[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) fi[j] = __int_as_float(st.idx[j]);
            store_vec<VEC>(ws, st.a);
            store_vec<VEC>(ws + C, fi);
          } else {
            store_vec<VEC>(ws, st.a);
            store_vec<VEC>(ws + C, st.b);
            store_vec<VEC>(ws + 2 * C, st.c);
            store_vec<VEC>(ws + 3 * C, st.d);
          }
        } else {
          const int64_t o = static_cast<int64_t>(w.row) * C + c0;
          const float deg = static_cast<float>(w.end - w.beg);
          float res[VEC], x1[VEC], x2[VEC];
          int xi[VEC];
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            x1[j] = 0.f; x2[j] = 0.f; xi[j] = -1;
            if constexpr (MODE == DGCN_AGGR_SOFTMAX) {
              const bool any = st.b[j] > 0.f;
              const float inv = any ? 1.f / st.b[j] : 0.f;
              res[j] = st.c[j] * inv;
              x1[j] = any ? st.a[j] + fast_log(st.b[j]) : 0.f;
              x2[j] = st.d[j] * inv;
              if (P.range_flag && !(fabsf(x1[j]) < kShiftSafe)) atomicOr(P.range_flag, 1);  // rare
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
          store_vec<VEC>(P.out + o, res);
          if constexpr (MODE == DGCN_AGGR_MAX) {
            if (P.aux1) store_vec_i<VEC>(static_cast<int32_t*>(P.aux1) + o, xi);
          } else if constexpr (MODE == DGCN_AGGR_SOFTMAX || MODE == DGCN_AGGR_POWER) {
            if (P.aux1) store_vec<VEC>(static_cast<float*>(P.aux1) + o, x1);
            if (P.aux2) store_vec<VEC>(P.aux2 + o, x2);
          }
        }
      }
    }
  }
}

// Merge the partial slots of split (hub) rows: one wave per split row (its first work item is listed in
// split_item), lanes over channels, slots folded in work-list order -> deterministic.
template <int MODE>
__global__ __launch_bounds__(kWgThreads) void gen_aggr_fwd_merge_kernel(const FwdParams P) {
  const int lane = lane_id();
  const int C = P.C;
  const int n_work = P.g.n_work;
  const int wave = blockIdx.x * kWavesPerWg + (threadIdx.x >> 6);
  if (wave >= P.g.n_split) return;
  const float p = P.p_dev ? *P.p_dev : P.p;
  const int i0 = uni(P.g.split_item[wave]);
  const int row = uni(P.g.work_row[i0]);
  const float deg = static_cast<float>(P.g.rowptr[row + 1] - P.g.rowptr[row]);
  int i1 = i0;
  while (i1 < n_work && uni(P.g.work_row[i1]) == row) ++i1;
  for (int c = lane; c < C; c += kWave) {
    State<1> st;
    state_init<MODE, 1>(st);
    for (int i = i0; i < i1; ++i) {
      const float* ws = P.ws + (static_cast<int64_t>(P.g.work_slot[i]) * 4) * C + c;
      State<1> o;
      state_init<MODE, 1>(o);
      if constexpr (MODE == DGCN_AGGR_MAX) {
        o.a[0] = ws[0];
        o.idx[0] = __float_as_int(ws[C]);
      } else {
        o.a[0] = ws[0];
        o.b[0] = ws[C];
        o.c[0] = ws[2 * C];
        o.d[0] = ws[3 * C];
      }
      state_merge<MODE, 1>(st, o);
    }
    const int64_t o = static_cast<int64_t>(row) * C + c;
    float res, x1 = 0.f, x2 = 0.f;
    if constexpr (MODE == DGCN_AGGR_SOFTMAX) {
      const bool any = st.b[0] > 0.f;
      const float inv = any ? 1.f / st.b[0] : 0.f;
      res = st.c[0] * inv;
      x1 = any ? st.a[0] + fast_log(st.b[0]) : 0.f;
      x2 = st.d[0] * inv;
      if (P.range_flag && !(fabsf(x1) < kShiftSafe)) atomicOr(P.range_flag, 1);
    } else if constexpr (MODE == DGCN_AGGR_POWER) {
      const float q = st.b[0] / fmaxf(deg, 1.f);
      const float r = fminf(fmaxf(q, kPowLo), kPowHi);
      res = fast_pow(r, 1.f / p);
      x1 = q;
      x2 = st.d[0];
    } else if constexpr (MODE == DGCN_AGGR_MAX) {
      res = st.idx[0] >= 0 ? st.a[0] : 0.f;
    } else if constexpr (MODE == DGCN_AGGR_MEAN) {
      res = st.b[0] / fmaxf(deg, 1.f);
    } else {
      res = st.b[0];
    }
    P.out[o] = res;
    if constexpr (MODE == DGCN_AGGR_MAX) {
      if (P.aux1) static_cast<int32_t*>(P.aux1)[o] = st.idx[0];
    } else if constexpr (MODE == DGCN_AGGR_SOFTMAX || MODE == DGCN_AGGR_POWER) {
      if (P.aux1) static_cast<float*>(P.aux1)[o] = x1;
      if (P.aux2) P.aux2[o] = x2;
    }
  }
}

// ---------------------------------------------------------------------------------------
// backward: walk the CSC (rows = sources).  For CSC position e with destination i and
// original edge id oe:   dz_e = R(z_e) * K(m_e, i)      
// ---------------------------------------------------------------------------------------
constexpr int kModeSoftmaxShifted = 100;  // internal: softmax backward with ONE gathered row per edge

template <int MODE, int VEC, int LPR, bool HAS_EA>
__device__ __forceinline__ void gen_aggr_bwd_body(const BwdParams& P) {
  constexpr int G = kWave / LPR;
  constexpr int U = (VEC == 4) ? 4 : 8;
  constexpr bool NEED_EID = HAS_EA || MODE == DGCN_AGGR_MAX;

  const int lane = lane_id();
  const int g = lane / LPR;
  const int cl = lane % LPR;
  const int C = P.C;
  const int n_items = P.g.n_work ? P.g.n_work : P.g.n_rows;
  const int total_waves = gridDim.x * kWavesPerWg;
  const int wave0 = virtual_block() * kWavesPerWg + (threadIdx.x >> 6);
  const float t = P.t_dev ? *P.t_dev : P.t;
  const float p = P.p_dev ? *P.p_dev : P.p;
  const float eps = P.eps;
  const int msg = P.msg;
  const bool learn_t = P.learn_t != 0;

  for (int item = wave0; item < n_items; item += total_waves) {
    const Work w = fetch_work(P.g, item);
    for (int cb = 0; cb < C; cb += LPR * VEC) {
      const int c0 = cb + cl * VEC;
      const bool act = c0 < C;
      float xs[VEC], acc[VEC], ksh[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) { xs[j] = 0.f; acc[j] = 0.f; ksh[j] = 0.f; }
      if (act) load_vec<VEC>(xs, P.x + static_cast<int64_t>(w.row) * P.x_stride + c0);
      if constexpr (MODE == kModeSoftmaxShifted) {
        if (act) load_vec<VEC>(ksh, P.kshift + c0);
      }

      for (int blk = w.beg; blk < w.end; blk += kWave) {
        const int nb = min(kWave, w.end - blk);
        int mycol = 0, myeid = 0;
        if (lane < nb) {
          mycol = P.g.col[blk + lane];
          if constexpr (NEED_EID) myeid = P.g.eperm ? P.g.eperm[blk + lane] : blk + lane;
        }
        for (int s0 = 0; s0 < nb; s0 += G * U) {
          float gc[U][VEC], a1[U][VEC], oo[U][VEC], ea[U][VEC];
          int ai[U][VEC];
          bool ok[U];
          int eid[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int ei = s0 + u * G + g;
            ok[u] = ei < nb;
            const int dst = __shfl(mycol, ei & (kWave - 1));
            eid[u] = 0;
            if constexpr (NEED_EID) eid[u] = __shfl(myeid, ei & (kWave - 1));
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
              gc[u][j] = 0.f; a1[u][j] = 0.f; oo[u][j] = 0.f; ea[u][j] = 0.f; ai[u][j] = -1;
            }
            if (ok[u] && act) {
              const int64_t ro = static_cast<int64_t>(dst) * C + c0;
              if constexpr (MODE == kModeSoftmaxShifted) {
                load_vec<VEC>(gc[u], P.gshift + ro);
              } else {
                load_vec<VEC>(gc[u], P.gcoef + ro);
              }
              if constexpr (MODE == DGCN_AGGR_SOFTMAX) {
                load_vec<VEC>(a1[u], static_cast<const float*>(P.aux1) + ro);
                if (learn_t) load_vec<VEC>(oo[u], P.out + ro);
              }
              if constexpr (MODE == DGCN_AGGR_MAX) {
                load_vec_i<VEC>(ai[u], static_cast<const int32_t*>(P.aux1) + ro);
              }
              if constexpr (HAS_EA) {
                load_vec<VEC>(ea[u], P.ea + static_cast<int64_t>(eid[u]) * C + c0);
              }
            }
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
            float dz[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
              const float z = HAS_EA ? xs[j] + ea[u][j] : xs[j];
              const float m = msg_apply(z, msg, eps);
              const float r = (msg == DGCN_MSG_RELU_EPS) ? (z > 0.f ? 1.f : 0.f) : 1.f;
              float k;
              if constexpr (MODE == DGCN_AGGR_SOFTMAX) {
                float wgt = fast_exp(t * m - a1[u][j]);
                if (learn_t) wgt *= 1.f + t * (m - oo[u][j]);
                k = gc[u][j] * wgt;
              } else if constexpr (MODE == kModeSoftmaxShifted) {
                // g_i exp(t m - L_i) = [g_i exp(K_c - L_i)] * exp(t m - K_c): the bracket was gathered
                k = gc[u][j] * fast_exp(t * m - ksh[j]);
              } else if constexpr (MODE == DGCN_AGGR_POWER) {
                const bool in = (m >= kPowLo) && (m <= kPowHi);
                const float uu = fminf(fmaxf(m, kPowLo), kPowHi);
                k = in ? gc[u][j] * fast_pow(uu, p - 1.f) : 0.f;
              } else if constexpr (MODE == DGCN_AGGR_MAX) {
                k = (ai[u][j] == eid[u]) ? gc[u][j] : 0.f;
              } else {
                k = gc[u][j];
              }
              dz[j] = r * k;
              acc[j] += dz[j];
            }
            if constexpr (HAS_EA) {
              if (act && P.grad_ea) {
                store_vec<VEC>(P.grad_ea + static_cast<int64_t>(eid[u]) * C + c0, dz);
              }
            }
          }
        }
      }
#pragma unroll
      for (int off = LPR; off < kWave; off <<= 1) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] += __shfl_xor(acc[j], off);
      }
      if (g == 0 && act) {
        if (w.slot >= 0) {
          store_vec<VEC>(P.ws + static_cast<int64_t>(w.slot) * C + c0, acc);
        } else {
          store_vec<VEC>(P.grad_x + static_cast<int64_t>(w.row) * C + c0, acc);
        }
      }
    }
  }
}

template <int MODE, int VEC, int LPR, bool HAS_EA>
__global__ __launch_bounds__(kWgThreads) void gen_aggr_bwd_kernel(const BwdParams P) {
  if constexpr (MODE == DGCN_AGGR_SOFTMAX) {
    // single-gather form when the caller prepared it and the device-side range check passed
    if (P.gshift != nullptr && !P.learn_t && *P.shift_ok != 0) {
      gen_aggr_bwd_body<kModeSoftmaxShifted, VEC, LPR, HAS_EA>(P);
      return;
    }
  }
  gen_aggr_bwd_body<MODE, VEC, LPR, HAS_EA>(P);
}

// out[i,c] = g[i,c] * exp(kshift[c] - L[i,c])   (node-wise prologue of the single-gather backward)
__global__ __launch_bounds__(kWgThreads) void softmax_bwd_prep_kernel(const float* __restrict__ g,
                                                                      const float* __restrict__ L,
                                                                      const float* __restrict__ kshift,
                                                                      float* __restrict__ out, int64_t n_vec4,
                                                                      int c_vec4) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n_vec4; i += stride) {
    const int cv = static_cast<int>(i % c_vec4);
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    const float4 lv = reinterpret_cast<const float4*>(L)[i];
    const float4 kv = reinterpret_cast<const float4*>(kshift)[cv];
    float4 o;
    o.x = gv.x * fast_exp(kv.x - lv.x);
    o.y = gv.y * fast_exp(kv.y - lv.y);
    o.z = gv.z * fast_exp(kv.z - lv.z);
    o.w = gv.w * fast_exp(kv.w - lv.w);
    reinterpret_cast<float4*>(out)[i] = o;
  }
}

__global__ __launch_bounds__(kWgThreads) void gen_aggr_bwd_merge_kernel(const BwdParams P) {
  const int lane = lane_id();
  const int C = P.C;
  const int n_work = P.g.n_work;
  const int wave = blockIdx.x * kWavesPerWg + (threadIdx.x >> 6);
  if (wave >= P.g.n_split) return;
  const int i0 = uni(P.g.split_item[wave]);
  const int row = uni(P.g.work_row[i0]);
  int i1 = i0;
  while (i1 < n_work && uni(P.g.work_row[i1]) == row) ++i1;
  for (int c = lane; c < C; c += kWave) {
    float acc = 0.f;
    for (int i = i0; i < i1; ++i) acc += P.ws[static_cast<int64_t>(P.g.work_slot[i]) * C + c];
    P.grad_x[static_cast<int64_t>(row) * C + c] = acc;
  }
}

// ---------------------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------------------
int lanes_per_row(int C, int vec) {
  const int need = (C + vec - 1) / vec;
  int lpr = 8;
  while (lpr < need && lpr < kWave) lpr <<= 1;
  return lpr;
}

int round_up8(int v) { return (v + 7) / 8 * 8; }

template <int MODE, int VEC, int LPR>
void launch_fwd_ea(const FwdParams& P, int grid, hipStream_t s) {
  if (P.ea) {
    hipLaunchKernelGGL((gen_aggr_fwd_kernel<MODE, VEC, LPR, true>), dim3(grid), dim3(kWgThreads), 0, s, P);
  } else {
    hipLaunchKernelGGL((gen_aggr_fwd_kernel<MODE, VEC, LPR, false>), dim3(grid), dim3(kWgThreads), 0, s, P);
  }
}

template <int MODE>
void launch_fwd_mode(const FwdParams& P, int vec, int lpr, int grid, hipStream_t s) {
  if (vec == 4) {
    switch (lpr) {
      case 8: launch_fwd_ea<MODE, 4, 8>(P, grid, s); break;
      case 16: launch_fwd_ea<MODE, 4, 16>(P, grid, s); break;
      case 32: launch_fwd_ea<MODE, 4, 32>(P, grid, s); break;
      default: launch_fwd_ea<MODE, 4, 64>(P, grid, s); break;
    }
  } else {
    launch_fwd_ea<MODE, 1, 64>(P, grid, s);
  }
  if (P.g.n_work && P.g.n_split > 0) {
    const int mg = (P.g.n_split + kWavesPerWg - 1) / kWavesPerWg;
    hipLaunchKernelGGL((gen_aggr_fwd_merge_kernel<MODE>), dim3(mg), dim3(kWgThreads), 0, s, P);
  }
}

template <int MODE, int VEC, int LPR>
void launch_bwd_ea(const BwdParams& P, int grid, hipStream_t s) {
  if (P.ea) {
    hipLaunchKernelGGL((gen_aggr_bwd_kernel<MODE, VEC, LPR, true>), dim3(grid), dim3(kWgThreads), 0, s, P);
  } else {
    hipLaunchKernelGGL((gen_aggr_bwd_kernel<MODE, VEC, LPR, false>), dim3(grid), dim3(kWgThreads), 0, s, P);
  }
}

template <int MODE>
void launch_bwd_mode(const BwdParams& P, int vec, int lpr, int grid, hipStream_t s) {
  if (vec == 4) {
    switch (lpr) {
      case 8: launch_bwd_ea<MODE, 4, 8>(P, grid, s); break;
      case 16: launch_bwd_ea<MODE, 4, 16>(P, grid, s); break;
      case 32: launch_bwd_ea<MODE, 4, 32>(P, grid, s); break;
      default: launch_bwd_ea<MODE, 4, 64>(P, grid, s); break;
    }
  } else {
    launch_bwd_ea<MODE, 1, 64>(P, grid, s);
  }
  if (P.g.n_work && P.g.n_split > 0) {
    const int mg = (P.g.n_split + kWavesPerWg - 1) / kWavesPerWg;
    hipLaunchKernelGGL(gen_aggr_bwd_merge_kernel, dim3(mg), dim3(kWgThreads), 0, s, P);
  }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace
}  // namespace dgcn

using namespace dgcn;

extern "C" size_t dgcn_gen_aggr_fwd_workspace_bytes(const dgcn_graph* g, int32_t channels) {
  if (!g || g->n_work == 0) return 0;
  return static_cast<size_t>(g->n_slots) * 4u * static_cast<size_t>(channels) * sizeof(float);
}

extern "C" int dgcn_softmax_bwd_prep_f32(const float* g, const float* L, const float* kshift, float* out,
                                         int64_t n_rows, int32_t channels, void* stream) {
  if (!g || !L || !kshift || !out) return DGCN_E_NULL;
  if (n_rows < 0 || channels <= 0 || channels % 4 != 0) return DGCN_E_SHAPE;
  if (!aligned16(g) || !aligned16(L) || !aligned16(kshift) || !aligned16(out)) return DGCN_E_ALIGN;
  if (n_rows == 0) return DGCN_OK;
  const int64_t n4 = n_rows * (channels / 4);
  int64_t blocks = (n4 + kWgThreads - 1) / kWgThreads;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(softmax_bwd_prep_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kWgThreads), 0,
                     static_cast<hipStream_t>(stream), g, L, kshift, out, n4, channels / 4);
  return launch_status();
}

extern "C" size_t dgcn_gen_aggr_bwd_workspace_bytes(const dgcn_graph* g, int32_t channels) {
  if (!g || g->t_n_work == 0) return 0;
  return static_cast<size_t>(g->t_n_slots) * static_cast<size_t>(channels) * sizeof(float);
}

extern "C" int dgcn_gen_aggr_fwd_f32(const dgcn_graph* g, const float* x, int64_t x_stride,
                                     const float* edge_attr, int32_t channels, int32_t mode,
                                     int32_t msg, int32_t flags, float t, float p, float eps,
                                     const float* t_dev, const float* p_dev, float* out,
                                     void* aux1, float* aux2, int32_t* range_flag, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  (void)flags;
  if (!g || !x || !out) return DGCN_E_NULL;
  if (g->n_dst < 0 || g->n_edges < 0 || channels <= 0 || x_stride < channels) return DGCN_E_SHAPE;
  if (mode < DGCN_AGGR_ADD || mode > DGCN_AGGR_POWER) return DGCN_E_MODE;
  if (msg != DGCN_MSG_IDENTITY && msg != DGCN_MSG_RELU_EPS) return DGCN_E_MODE;
  if (g->n_dst == 0) return DGCN_OK;
  if (!g->rowptr || (g->n_edges > 0 && !g->col)) return DGCN_E_NULL;
  if (g->n_work && (!g->work_row || !g->work_beg || !g->work_end || !g->work_slot)) return DGCN_E_NULL;
  if (g->n_work && g->n_split > 0 && !g->split_item) return DGCN_E_NULL;
  if (workspace_bytes < dgcn_gen_aggr_fwd_workspace_bytes(g, channels)) return DGCN_E_WORKSPACE;
  if (g->n_work && g->n_slots > 0 && !workspace) return DGCN_E_NULL;

  const bool vec4 = (channels % 4 == 0) && (x_stride % 4 == 0) && aligned16(x) && aligned16(out) &&
                    (!edge_attr || aligned16(edge_attr)) && (!aux1 || aligned16(aux1)) &&
                    (!aux2 || aligned16(aux2)) && (!workspace || aligned16(workspace));
  const int vec = vec4 ? 4 : 1;
  const int lpr = vec4 ? lanes_per_row(channels, 4) : 64;

  FwdParams P;
  P.g = WalkGraph{g->n_dst, g->n_work, g->rowptr, g->col, g->eperm,
                  g->work_row, g->work_beg, g->work_end, g->work_slot, g->n_split, g->split_item};
  P.x = x; P.x_stride = x_stride; P.ea = edge_attr; P.C = channels; P.msg = msg;
  P.t = t; P.p = p; P.eps = eps; P.t_dev = t_dev; P.p_dev = p_dev;
  P.out = out; P.aux1 = aux1; P.aux2 = aux2; P.ws = static_cast<float*>(workspace);
  P.range_flag = (mode == DGCN_AGGR_SOFTMAX) ? range_flag : nullptr;

  const int n_items = g->n_work ? g->n_work : g->n_dst;
#ifdef DGCN_FWD_WAVES_PER_CU
  const int grid = round_up8(grid_for_waves(n_items, DGCN_FWD_WAVES_PER_CU));
#else
  const int grid = round_up8(grid_for_waves(n_items));
#endif
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (mode) {
    case DGCN_AGGR_ADD: launch_fwd_mode<DGCN_AGGR_ADD>(P, vec, lpr, grid, s); break;
    case DGCN_AGGR_MEAN: launch_fwd_mode<DGCN_AGGR_MEAN>(P, vec, lpr, grid, s); break;
    case DGCN_AGGR_MAX: launch_fwd_mode<DGCN_AGGR_MAX>(P, vec, lpr, grid, s); break;
    case DGCN_AGGR_SOFTMAX: launch_fwd_mode<DGCN_AGGR_SOFTMAX>(P, vec, lpr, grid, s); break;
    default: launch_fwd_mode<DGCN_AGGR_POWER>(P, vec, lpr, grid, s); break;
  }
  return launch_status();
}

extern "C" int dgcn_gen_aggr_bwd_f32(const dgcn_graph* g, const float* x, int64_t x_stride,
                                     const float* edge_attr, int32_t channels, int32_t mode,
                                     int32_t msg, int32_t flags, float t, float p, float eps,
                                     const float* t_dev, const float* p_dev, const float* gcoef,
                                     const void* aux1, const float* out, const float* gshift,
                                     const float* kshift, const int32_t* shift_ok, float* grad_x,
                                     float* grad_edge_attr, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  if (!g || !x || !gcoef || !grad_x) return DGCN_E_NULL;
  if (g->n_src < 0 || g->n_edges < 0 || channels <= 0 || x_stride < channels) return DGCN_E_SHAPE;
  if (mode < DGCN_AGGR_ADD || mode > DGCN_AGGR_POWER) return DGCN_E_MODE;
  if (msg != DGCN_MSG_IDENTITY && msg != DGCN_MSG_RELU_EPS) return DGCN_E_MODE;
  if ((mode == DGCN_AGGR_SOFTMAX || mode == DGCN_AGGR_MAX) && !aux1) return DGCN_E_NULL;
  if (mode == DGCN_AGGR_SOFTMAX && (flags & DGCN_FLAG_LEARN_T) && !out) return DGCN_E_NULL;
  if (g->n_src == 0) return DGCN_OK;
  if (!g->t_rowptr || (g->n_edges > 0 && (!g->t_col || !g->t_eperm))) return DGCN_E_NULL;
  if (g->t_n_work && (!g->t_work_row || !g->t_work_beg || !g->t_work_end || !g->t_work_slot)) return DGCN_E_NULL;
  if (g->t_n_work && g->t_n_split > 0 && !g->t_split_item) return DGCN_E_NULL;
  if (workspace_bytes < dgcn_gen_aggr_bwd_workspace_bytes(g, channels)) return DGCN_E_WORKSPACE;
  if (g->t_n_work && g->t_n_slots > 0 && !workspace) return DGCN_E_NULL;

  const bool vec4 = (channels % 4 == 0) && (x_stride % 4 == 0) && aligned16(x) && aligned16(gcoef) &&
                    aligned16(grad_x) && (!edge_attr || aligned16(edge_attr)) &&
                    (!aux1 || aligned16(aux1)) && (!out || aligned16(out)) &&
                    (!grad_edge_attr || aligned16(grad_edge_attr)) &&
                    (!workspace || aligned16(workspace));
  const int vec = vec4 ? 4 : 1;
  const int lpr = vec4 ? lanes_per_row(channels, 4) : 64;

  BwdParams P;
  P.g = WalkGraph{g->n_src, g->t_n_work, g->t_rowptr, g->t_col, g->t_eperm,
                  g->t_work_row, g->t_work_beg, g->t_work_end, g->t_work_slot, g->t_n_split, g->t_split_item};
  P.x = x; P.x_stride = x_stride; P.ea = edge_attr; P.C = channels; P.msg = msg;
  P.learn_t = (flags & DGCN_FLAG_LEARN_T) ? 1 : 0;
  P.t = t; P.p = p; P.eps = eps; P.t_dev = t_dev; P.p_dev = p_dev;
  P.gcoef = gcoef; P.aux1 = aux1; P.out = out; P.grad_x = grad_x; P.grad_ea = grad_edge_attr;
  P.gshift = nullptr; P.kshift = nullptr; P.shift_ok = nullptr;
  if (mode == DGCN_AGGR_SOFTMAX && gshift && kshift && shift_ok && vec4 && aligned16(gshift) && aligned16(kshift)) {
    P.gshift = gshift; P.kshift = kshift; P.shift_ok = shift_ok;
  }
  P.ws = static_cast<float*>(workspace);

  const int n_items = g->t_n_work ? g->t_n_work : g->n_src;
  const int grid = round_up8(grid_for_waves(n_items));
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (mode) {
    case DGCN_AGGR_ADD: launch_bwd_mode<DGCN_AGGR_ADD>(P, vec, lpr, grid, s); break;
    case DGCN_AGGR_MEAN: launch_bwd_mode<DGCN_AGGR_ADD>(P, vec, lpr, grid, s); break;  // gcoef pre-scaled
    case DGCN_AGGR_MAX: launch_bwd_mode<DGCN_AGGR_MAX>(P, vec, lpr, grid, s); break;
    case DGCN_AGGR_SOFTMAX: launch_bwd_mode<DGCN_AGGR_SOFTMAX>(P, vec, lpr, grid, s); break;
    default: launch_bwd_mode<DGCN_AGGR_POWER>(P, vec, lpr, grid, s); break;
  }
  return launch_status();
}
