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

#include "gen_aggr_common.h"
#include "gen_aggr_state.h"

extern "C" size_t dgcn_gen_aggr_fwd_workspace_bytes(const dgcn_graph* g, int32_t channels);

namespace dgcn {
namespace {

// Arg-max id as stored for the backward: with the relu message, m = relu(z) + eps equals eps exactly where no neighbour
// has z > 0 -- no edge receives a gradient there, and saying so in the id (-1) lets a backward run from the ids alone,
// without the pre-activations (dgcn_enc_max_bwd_weight_f32; the fused edge GEMM's forward marks the same way)
__device__ __forceinline__ int max_id_for_bwd(float a, int idx, int msg, float eps) {
  return (msg == DGCN_MSG_RELU_EPS && !(a > eps)) ? -1 : idx;
}


// ---------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------
template <int MODE, int VEC, int LPR, int SW, int EA, bool RELU, bool WITH_D>
__device__ __forceinline__ void gen_aggr_fwd_body(const FwdParams& P) {
  constexpr int G = SW / LPR;               // edges of one item walked in parallel
  constexpr int R = kWave / SW;             // items walked side by side by one wave
#ifdef DGCN_FWD_U
  constexpr int U = DGCN_FWD_U;
#else
  constexpr int U = (VEC == 4) ? 4 : 8;     // load batches in flight per lane
#endif
  constexpr bool NEED_EID = EA != 0 || MODE == DGCN_AGGR_MAX;

  const int lane = lane_id();
  const int sl = lane % SW;                 // lane within its sub-group
  const int sbase = lane - sl;
  const int g = sl / LPR;
  const int cl = sl % LPR;
  const int C = P.C;
  const uint32_t xs32 = static_cast<uint32_t>(P.x_stride);
  const int n_items = P.g.n_work ? P.g.n_work : P.g.n_rows;
  const int total_waves = gridDim.x * kWavesPerWg;
  const int wave0 = virtual_block() * kWavesPerWg + (threadIdx.x >> 6);
  const float t = P.t_dev ? *P.t_dev : P.t;
  const float p = P.p_dev ? *P.p_dev : P.p;
  const float eps = P.eps;
  const float eps_r = RELU ? eps : 0.f;     // the part of m = relu(z) + eps that the softmax fold leaves out
  const float t2 = t * 1.4426950408889634f; // log2 domain
  const float c0 = t2 * eps_r;

  // Software pipeline over the items of this wave: while item i is walked, the column ids of item i+1 and the
  // row bounds of item i+2 are already in flight, so a row does not start with two dependent memory latencies.
  const int stride = total_waves * R;
  const int sub = lane / SW;
  Work w = fetch_work<SW>(P.g, wave0 * R + sub, n_items);
  Work wn = fetch_work<SW>(P.g, wave0 * R + stride + sub, n_items);
  int col0, eid0;
  load_cols<SW, NEED_EID>(P.g, w, w.beg, sl, col0, eid0);
  for (int base = wave0 * R; base < n_items; base += stride) {
#ifndef DGCN_NO_PREFETCH
    int coln, eidn;
    load_cols<SW, NEED_EID>(P.g, wn, wn.beg, sl, coln, eidn);
    const Work wnn = fetch_work<SW>(P.g, base + 2 * stride + sub, n_items);
#endif
    for (int cb = 0; cb < C; cb += LPR * VEC) {
      const int c0ch = cb + cl * VEC;
      const bool act = c0ch < C;
      State<VEC> st;
      state_init<MODE, VEC>(st);
      EncW<(EA == 2 ? VEC : 1)> enc;
      if constexpr (EA == 2) enc_load<VEC>(enc, P.enc_w, P.enc_b, c0ch, act);

      int mycol = col0, myeid = eid0;
      for (int blk = w.beg; any_sub<SW>(blk < w.end); blk += SW) {
        const int nb = max(0, min(SW, w.end - blk));
        if (blk != w.beg) load_cols<SW, NEED_EID>(P.g, w, blk, sl, mycol, myeid);
        for (int s0 = 0; any_sub<SW>(s0 < nb); s0 += G * U) {
          float v[U][VEC];
          bool ok[U];
          int eid[U];
          const bool full = all_sub<SW>(nb - s0 >= G * U) && (cb + LPR * VEC <= C);   // wave-uniform
          if (full) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
              const int ei = s0 + u * G + g;
              ok[u] = true;
              const int src = __shfl(mycol, sbase + (ei & (SW - 1)));
              eid[u] = 0;
              if constexpr (NEED_EID) eid[u] = __shfl(myeid, sbase + (ei & (SW - 1)));
              load_vec<VEC>(v[u], row_ptr(P.x, src, xs32) + c0ch);
              if constexpr (EA == 1) {
                float a[VEC];
                load_vec<VEC>(a, P.ea + static_cast<int64_t>(eid[u]) * C + c0ch);
#pragma unroll
                for (int j = 0; j < VEC; ++j) v[u][j] += a[j];
              }
              if constexpr (EA == 2) {
                float fe[kEncF], a[VEC];
                enc_feat_row(fe, P.enc_feat, eid[u]);
                enc_apply<VEC>(a, enc, fe);
#pragma unroll
                for (int j = 0; j < VEC; ++j) v[u][j] += a[j];
              }
            }
            accumulate<MODE, VEC, U, RELU, WITH_D, true>(st, v, ok, eid, eps, t2, c0, p);
          } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
              const int ei = s0 + u * G + g;
              ok[u] = ei < nb;
              const int src = __shfl(mycol, sbase + (ei & (SW - 1)));
              eid[u] = 0;
              if constexpr (NEED_EID) eid[u] = __shfl(myeid, sbase + (ei & (SW - 1)));
#pragma unroll
              for (int j = 0; j < VEC; ++j) v[u][j] = 0.f;
              if (ok[u] && act) {
                load_vec<VEC>(v[u], row_ptr(P.x, src, xs32) + c0ch);
                if constexpr (EA == 1) {
                  float a[VEC];
                  load_vec<VEC>(a, P.ea + static_cast<int64_t>(eid[u]) * C + c0ch);
#pragma unroll
                  for (int j = 0; j < VEC; ++j) v[u][j] += a[j];
                }
                if constexpr (EA == 2) {
                  float fe[kEncF], a[VEC];
                  enc_feat_row(fe, P.enc_feat, eid[u]);
                  enc_apply<VEC>(a, enc, fe);
#pragma unroll
                  for (int j = 0; j < VEC; ++j) v[u][j] += a[j];
                }
              }
            }
            accumulate<MODE, VEC, U, RELU, WITH_D, false>(st, v, ok, eid, eps, t2, c0, p);
          }
        }
      }

      // combine the G edge groups of each item (all lanes participate)
#pragma unroll
      for (int off = LPR; off < SW; off <<= 1) {
        const State<VEC> o = state_shfl_xor<MODE, VEC>(st, off);
        state_merge<MODE, VEC>(st, o);
      }

      if (g == 0 && act && w.row >= 0) {
        if constexpr (MODE == DGCN_AGGR_SOFTMAX) {
          // back to sums over m = r + eps:  sum e m = A + eps D,  sum e m^2 = A2 + 2 eps A + eps^2 D
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            if constexpr (WITH_D) st.d[j] = fmaf(eps_r, fmaf(eps_r, st.b[j], 2.f * st.c[j]), st.d[j]);
            st.c[j] = fmaf(eps_r, st.b[j], st.c[j]);
          }
        }
        if (w.slot >= 0) {
          float* ws = P.ws + (static_cast<int64_t>(w.slot) * 4) * C + c0ch;
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
          const int64_t o = static_cast<int64_t>(w.row) * C + c0ch;
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
              x1[j] = any ? (st.a[j] + fast_log2(st.b[j])) * 0.6931471805599453f : 0.f;
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
              xi[j] = max_id_for_bwd(st.a[j], st.idx[j], P.msg, P.eps);
            } else if constexpr (MODE == DGCN_AGGR_MEAN) {
              res[j] = st.b[j] / fmaxf(deg, 1.f);
            } else {
              res[j] = st.b[j];
            }
          }
          if (P.add_root) {
            float xr[VEC];
            load_vec<VEC>(xr, row_ptr(P.x, w.row, xs32) + c0ch);
#pragma unroll
            for (int j = 0; j < VEC; ++j) res[j] += xr[j];
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
#ifndef DGCN_NO_PREFETCH
    w = wn;
    wn = wnn;
    col0 = coln;
    eid0 = eidn;
#else
    w = fetch_work<SW>(P.g, base + stride + sub, n_items);
    load_cols<SW, NEED_EID>(P.g, w, w.beg, sl, col0, eid0);
#endif
  }
}

// WITH_D (second moment for learnable t / p) is a kernel-level parameter: its extra accumulators must not
// cost the common variant registers.  The message kind is a wave-uniform branch inside (same register budget).
#ifdef DGCN_FWD_WPE
#define DGCN_FWD_OCC __attribute__((amdgpu_waves_per_eu(DGCN_FWD_WPE, DGCN_FWD_WPE)))
#else
#define DGCN_FWD_OCC
#endif
template <int MODE, int VEC, int LPR, int SW, int EA, bool WITH_D>
__global__ __launch_bounds__(kWgThreads) DGCN_FWD_OCC void gen_aggr_fwd_kernel(const FwdParams P) {
  if (P.msg == DGCN_MSG_RELU_EPS) {
    gen_aggr_fwd_body<MODE, VEC, LPR, SW, EA, true, WITH_D>(P);
  } else {
    gen_aggr_fwd_body<MODE, VEC, LPR, SW, EA, false, WITH_D>(P);
  }
}

// ---------------------------------------------------------------------------------------
// forward with the per-edge encoder (EA == 2), wave-uniform walk
// ---------------------------------------------------------------------------------------
// blocks.ComposedEdgeEmbedding hands every GENConv of the reversible ogbn-proteins models 8 raw features per edge and
// a composed Linear(8 -> C) (C = hidden / group = 112 at the reference's width): per edge 32 bytes of features, one
// gathered 4C-byte row of x, C x 8 multiply-adds.  The general row walk above spends 164 registers on it (three waves
// per SIMD), keeps eight edges in flight per wave and alternates load and arithmetic phases: 69 % of its wave cycles
// wait (profiles/r03_revgcn8_composed_kernel_breakdown.md).  Here ONE row (or hub piece) per wave, all 64 lanes on
// the channels of one edge at a time (VEC = 2 channels per lane up to C = 128, 4 up to 256), so that everything that
// depends on the edge alone is wave-uniform:
//   * 64 edges are staged per block: lane l fetches the column id, the edge id and the 32-byte feature row of edge
//     blk + l (one coalesced and one 32-byte gathered load per lane) and parks the features in LDS; an edge's features
//     then reach all lanes as two broadcast ds_read_b128 right before they are used -- no registers held across the
//     memory wait, no per-lane copy of ids (v_readlane with a scalar index);
//   * the x rows of the NEXT batch of eight edges are requested before the current batch is folded (two register
//     sets), and the next block's staging loads are in flight during the current block;
//   * the encoder's weights are loaded once per wave (the lane's channels never change), not once per row.
// ~90 registers: five or six waves per SIMD, 16 row loads in flight each.
constexpr int kEncBlk = kWave;

template <int MODE, int VEC, bool WITH_D>
__device__ __forceinline__ void enc_fwd_finish(const FwdParams& P, const Work& w, State<VEC>& st, int c0ch, float eps_r,
                                               float p, uint32_t xs32) {
  const int C = P.C;
  if constexpr (MODE == DGCN_AGGR_SOFTMAX) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      if constexpr (WITH_D) st.d[j] = fmaf(eps_r, fmaf(eps_r, st.b[j], 2.f * st.c[j]), st.d[j]);
      st.c[j] = fmaf(eps_r, st.b[j], st.c[j]);
    }
  }
  if (w.slot >= 0) {
    float* ws = P.ws + (static_cast<int64_t>(w.slot) * 4) * C + c0ch;
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
    return;
  }
  const int64_t o = static_cast<int64_t>(w.row) * C + c0ch;
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
      x1[j] = any ? (st.a[j] + fast_log2(st.b[j])) * 0.6931471805599453f : 0.f;
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
      xi[j] = max_id_for_bwd(st.a[j], st.idx[j], P.msg, P.eps);
    } else if constexpr (MODE == DGCN_AGGR_MEAN) {
      res[j] = st.b[j] / fmaxf(deg, 1.f);
    } else {
      res[j] = st.b[j];
    }
  }
  if (P.add_root) {
    float xr[VEC];
    load_vec<VEC>(xr, row_ptr(P.x, w.row, xs32) + c0ch);
#pragma unroll
    for (int j = 0; j < VEC; ++j) res[j] += xr[j];
  }
  store_vec<VEC>(P.out + o, res);
  if constexpr (MODE == DGCN_AGGR_MAX) {
    if (P.aux1) store_vec_i<VEC>(static_cast<int32_t*>(P.aux1) + o, xi);
  } else if constexpr (MODE == DGCN_AGGR_SOFTMAX || MODE == DGCN_AGGR_POWER) {
    if (P.aux1) store_vec<VEC>(static_cast<float*>(P.aux1) + o, x1);
    if (P.aux2) store_vec<VEC>(P.aux2 + o, x2);
  }
}

template <int MODE, int VEC, bool RELU, bool WITH_D>
__device__ __forceinline__ void enc_fwd_body(const FwdParams& P) {
  constexpr int U = 8;                       // edges per batch (two batches of row loads in flight)
  __shared__ __attribute__((aligned(16))) float sfeat[kWavesPerWg][kEncBlk * kEncF];
  const int lane = lane_id();
  const int wv = threadIdx.x >> 6;
  const int C = P.C;
  const int c0ch = lane * VEC;
  const bool act = c0ch < C;
  const uint32_t xs32 = static_cast<uint32_t>(P.x_stride);
  const int n_items = P.g.n_work ? P.g.n_work : P.g.n_rows;
  const float t = P.t_dev ? *P.t_dev : P.t;
  const float p = P.p_dev ? *P.p_dev : P.p;
  const float eps = P.eps;
  const float eps_r = RELU ? eps : 0.f;
  const float t2 = t * 1.4426950408889634f;
  const float c0 = t2 * eps_r;
  EncW<VEC> enc;
  enc_load<VEC>(enc, P.enc_w, P.enc_b, c0ch, act);

  // staging loads of one 64-edge block: this lane's edge
  struct Stage { int col, eid; float4 f0, f1; };
  auto stage_load = [&](const Work& w, int blk) -> Stage {
    Stage s;
    s.col = 0; s.eid = 0;
    s.f0 = make_float4(0.f, 0.f, 0.f, 0.f); s.f1 = s.f0;
    if (lane < w.end - blk) {
      s.col = P.g.col[blk + lane];
      s.eid = P.g.eperm ? P.g.eperm[blk + lane] : blk + lane;
      const float4* fp = reinterpret_cast<const float4*>(P.enc_feat + static_cast<int64_t>(s.eid) * kEncF);
      s.f0 = fp[0]; s.f1 = fp[1];
    }
    return s;
  };

  float* sf = sfeat[wv];
  ItemQueue q;
  int item = q.first(P.ticket);
  Work w = fetch_work<kWave>(P.g, item, n_items);
  Stage sg = stage_load(w, w.beg);
  while (item < n_items) {
    const int next = q.next(P.ticket);
    const Work wn = fetch_work<kWave>(P.g, next, n_items);
    State<VEC> st;
    state_init<MODE, VEC>(st);
    for (int blk = w.beg; blk < w.end || blk == w.beg; blk += kEncBlk) {
      const int nb = max(0, min(kEncBlk, w.end - blk));
      // park this block's features (a wave's LDS operations complete in order: the previous block's reads are done),
      // request the next block's -- or the next item's first block
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      *reinterpret_cast<float4*>(sf + lane * kEncF) = sg.f0;
      *reinterpret_cast<float4*>(sf + lane * kEncF + 4) = sg.f1;
      const int mycol = sg.col, myeid = sg.eid;
      const bool last_blk = blk + kEncBlk >= w.end;
      sg = last_blk ? stage_load(wn, wn.beg) : stage_load(w, blk + kEncBlk);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();

      float va[U][VEC], vb[U][VEC];
      auto load_batch = [&](float (&v)[U][VEC], int s0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) v[u][j] = 0.f;
          if (s0 + u < nb) {                                     // wave-uniform
            const int src = __builtin_amdgcn_readlane(mycol, s0 + u);
            if (act) load_vec<VEC>(v[u], row_ptr(P.x, src, xs32) + c0ch);
          }
        }
      };
      auto fold_batch = [&](float (&v)[U][VEC], int s0) {
        bool ok[U];
        int eid[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          ok[u] = s0 + u < nb;
          eid[u] = 0;
          if (ok[u]) {
            if constexpr (MODE == DGCN_AGGR_MAX) eid[u] = __builtin_amdgcn_readlane(myeid, s0 + u);
            const float4 a = *reinterpret_cast<const float4*>(sf + (s0 + u) * kEncF);       // broadcast reads
            const float4 b = *reinterpret_cast<const float4*>(sf + (s0 + u) * kEncF + 4);
            const float fe[kEncF] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            float e[VEC];
            enc_apply<VEC>(e, enc, fe);
#pragma unroll
            for (int j = 0; j < VEC; ++j) v[u][j] += e[j];
          }
        }
        if (s0 + U <= nb) accumulate<MODE, VEC, U, RELU, WITH_D, true>(st, v, ok, eid, eps, t2, c0, p);
        else accumulate<MODE, VEC, U, RELU, WITH_D, false>(st, v, ok, eid, eps, t2, c0, p);
      };
      if (nb > 0) {
        load_batch(va, 0);
        for (int s0 = 0; s0 < nb; s0 += 2 * U) {
          if (s0 + U < nb) load_batch(vb, s0 + U);
          fold_batch(va, s0);
          if (s0 + U < nb) {
            if (s0 + 2 * U < nb) load_batch(va, s0 + 2 * U);
            fold_batch(vb, s0 + U);
          }
        }
      }
      if (last_blk) break;
    }
    if (act && w.row >= 0) enc_fwd_finish<MODE, VEC, WITH_D>(P, w, st, c0ch, eps_r, p, xs32);
    w = wn;
    item = next;
  }
}

template <int MODE, int VEC, bool WITH_D>
__global__ __launch_bounds__(kWgThreads) void gen_aggr_enc_fwd_kernel(const FwdParams P) {
  if (P.msg == DGCN_MSG_RELU_EPS) {
    enc_fwd_body<MODE, VEC, true, WITH_D>(P);
  } else {
    enc_fwd_body<MODE, VEC, false, WITH_D>(P);
  }
}

template <int MODE, int VEC>
void launch_enc_fwd(const FwdParams& P, int grid, hipStream_t s) {
  constexpr bool CAN_D = MODE == DGCN_AGGR_SOFTMAX || MODE == DGCN_AGGR_POWER;
  if constexpr (CAN_D) {
    if (P.aux2) {
      hipLaunchKernelGGL((gen_aggr_enc_fwd_kernel<MODE, VEC, true>), dim3(grid), dim3(kWgThreads), 0, s, P);
      return;
    }
  }
  hipLaunchKernelGGL((gen_aggr_enc_fwd_kernel<MODE, VEC, false>), dim3(grid), dim3(kWgThreads), 0, s, P);
}

// Merge the partial slots of split (hub) rows.  One wave per (split row, block of 16 channels): the lanes are four
// piece groups x 16 channels, every group folds a contiguous quarter of the row's pieces (a 11,000-edge row of a
// small graph is 179 pieces: 3 batches of loads per lane instead of 12 behind one another), then the four partial
// states meet through two shuffles and group 0 writes.  The pieces of a row own CONSECUTIVE items and slots
// (graph_build.hip work_fill_kernel), so nothing here depends on a loaded index; the fold order is fixed ->
// deterministic.
constexpr int kMergeCh = 16;                       // channels per merge wave
constexpr int kMergeGroups = kWave / kMergeCh;     // piece groups per merge wave
template <int MODE>
__global__ __launch_bounds__(kWgThreads) void gen_aggr_fwd_merge_kernel(const FwdParams P) {
  const int lane = lane_id();
  const int C = P.C;
  const int cblocks = (C + kMergeCh - 1) / kMergeCh;
  const int wave = blockIdx.x * kWavesPerWg + (threadIdx.x >> 6);
  if (wave >= P.g.n_split * cblocks) return;
  const float p = P.p_dev ? *P.p_dev : P.p;
  const int i0 = uni(P.g.split_item[wave / cblocks]);
  const int row = uni(P.g.work_row[i0]);
  const int slot0 = uni(P.g.work_slot[i0]);
  const int rbeg = uni(P.g.rowptr[row]), rend = uni(P.g.rowptr[row + 1]);
  const float deg = static_cast<float>(rend - rbeg);
  const int chunk = uni(P.g.work_end[i0]) - uni(P.g.work_beg[i0]);          // every piece but the last is this long
  const int npieces = (rend - rbeg + chunk - 1) / chunk;
  {
    const int c_raw = (wave % cblocks) * kMergeCh + (lane & (kMergeCh - 1));
    const int c = min(c_raw, C - 1);               // every lane stays in the shuffles; lanes past C do not write
    const int grp = lane / kMergeCh;
    const int per = (npieces + kMergeGroups - 1) / kMergeGroups;
    const int pbeg = min(grp * per, npieces), pend = min(pbeg + per, npieces);
    State<1> st;
    state_init<MODE, 1>(st);
    constexpr int NQ = (MODE == DGCN_AGGR_MAX) ? 2 : 4;
    constexpr int PB = (MODE == DGCN_AGGR_MAX) ? 16 : 8;       // pieces in flight (32 loads)
    for (int i = pbeg; i < pend; i += PB) {
      float v[PB][4];
#pragma unroll
      for (int k = 0; k < PB; ++k) {
        const float* ws = P.ws + (static_cast<int64_t>(slot0 + min(i + k, pend - 1)) * 4) * C + c;
#pragma unroll
        for (int q = 0; q < NQ; ++q) v[k][q] = ws[static_cast<int64_t>(q) * C];
      }
#pragma unroll
      for (int k = 0; k < PB; ++k) {
        if (i + k < pend) {
          State<1> o;
          state_init<MODE, 1>(o);
          o.a[0] = v[k][0];
          if constexpr (MODE == DGCN_AGGR_MAX) {
            o.idx[0] = __float_as_int(v[k][1]);
          } else {
            o.b[0] = v[k][1]; o.c[0] = v[k][2]; o.d[0] = v[k][3];
          }
          state_merge<MODE, 1>(st, o);
        }
      }
    }
#pragma unroll
    for (int off = kMergeCh; off < kWave; off <<= 1) {
      const State<1> o = state_shfl_xor<MODE, 1>(st, off);
      state_merge<MODE, 1>(st, o);
    }
    if (grp != 0 || c_raw >= C) return;
    const int64_t o = static_cast<int64_t>(row) * C + c;
    float res, x1 = 0.f, x2 = 0.f;
    if constexpr (MODE == DGCN_AGGR_SOFTMAX) {
      const bool any = st.b[0] > 0.f;
      const float inv = any ? 1.f / st.b[0] : 0.f;
      res = st.c[0] * inv;
      x1 = any ? (st.a[0] + fast_log2(st.b[0])) * 0.6931471805599453f : 0.f;
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
    P.out[o] = P.add_root ? res + P.x[static_cast<int64_t>(row) * P.x_stride + c] : res;
    if constexpr (MODE == DGCN_AGGR_MAX) {
      if (P.aux1) static_cast<int32_t*>(P.aux1)[o] = max_id_for_bwd(st.a[0], st.idx[0], P.msg, P.eps);
    } else if constexpr (MODE == DGCN_AGGR_SOFTMAX || MODE == DGCN_AGGR_POWER) {
      if (P.aux1) static_cast<float*>(P.aux1)[o] = x1;
      if (P.aux2) P.aux2[o] = x2;
    }
  }
}

template <int MODE, int VEC, int LPR, int SW, int EA>
void launch_fwd_d(const FwdParams& P, int grid, hipStream_t s) {
  constexpr bool CAN_D = MODE == DGCN_AGGR_SOFTMAX || MODE == DGCN_AGGR_POWER;
  if constexpr (CAN_D) {
    if (P.aux2) {
      hipLaunchKernelGGL((gen_aggr_fwd_kernel<MODE, VEC, LPR, SW, EA, true>), dim3(grid), dim3(kWgThreads), 0, s, P);
      return;
    }
  }
  hipLaunchKernelGGL((gen_aggr_fwd_kernel<MODE, VEC, LPR, SW, EA, false>), dim3(grid), dim3(kWgThreads), 0, s, P);
}

template <int MODE, int VEC, int LPR, int SW>
void launch_fwd_ea(const FwdParams& P, int grid, hipStream_t s) {
  if constexpr (VEC == 4 && LPR <= 16) {   // C < 64 (wider rows take gen_aggr_enc_fwd_kernel, enc_uniform_walk)
    if (P.enc_feat) {   // fused edge encoder: float4 layouts only (the host entry point checks)
      launch_fwd_d<MODE, VEC, LPR, SW, 2>(P, grid, s);
      return;
    }
  }
  if (P.ea) {
    launch_fwd_d<MODE, VEC, LPR, SW, 1>(P, grid, s);
  } else {
    launch_fwd_d<MODE, VEC, LPR, SW, 0>(P, grid, s);
  }
}

// the wave-uniform encoder walk serves rows of 64 .. 256 channels (at least half of the lanes busy)
inline bool enc_uniform_walk(const FwdParams& P, int vec) { return vec == 4 && P.enc_feat && P.C >= 64 && P.C <= 256; }

template <int MODE>
void launch_fwd_mode(const FwdParams& P, int vec, int lpr, int grid, hipStream_t s) {
  if (enc_uniform_walk(P, vec)) {
    // persistent grid: four waves per SIMD (the kernels' register budget), items claimed from the ticket counter
    const int n_items = P.g.n_work ? P.g.n_work : P.g.n_rows;
    const int pgrid = min((n_items + kWavesPerWg - 1) / kWavesPerWg, num_cus() * 4);
    if (P.C <= 128) launch_enc_fwd<MODE, 2>(P, pgrid, s); else launch_enc_fwd<MODE, 4>(P, pgrid, s);
  } else if (vec == 4) {
    // (LPR, SW) pairs: SW = LPR * edge groups per row (kEdgeGroups) when the graph has enough rows to fill the
    // chip that way, else one row per wave (more, shorter waves)
    const int sw_sel = subgroup_width(lpr, P.g.n_work ? P.g.n_work : P.g.n_rows, P.n_edges_hint);
    const bool one = sw_sel == kWave;
    switch (lpr) {
      case 4: one ? launch_fwd_ea<MODE, 4, 4, 64>(P, grid, s) : launch_fwd_ea<MODE, 4, 4, kSubWidth(4)>(P, grid, s); break;
      case 8: one ? launch_fwd_ea<MODE, 4, 8, 64>(P, grid, s) : launch_fwd_ea<MODE, 4, 8, kSubWidth(8)>(P, grid, s); break;
      case 16: one ? launch_fwd_ea<MODE, 4, 16, 64>(P, grid, s) : launch_fwd_ea<MODE, 4, 16, kSubWidth(16)>(P, grid, s); break;
      case 32: (sw_sel == 32) ? launch_fwd_ea<MODE, 4, 32, 32>(P, grid, s) : launch_fwd_ea<MODE, 4, 32, 64>(P, grid, s); break;
      default: launch_fwd_ea<MODE, 4, 64, 64>(P, grid, s); break;
    }
  } else {
    launch_fwd_ea<MODE, 1, 64, 64>(P, grid, s);
  }
  if (P.g.n_work && P.g.n_split > 0) {
    const int mwaves = P.g.n_split * ((P.C + kMergeCh - 1) / kMergeCh);
    const int mg = (mwaves + kWavesPerWg - 1) / kWavesPerWg;
    hipLaunchKernelGGL((gen_aggr_fwd_merge_kernel<MODE>), dim3(mg), dim3(kWgThreads), 0, s, P);
  }
}


int gen_aggr_fwd_impl(const dgcn_graph* g, const float* x, int64_t x_stride,
                      const float* edge_attr, const EncArgs* enc, int32_t channels, int32_t mode,
                      int32_t msg, int32_t flags, float t, float p, float eps,
                      const float* t_dev, const float* p_dev, float* out,
                      void* aux1, float* aux2, int32_t* range_flag, void* workspace,
                      size_t workspace_bytes, void* stream) {
  if (!g || !x || !out) return DGCN_E_NULL;
  if (const int rc = enc_check(enc, channels)) return rc;
  if ((flags & DGCN_FLAG_ADD_ROOT) && g->n_dst > g->n_src) return DGCN_E_SHAPE;   // root rows are x[0 .. n_dst)
  if (g->n_dst < 0 || g->n_edges < 0 || channels <= 0 || x_stride < channels) return DGCN_E_SHAPE;
  if (x_stride > 0x7fffffffLL) return DGCN_E_SHAPE;   // row addresses use a 32x32->64 multiply
  if (mode < DGCN_AGGR_ADD || mode > DGCN_AGGR_POWER) return DGCN_E_MODE;
  if (msg != DGCN_MSG_IDENTITY && msg != DGCN_MSG_RELU_EPS) return DGCN_E_MODE;
  if (g->n_dst == 0) return DGCN_OK;
  if (!g->rowptr || (g->n_edges > 0 && !g->col)) return DGCN_E_NULL;
  if (g->n_work && (!g->work_row || !g->work_beg || !g->work_end || !g->work_slot)) return DGCN_E_NULL;
  if (g->n_work && g->n_split > 0 && !g->split_item) return DGCN_E_NULL;
  {
    // split-row slots; the scheduling state behind them is needed by the encoder walk only (a caller of the plain
    // entry point on a graph without split rows may pass no workspace at all)
    const size_t full = dgcn_gen_aggr_fwd_workspace_bytes(g, channels);
    const bool walk = enc && channels % 4 == 0 && channels >= 64 && channels <= 256;
    const size_t slots = g->n_work ? static_cast<size_t>(g->n_slots) * 4u * static_cast<size_t>(channels) * sizeof(float) : 0;
    if (workspace_bytes < (walk ? full : slots)) return DGCN_E_WORKSPACE;
  }
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
  P.add_root = (flags & DGCN_FLAG_ADD_ROOT) ? 1 : 0;
  P.enc_feat = enc ? enc->feat : nullptr;
  P.enc_w = enc ? enc->w : nullptr;
  P.enc_b = enc ? enc->b : nullptr;
  P.n_edges_hint = g->n_edges;
  if (enc && !vec4) return DGCN_E_ALIGN;
  P.ticket = nullptr;
  if (enc_uniform_walk(P, vec) && !(flags & DGCN_FLAG_STATIC_ITEMS)) {
    if (!workspace) return DGCN_E_NULL;
    P.ticket = reinterpret_cast<int32_t*>(static_cast<char*>(workspace) +
                                          dgcn_gen_aggr_fwd_workspace_bytes(g, channels) - kTicketBytes);
    if (const int zrc = zero_async(P.ticket, kTicketBytes, static_cast<hipStream_t>(stream))) return zrc;   // a kernel, not a memset node: dgcn_common.h
  }

  const int per_wave = vec4 ? kWave / subgroup_width(lpr, (g->n_work ? g->n_work : g->n_dst), g->n_edges) : 1;   // items walked side by side by one wave
  const int n_items = ((g->n_work ? g->n_work : g->n_dst) + per_wave - 1) / per_wave;
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

}  // namespace
}  // namespace dgcn

using namespace dgcn;

extern "C" size_t dgcn_gen_aggr_fwd_workspace_bytes(const dgcn_graph* g, int32_t channels) {
  if (!g) return 0;
  // partial-state slots of the split rows + the work-item counter of the encoder walk (last kTicketBytes)
  const size_t slots = g->n_work ? static_cast<size_t>(g->n_slots) * 4u * static_cast<size_t>(channels) * sizeof(float) : 0;
  return (slots + 255u) / 256u * 256u + kTicketBytes;
}


extern "C" int dgcn_gen_aggr_fwd_f32(const dgcn_graph* g, const float* x, int64_t x_stride,
                                     const float* edge_attr, int32_t channels, int32_t mode,
                                     int32_t msg, int32_t flags, float t, float p, float eps,
                                     const float* t_dev, const float* p_dev, float* out,
                                     void* aux1, float* aux2, int32_t* range_flag, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  return gen_aggr_fwd_impl(g, x, x_stride, edge_attr, nullptr, channels, mode, msg, flags, t, p, eps, t_dev, p_dev,
                           out, aux1, aux2, range_flag, workspace, workspace_bytes, stream);
}

extern "C" int dgcn_gen_aggr_enc_fwd_f32(const dgcn_graph* g, const float* x, int64_t x_stride,
                                         const float* enc_feat, const float* enc_weight, const float* enc_bias,
                                         int32_t n_feat, int32_t channels, int32_t mode, int32_t msg,
                                         int32_t flags, float t, float p, float eps, const float* t_dev,
                                         const float* p_dev, float* out, void* aux1, float* aux2,
                                         int32_t* range_flag, void* workspace, size_t workspace_bytes,
                                         void* stream) {
  const EncArgs enc{enc_feat, enc_weight, enc_bias, n_feat};
  return gen_aggr_fwd_impl(g, x, x_stride, nullptr, &enc, channels, mode, msg, flags, t, p, eps, t_dev, p_dev, out,
                           aux1, aux2, range_flag, workspace, workspace_bytes, stream);
}

