// Sparse generalized aggregation for gfx950 (MI355X): backward (see gen_aggr_fwd.hip for the execution shape).
//
// Walks the CSC (rows = sources) with the same one-wave-per-item scheme and produces grad_x (and grad_edge_attr,
// or the fused edge encoder's dW | db partials) in one deterministic pass; softmax takes the single-gather form
// prepared by softmax_bwd_prep_kernel when the forward's range flag allows it.

#include "gen_aggr_common.h"

extern "C" size_t dgcn_gen_aggr_bwd_workspace_bytes(const dgcn_graph* g, int32_t channels);

namespace dgcn {
namespace {

// ---------------------------------------------------------------------------------------
// backward: walk the CSC (rows = sources).  For CSC position e with destination i and
// original edge id oe:   dz_e = R(z_e) * K(m_e, i)      
// ---------------------------------------------------------------------------------------
constexpr int kModeSoftmaxShifted = 100;  // internal: softmax backward with ONE gathered row per edge

template <int MODE, int VEC, int LPR, int SW, int EA>
__device__ __forceinline__ void gen_aggr_bwd_body(const BwdParams& P) {
  constexpr int G = SW / LPR;
  constexpr int R = kWave / SW;
  constexpr int U = (EA == 2) ? 2 : ((VEC == 4) ? 4 : 8);   // EA == 2 keeps U feature rows + 9 VEC sums live
  constexpr bool NEED_EID = EA != 0 || MODE == DGCN_AGGR_MAX;
  constexpr int EV = (EA == 2) ? VEC : 1;
  EncW<EV> enc, genc;       // encoder weights of this lane's channels, and the sums dW | db over this wave's edges
#pragma unroll
  for (int j = 0; j < EV; ++j) {
    genc.b[j] = 0.f;
#pragma unroll
    for (int f = 0; f < kEncF; ++f) genc.w[j][f] = 0.f;
  }

  const int lane = lane_id();
  const int sl = lane % SW;
  const int sbase = lane - sl;
  const int g = sl / LPR;
  const int cl = sl % LPR;
  const int C = P.C;
  const int n_items = P.g.n_work ? P.g.n_work : P.g.n_rows;
  const int total_waves = gridDim.x * kWavesPerWg;
  const int wave0 = virtual_block() * kWavesPerWg + (threadIdx.x >> 6);
  const float t = P.t_dev ? *P.t_dev : P.t;
  const float p = P.p_dev ? *P.p_dev : P.p;
  const float eps = P.eps;
  const int msg = P.msg;
  const bool learn_t = P.learn_t != 0;
  const bool p_is_one = p == 1.f;          // the reference's default exponent: u^(p-1) = 1, no log2 / exp2 per element
  const bool ea_is_z = (EA == 1) && P.ea_is_z != 0;
  // MAX without edge rows: per-edge arg-max bit masks (<= 4 words) travel with the column ids -- loaded one item ahead,
  // parked in LDS per block, read back per edge with one ds_read instead of a dependent global gather
  constexpr bool MASKP = (MODE == DGCN_AGGR_MAX) && (EA == 0);
  __shared__ uint32_t smask[MASKP ? kWavesPerWg * kWave * 4 : 1];
  const bool mask_lds = MASKP && P.maxmask != nullptr && P.mask_words <= 4;
  uint32_t* my_smask = smask + (MASKP ? (threadIdx.x >> 6) * kWave * 4 : 0);
  auto load_mask = [&](const Work& ww, int blk, int cpos, uint32_t (&mk)[4]) {
    mk[0] = mk[1] = mk[2] = mk[3] = 0u;
    if (mask_lds && sl < ww.end - blk) {
      const uint32_t* mp = P.maxmask + static_cast<int64_t>(cpos) * P.mask_words;
      if (P.mask_words == 4) {
        const uint4 v = *reinterpret_cast<const uint4*>(mp);
        mk[0] = v.x; mk[1] = v.y; mk[2] = v.z; mk[3] = v.w;
      } else if (P.mask_words == 2) {
        const uint2 v = *reinterpret_cast<const uint2*>(mp);
        mk[0] = v.x; mk[1] = v.y;
      } else {
        mk[0] = mp[0];
      }
    }
  };
  auto park_mask = [&](const uint32_t (&mk)[4]) {
    if (mask_lds) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();                 // earlier reads of the previous block are done
      *reinterpret_cast<uint4*>(my_smask + lane * 4) = make_uint4(mk[0], mk[1], mk[2], mk[3]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  };

  // same software pipeline over items as in the forward kernel
  const int stride = total_waves * R;
  const int sub = lane / SW;
  Work w = fetch_work<SW>(P.g, wave0 * R + sub, n_items);
  Work wn = fetch_work<SW>(P.g, wave0 * R + stride + sub, n_items);
  int col0, eid0;
  load_cols<SW, NEED_EID>(P.g, w, w.beg, sl, col0, eid0);
  uint32_t mk0[4], mkn[4];
  if constexpr (MASKP) load_mask(w, w.beg, eid0, mk0);
  for (int base = wave0 * R; base < n_items; base += stride) {
#ifndef DGCN_NO_PREFETCH_BWD
    int coln, eidn;
    load_cols<SW, NEED_EID>(P.g, wn, wn.beg, sl, coln, eidn);
    if constexpr (MASKP) load_mask(wn, wn.beg, eidn, mkn);
    const Work wnn = fetch_work<SW>(P.g, base + 2 * stride + sub, n_items);
#endif
    for (int cb = 0; cb < C; cb += LPR * VEC) {
      const int c0 = cb + cl * VEC;
      const bool act = c0 < C;
      float xs[VEC], acc[VEC], ksh[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) { xs[j] = 0.f; acc[j] = 0.f; ksh[j] = 0.f; }
      if (act && w.row >= 0 && !ea_is_z) load_vec<VEC>(xs, P.x + static_cast<int64_t>(w.row) * P.x_stride + c0);
      if constexpr (MODE == kModeSoftmaxShifted) {
        if (act) load_vec<VEC>(ksh, P.kshift + c0);
      }
      if constexpr (EA == 2) enc_load<VEC>(enc, P.enc_w, P.enc_b, c0, act);

      int mycol = col0, myeid = eid0;
      for (int blk = w.beg; any_sub<SW>(blk < w.end); blk += SW) {
        const int nb = max(0, min(SW, w.end - blk));
        if (blk != w.beg) load_cols<SW, NEED_EID>(P.g, w, blk, sl, mycol, myeid);
        if constexpr (MASKP) {
          if (blk != w.beg) {
            uint32_t mkb[4];
            load_mask(w, blk, myeid, mkb);
            park_mask(mkb);
          } else {
            park_mask(mk0);
          }
        }
        for (int s0 = 0; any_sub<SW>(s0 < nb); s0 += G * U) {
          float gc[U][VEC], a1[U][VEC], oo[U][VEC], ea[U][VEC];
          float fe[(EA == 2) ? U : 1][kEncF];
          int ai[U][VEC];
          bool ok[U];
          int eid[U];
          int dsts[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int ei = s0 + u * G + g;
            ok[u] = ei < nb;
            const int dst = __shfl(mycol, sbase + (ei & (SW - 1)));
            dsts[u] = dst;
            eid[u] = 0;
            if constexpr (NEED_EID) eid[u] = __shfl(myeid, sbase + (ei & (SW - 1)));
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
              gc[u][j] = 0.f; a1[u][j] = 0.f; oo[u][j] = 0.f; ea[u][j] = 0.f; ai[u][j] = -1;
            }
            if (ok[u] && act) {
              const int64_t ro = static_cast<int64_t>(dst) * C + c0;
              if constexpr (MODE == kModeSoftmaxShifted) {
                load_vec<VEC>(gc[u], P.gshift + ro);
              } else if constexpr (MODE != DGCN_AGGR_MAX) {   // MAX: g is fetched below, only where an arg-max matches
                load_vec<VEC>(gc[u], P.gcoef + ro);
              }
              if constexpr (MODE == DGCN_AGGR_SOFTMAX) {
                load_vec<VEC>(a1[u], static_cast<const float*>(P.aux1) + ro);
                if (learn_t) load_vec<VEC>(oo[u], P.out + ro);
              }
              if constexpr (MODE == DGCN_AGGR_MAX) {
                if (P.maxmask) {
                  // 4 bytes of this edge's arg-max bit mask (eid is its CSR position here) instead of 4 VEC bytes of
                  // the destination's arg-max row: the VEC bits of a lane never straddle a word (c0 % VEC == 0)
                  uint32_t word;
                  if (MASKP && mask_lds) {
                    word = my_smask[(sbase + (ei & (SW - 1))) * 4 + (c0 >> 5)];
                  } else {
                    word = P.maxmask[static_cast<int64_t>(eid[u]) * P.mask_words + (c0 >> 5)];
                  }
#pragma unroll
                  for (int j = 0; j < VEC; ++j) ai[u][j] = ((word >> ((c0 & 31) + j)) & 1u) ? eid[u] : -1;
                } else {
                  load_vec_i<VEC>(ai[u], static_cast<const int32_t*>(P.aux1) + ro);
                }
              }
              if constexpr (EA == 1) {
                // max over saved pre-activations: the forward marked the channels without a positive neighbour in the
                // arg-max ids, so the relu mask is known without reading z (P.ea may be null)
                if (!(MODE == DGCN_AGGR_MAX && ea_is_z)) {
                  load_vec<VEC>(ea[u], P.ea + static_cast<int64_t>(eid[u]) * C + c0);
                }
              }
              if constexpr (EA == 2) enc_feat_row(fe[u], P.enc_feat, eid[u]);
            }
          }
          if constexpr (MODE == DGCN_AGGR_MAX) {
            // An edge receives gradient only in the channels whose arg-max it is (about 1/deg of them): read the
            // arg-max ids for every edge, but g only for the lanes with a hit -- half the gathered bytes.
#pragma unroll
            for (int u = 0; u < U; ++u) {
              bool hit = false;
#pragma unroll
              for (int j = 0; j < VEC; ++j) hit = hit || (ai[u][j] == eid[u]);
              if (ok[u] && act && hit) load_vec<VEC>(gc[u], P.gcoef + static_cast<int64_t>(dsts[u]) * C + c0);
            }
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
            float dz[VEC];
            if constexpr (EA == 2) {
              if (act) enc_apply<VEC>(ea[u], enc, fe[u]);
            }
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
              const float z = (EA != 0) ? xs[j] + ea[u][j] : xs[j];   // xs == 0 when the edge rows are z itself
              const float m = msg_apply(z, msg, eps);
              const float r = (msg == DGCN_MSG_RELU_EPS && !(MODE == DGCN_AGGR_MAX && ea_is_z)) ? (z > 0.f ? 1.f : 0.f) : 1.f;
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
                k = in ? (p_is_one ? gc[u][j] : gc[u][j] * fast_pow(uu, p - 1.f)) : 0.f;
              } else if constexpr (MODE == DGCN_AGGR_MAX) {
                k = (ai[u][j] == eid[u]) ? gc[u][j] : 0.f;
              } else {
                k = gc[u][j];
              }
              dz[j] = r * k;
              acc[j] += dz[j];
            }
            if constexpr (EA == 1) {
              if (act && P.grad_ea) {
                store_vec<VEC>(P.grad_ea + static_cast<int64_t>(eid[u]) * C + c0, dz);
              }
            }
            if constexpr (EA == 2) {
              if (act) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                  genc.b[j] += dz[j];
#pragma unroll
                  for (int f = 0; f < kEncF; ++f) genc.w[j][f] = fmaf(dz[j], fe[u][f], genc.w[j][f]);
                }
              }
            }
          }
        }
      }
#pragma unroll
      for (int off = LPR; off < SW; off <<= 1) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] += __shfl_xor(acc[j], off);
      }
      if (g == 0 && act && w.row >= 0) {
        if (w.slot >= 0) {
          store_vec<VEC>(P.ws + static_cast<int64_t>(w.slot) * C + c0, acc);
        } else {
          if (P.groot) {
            float gr[VEC];
            load_vec<VEC>(gr, P.groot + static_cast<int64_t>(w.row) * C + c0);
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[j] += gr[j];
          }
          store_vec<VEC>(P.grad_x + static_cast<int64_t>(w.row) * C + c0, acc);
        }
      }
    }
#ifndef DGCN_NO_PREFETCH_BWD
    w = wn;
    wn = wnn;
    col0 = coln;
    eid0 = eidn;
    if constexpr (MASKP) { mk0[0] = mkn[0]; mk0[1] = mkn[1]; mk0[2] = mkn[2]; mk0[3] = mkn[3]; }
#else
    w = fetch_work<SW>(P.g, base + stride + sub, n_items);
    load_cols<SW, NEED_EID>(P.g, w, w.beg, sl, col0, eid0);
    if constexpr (MASKP) load_mask(w, w.beg, eid0, mk0);
#endif
  }

  if constexpr (EA == 2) {
    // dW | db of this workgroup: lanes with the same channel group (cl) are summed with shuffles, the four waves
    // through LDS in a fixed order, and the workgroup writes one (C, kEncF + 1) partial; the host sums the partials.
    constexpr int NV = VEC * (kEncF + 1);
    __shared__ float red[kWavesPerWg][LPR * NV];
    float vals[NV];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
#pragma unroll
      for (int f = 0; f < kEncF; ++f) vals[j * (kEncF + 1) + f] = genc.w[j][f];
      vals[j * (kEncF + 1) + kEncF] = genc.b[j];
    }
#pragma unroll
    for (int off = LPR; off < kWave; off <<= 1) {
#pragma unroll
      for (int q = 0; q < NV; ++q) vals[q] += __shfl_xor(vals[q], off);
    }
    const int wv = threadIdx.x >> 6;
    if (lane < LPR) {
#pragma unroll
      for (int q = 0; q < NV; ++q) red[wv][lane * NV + q] = vals[q];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < LPR * NV; i += kWgThreads) {
      const int ch = (i / NV) * VEC + (i % NV) / (kEncF + 1);
      if (ch < C) {
        const float tsum = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
        P.enc_gpart[(static_cast<int64_t>(blockIdx.x) * C + ch) * (kEncF + 1) + (i % NV) % (kEncF + 1)] = tsum;
      }
    }
  }
}

template <int MODE, int VEC, int LPR, int SW, int EA>
__global__ __launch_bounds__(kWgThreads) void gen_aggr_bwd_kernel(const BwdParams P) {
  if constexpr (MODE == DGCN_AGGR_SOFTMAX) {
    // single-gather form when the caller prepared it and the device-side range check passed
    if (P.gshift != nullptr && !P.learn_t && ((*P.shift_ok != 0) != (P.shift_bad != 0))) {
      gen_aggr_bwd_body<kModeSoftmaxShifted, VEC, LPR, SW, EA>(P);
      return;
    }
  }
  gen_aggr_bwd_body<MODE, VEC, LPR, SW, EA>(P);
}

// ---------------------------------------------------------------------------------------
// backward with the per-edge encoder (EA == 2), wave-uniform walk (see enc_fwd_body in gen_aggr_fwd.hip)
// ---------------------------------------------------------------------------------------
// One source row (or hub piece) per wave, all 64 lanes on the channels of ONE edge at a time.  Per 64-edge block lane l
// stages edge blk + l: destination id, original edge id, its 32 bytes of raw features (parked in LDS, read back as two
// broadcast ds_read_b128 when the edge is processed).  Per edge: the destination's coefficient row(s) are gathered
// (g, plus log-sum-exp / arg-max ids / outputs by aggregator), z = x_s + W f_e + b is recomputed from registers,
// dz accumulates into grad_x[s] and into this wave's dW | db sums.  The rows of the NEXT batch are requested before the
// current batch is processed.
template <int MODE, int VEC>
__device__ __forceinline__ void enc_bwd_body(const BwdParams& P) {
  constexpr bool SHIFTED = MODE == kModeSoftmaxShifted;
  constexpr bool SOFT = MODE == DGCN_AGGR_SOFTMAX;
  constexpr bool MAXM = MODE == DGCN_AGGR_MAX;
  // edges per batch.  One gathered row per edge (add / mean / power / shifted softmax): batches of four, the next
  // batch's rows requested before the current one is processed (U = 8 needs 174 VGPRs); two or three rows per edge
  // (max: arg-max ids + g; softmax: g + log-sum-exp [+ out]): one batch at a time, four waves per SIMD hide the rest
  constexpr int U = 4;
  constexpr bool DB = !(SOFT || MAXM);
  constexpr int NV = VEC * (kEncF + 1);
  __shared__ __attribute__((aligned(16))) float sfeat[kWavesPerWg][kWave * kEncF];
  __shared__ float red[kWavesPerWg][kWave * NV];
  const int lane = lane_id();
  const int wv = threadIdx.x >> 6;
  const int C = P.C;
  const int c0 = lane * VEC;
  const bool act = c0 < C;
  const int n_items = P.g.n_work ? P.g.n_work : P.g.n_rows;
  const float t = P.t_dev ? *P.t_dev : P.t;
  const float p = P.p_dev ? *P.p_dev : P.p;
  const float eps = P.eps;
  const int msg = P.msg;
  const bool learn_t = SOFT && P.learn_t != 0;
  const bool p_is_one = p == 1.f;
  EncW<VEC> enc, genc;
  enc_load<VEC>(enc, P.enc_w, P.enc_b, c0, act);
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    genc.b[j] = 0.f;
#pragma unroll
    for (int f = 0; f < kEncF; ++f) genc.w[j][f] = 0.f;
  }
  float ksh[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) ksh[j] = 0.f;
  if constexpr (SHIFTED) {
    if (act) load_vec<VEC>(ksh, P.kshift + c0);
  }

  struct Stage { int col, eid; float4 f0, f1; };
  auto stage_load = [&](const Work& w, int blk) -> Stage {
    Stage sg;
    sg.col = 0; sg.eid = 0;
    sg.f0 = make_float4(0.f, 0.f, 0.f, 0.f); sg.f1 = sg.f0;
    if (lane < w.end - blk) {
      sg.col = P.g.col[blk + lane];
      sg.eid = P.g.eperm ? P.g.eperm[blk + lane] : blk + lane;
      const float4* fp = reinterpret_cast<const float4*>(P.enc_feat + static_cast<int64_t>(sg.eid) * kEncF);
      sg.f0 = fp[0]; sg.f1 = fp[1];
    }
    return sg;
  };
  struct Rows { float gc[U][VEC], a1[U][VEC], oo[U][VEC]; int ai[U][VEC]; };

  float* sf = sfeat[wv];
  ItemQueue q;
  int item = q.first(P.ticket);
  Work w = fetch_work<kWave>(P.g, item, n_items);
  Stage sg = stage_load(w, w.beg);
  while (item < n_items) {
    const int next = q.next(P.ticket);
    const Work wn = fetch_work<kWave>(P.g, next, n_items);
    float xs[VEC], acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { xs[j] = 0.f; acc[j] = 0.f; }
    if (act && w.row >= 0) load_vec<VEC>(xs, P.x + static_cast<int64_t>(w.row) * P.x_stride + c0);
#pragma unroll
    for (int j = 0; j < VEC; ++j) xs[j] += enc.b[j];                    // z = (x_s + b) + W f_e
    for (int blk = w.beg; blk < w.end || blk == w.beg; blk += kWave) {
      const int nb = max(0, min(kWave, w.end - blk));
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();                 // the previous block's feature reads are done
      *reinterpret_cast<float4*>(sf + lane * kEncF) = sg.f0;
      *reinterpret_cast<float4*>(sf + lane * kEncF + 4) = sg.f1;
      const int mycol = sg.col, myeid = sg.eid;
      const bool last_blk = blk + kWave >= w.end;
      sg = last_blk ? stage_load(wn, wn.beg) : stage_load(w, blk + kWave);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();

      auto load_batch = [&](Rows& R, int s0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) { R.gc[u][j] = 0.f; R.a1[u][j] = 0.f; R.oo[u][j] = 0.f; R.ai[u][j] = -1; }
          if (s0 + u < nb) {                                             // wave-uniform
            const int dst = __builtin_amdgcn_readlane(mycol, s0 + u);
            if (act) {
              const int64_t ro = static_cast<int64_t>(dst) * C + c0;
              if constexpr (SHIFTED) {
                load_vec<VEC>(R.gc[u], P.gshift + ro);
              } else {
                load_vec<VEC>(R.gc[u], P.gcoef + ro);
              }
              if constexpr (SOFT) {
                load_vec<VEC>(R.a1[u], static_cast<const float*>(P.aux1) + ro);
                if (learn_t) load_vec<VEC>(R.oo[u], P.out + ro);
              }
              if constexpr (MAXM) load_vec_i<VEC>(R.ai[u], static_cast<const int32_t*>(P.aux1) + ro);
            }
          }
        }
      };
      auto fold_batch = [&](const Rows& R, int s0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (s0 + u < nb) {                                             // wave-uniform
            int eid = 0;
            if constexpr (MAXM) eid = __builtin_amdgcn_readlane(myeid, s0 + u);
            const float4 fa = *reinterpret_cast<const float4*>(sf + (s0 + u) * kEncF);   // broadcast reads
            const float4 fb = *reinterpret_cast<const float4*>(sf + (s0 + u) * kEncF + 4);
            const float fe[kEncF] = {fa.x, fa.y, fa.z, fa.w, fb.x, fb.y, fb.z, fb.w};
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
              float z = xs[j];
#pragma unroll
              for (int f = 0; f < kEncF; ++f) z = fmaf(enc.w[j][f], fe[f], z);
              const float m = msg_apply(z, msg, eps);
              const float r = (msg == DGCN_MSG_RELU_EPS) ? (z > 0.f ? 1.f : 0.f) : 1.f;
              float k;
              if constexpr (SOFT) {
                float wgt = fast_exp(t * m - R.a1[u][j]);
                if (learn_t) wgt *= 1.f + t * (m - R.oo[u][j]);
                k = R.gc[u][j] * wgt;
              } else if constexpr (SHIFTED) {
                k = R.gc[u][j] * fast_exp(t * m - ksh[j]);
              } else if constexpr (MODE == DGCN_AGGR_POWER) {
                const bool in = (m >= kPowLo) && (m <= kPowHi);
                const float uu = fminf(fmaxf(m, kPowLo), kPowHi);
                k = in ? (p_is_one ? R.gc[u][j] : R.gc[u][j] * fast_pow(uu, p - 1.f)) : 0.f;
              } else if constexpr (MAXM) {
                k = (R.ai[u][j] == eid) ? R.gc[u][j] : 0.f;
              } else {
                k = R.gc[u][j];
              }
              const float dz = r * k;
              acc[j] += dz;
              genc.b[j] += dz;
#pragma unroll
              for (int f = 0; f < kEncF; ++f) genc.w[j][f] = fmaf(dz, fe[f], genc.w[j][f]);
            }
          }
        }
      };
      if constexpr (DB) {
        if (nb > 0) {
          Rows ra, rb;
          load_batch(ra, 0);
          for (int s0 = 0; s0 < nb; s0 += 2 * U) {
            if (s0 + U < nb) load_batch(rb, s0 + U);
            fold_batch(ra, s0);
            if (s0 + U < nb) {
              if (s0 + 2 * U < nb) load_batch(ra, s0 + 2 * U);
              fold_batch(rb, s0 + U);
            }
          }
        }
      } else {
        for (int s0 = 0; s0 < nb; s0 += U) {
          Rows ra;
          load_batch(ra, s0);
          fold_batch(ra, s0);
        }
      }
      if (last_blk) break;
    }
    if (act && w.row >= 0) {
      if (w.slot >= 0) {
        store_vec<VEC>(P.ws + static_cast<int64_t>(w.slot) * C + c0, acc);
      } else {
        if (P.groot) {
          float gr[VEC];
          load_vec<VEC>(gr, P.groot + static_cast<int64_t>(w.row) * C + c0);
#pragma unroll
          for (int j = 0; j < VEC; ++j) acc[j] += gr[j];
        }
        store_vec<VEC>(P.grad_x + static_cast<int64_t>(w.row) * C + c0, acc);
      }
    }
    w = wn;
    item = next;
  }

  // dW | db of this workgroup: the four waves through LDS in a fixed order, one (C, kEncF + 1) partial per workgroup
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
#pragma unroll
    for (int f = 0; f < kEncF; ++f) red[wv][lane * NV + j * (kEncF + 1) + f] = genc.w[j][f];
    red[wv][lane * NV + j * (kEncF + 1) + kEncF] = genc.b[j];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kWave * NV; i += kWgThreads) {
    const int ch = (i / NV) * VEC + (i % NV) / (kEncF + 1);
    if (ch < C) {
      const float tsum = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
      P.enc_gpart[(static_cast<int64_t>(blockIdx.x) * C + ch) * (kEncF + 1) + (i % NV) % (kEncF + 1)] = tsum;
    }
  }
}

template <int MODE, int VEC>
__global__ __launch_bounds__(kWgThreads) void gen_aggr_enc_bwd_kernel(const BwdParams P) {
  if constexpr (MODE == DGCN_AGGR_SOFTMAX) {
    if (P.gshift != nullptr && !P.learn_t && ((*P.shift_ok != 0) != (P.shift_bad != 0))) {
      enc_bwd_body<kModeSoftmaxShifted, VEC>(P);
      return;
    }
  }
  enc_bwd_body<MODE, VEC>(P);
}

// Per-destination coefficient of the power-mean / mean backward, one streaming pass instead
// of five elementwise torch kernels:  out[i,c] = g[i,c] * r^(1/p - 1) * [lo <= q <= hi] / max(deg_i, 1),
// r = clamp(q, lo, hi), q = the forward's pre-clamp mean (aux1); q == nullptr gives the MEAN form g / max(deg, 1).
__global__ __launch_bounds__(kWgThreads) void power_bwd_prep_kernel(const float* __restrict__ g,
                                                                    const float* __restrict__ q,
                                                                    const int32_t* __restrict__ rowptr,
                                                                    const float* __restrict__ p_dev, float p,
                                                                    float* __restrict__ out, int64_t n_elems, int C) {
  const float pp = p_dev ? *p_dev : p;
  const float ex = 1.f / pp - 1.f;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n_elems; i += stride) {
    const int64_t row = i / C;
    const float deg = fmaxf(static_cast<float>(rowptr[row + 1] - rowptr[row]), 1.f);
    float v = g[i] / deg;
    if (q) {
      const float qq = q[i];
      const bool in = (qq >= kPowLo) && (qq <= kPowHi);
      const float r = fminf(fmaxf(qq, kPowLo), kPowHi);
      v = in ? v * fast_pow(r, ex) : 0.f;
    }
    out[i] = v;
  }
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

constexpr int kMergeCh = 16;                       // channels per merge wave (as in gen_aggr_fwd.hip)
constexpr int kMergeGroups = kWave / kMergeCh;     // piece groups per merge wave
__global__ __launch_bounds__(kWgThreads) void gen_aggr_bwd_merge_kernel(const BwdParams P) {
  // one wave per (split row, block of 16 channels): four piece groups x 16 channels, every group sums a contiguous
  // quarter of the row's pieces, 16 partial rows in flight, and two shuffles add the groups in a fixed order.  The
  // pieces of a row own consecutive items and slots (graph_build.hip work_fill_kernel): no dependent index loads.
  const int lane = lane_id();
  const int C = P.C;
  const int cblocks = (C + kMergeCh - 1) / kMergeCh;
  const int wave = blockIdx.x * kWavesPerWg + (threadIdx.x >> 6);
  if (wave >= P.g.n_split * cblocks) return;
  const int i0 = uni(P.g.split_item[wave / cblocks]);
  const int row = uni(P.g.work_row[i0]);
  const int slot0 = uni(P.g.work_slot[i0]);
  const int rbeg = uni(P.g.rowptr[row]), rend = uni(P.g.rowptr[row + 1]);
  const int chunk = uni(P.g.work_end[i0]) - uni(P.g.work_beg[i0]);
  const int npieces = (rend - rbeg + chunk - 1) / chunk;
  const int c_raw = (wave % cblocks) * kMergeCh + (lane & (kMergeCh - 1));
  const int c = min(c_raw, C - 1);                 // every lane stays in the shuffles; lanes past C do not write
  const int grp = lane / kMergeCh;
  const int per = (npieces + kMergeGroups - 1) / kMergeGroups;
  const int pbeg = min(grp * per, npieces), pend = min(pbeg + per, npieces);
  float acc = 0.f;
  for (int i = pbeg; i < pend; i += 16) {
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = (i + k < pend) ? P.ws[static_cast<int64_t>(slot0 + i + k) * C + c] : 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc += v[k];
  }
#pragma unroll
  for (int off = kMergeCh; off < kWave; off <<= 1) acc += __shfl_xor(acc, off);
  if (grp != 0 || c_raw >= C) return;
  P.grad_x[static_cast<int64_t>(row) * C + c] = P.groot ? acc + P.groot[static_cast<int64_t>(row) * C + c] : acc;
}
template <int MODE, int VEC, int LPR, int SW>
void launch_bwd_ea(const BwdParams& P, int grid, hipStream_t s) {
  if constexpr (VEC == 4 && LPR <= 16 && SW == kWave) {   // per-edge encoder at C < 64 (wider rows: gen_aggr_enc_bwd_kernel)
    if (P.enc_feat) {
      hipLaunchKernelGGL((gen_aggr_bwd_kernel<MODE, VEC, LPR, SW, 2>), dim3(grid), dim3(kWgThreads), 0, s, P);
      return;
    }
  }
  if (P.ea || P.ea_is_z) {
    hipLaunchKernelGGL((gen_aggr_bwd_kernel<MODE, VEC, LPR, SW, 1>), dim3(grid), dim3(kWgThreads), 0, s, P);
  } else {
    hipLaunchKernelGGL((gen_aggr_bwd_kernel<MODE, VEC, LPR, SW, 0>), dim3(grid), dim3(kWgThreads), 0, s, P);
  }
}

template <int MODE>
void launch_bwd_mode(const BwdParams& P, int vec, int lpr, int grid, hipStream_t s) {
  if (vec == 4 && P.enc_feat && P.C >= 64 && P.C <= 256) {      // wave-uniform encoder walk (as in the forward)
    if (P.C <= 128) hipLaunchKernelGGL((gen_aggr_enc_bwd_kernel<MODE, 2>), dim3(grid), dim3(kWgThreads), 0, s, P);
    else hipLaunchKernelGGL((gen_aggr_enc_bwd_kernel<MODE, 4>), dim3(grid), dim3(kWgThreads), 0, s, P);
  } else if (vec == 4) {
    // (LPR, SW) pairs: SW = LPR * edge groups per row (kEdgeGroups) when the graph has enough rows to fill the
    // chip that way, else one row per wave (more, shorter waves)
    const int sw_sel = subgroup_width(lpr, P.g.n_work ? P.g.n_work : P.g.n_rows, P.n_edges_hint);
    // (per-edge encoder at C < 64: the side-by-side layouts need > 256 registers there, one row per wave only)
    const bool one = sw_sel == kWave || P.enc_feat != nullptr;
    switch (lpr) {
      case 4: one ? launch_bwd_ea<MODE, 4, 4, 64>(P, grid, s) : launch_bwd_ea<MODE, 4, 4, kSubWidth(4)>(P, grid, s); break;
      case 8: one ? launch_bwd_ea<MODE, 4, 8, 64>(P, grid, s) : launch_bwd_ea<MODE, 4, 8, kSubWidth(8)>(P, grid, s); break;
      case 16: one ? launch_bwd_ea<MODE, 4, 16, 64>(P, grid, s) : launch_bwd_ea<MODE, 4, 16, kSubWidth(16)>(P, grid, s); break;
      case 32: (sw_sel == 32) ? launch_bwd_ea<MODE, 4, 32, 32>(P, grid, s) : launch_bwd_ea<MODE, 4, 32, 64>(P, grid, s); break;
      default: launch_bwd_ea<MODE, 4, 64, 64>(P, grid, s); break;
    }
  } else {
    launch_bwd_ea<MODE, 1, 64, 64>(P, grid, s);
  }
  if (P.g.n_work && P.g.n_split > 0) {
    const int mwaves = P.g.n_split * ((P.C + kMergeCh - 1) / kMergeCh);
    const int mg = (mwaves + kWavesPerWg - 1) / kWavesPerWg;
    hipLaunchKernelGGL(gen_aggr_bwd_merge_kernel, dim3(mg), dim3(kWgThreads), 0, s, P);
  }
}


int bwd_grid(const dgcn_graph* g, int channels, bool vec4, bool enc) {
  const int lpr = vec4 ? lanes_per_row(channels, 4) : 64;
  int per_wave = vec4 ? kWave / subgroup_width(lpr, (g->t_n_work ? g->t_n_work : g->n_src), g->n_edges) : 1;
  if (enc && channels < 64) per_wave = 1;          // launch_bwd_mode: one row per wave for the narrow encoder shapes
  const int n_items = ((g->t_n_work ? g->t_n_work : g->n_src) + per_wave - 1) / per_wave;
  int grid = round_up8(grid_for_waves(n_items));
  if (enc && grid > kEncMaxParts) grid = kEncMaxParts;
  return grid;
}

// Arg-max bit masks for the max backward: bit c of mask[p] says whether CSR position p is the arg-max of its destination
// row in channel c.  One wave per destination row, lane = channel: the arg-max edge id is located among the row's
// (ascending) original edge ids by bisection, OR-ed into an LDS tile of the row's 64-edge chunk, and the chunk's masks
// leave with one coalesced store per lane.  Work ~ rows x channels (the winners), not edges x channels.
template <int W>
__global__ __launch_bounds__(kWgThreads) void max_mask_build_kernel(const int32_t* __restrict__ argmax,
                                                                    const int32_t* __restrict__ rowptr,
                                                                    const int32_t* __restrict__ eperm, int n_rows, int C,
                                                                    uint32_t* __restrict__ mask) {
  constexpr int KC = (W + 1) / 2;
  __shared__ uint32_t tile_all[kWavesPerWg][kWave * W];
  uint32_t* tile = tile_all[threadIdx.x >> 6];
  const int lane = lane_id();
  const int total_waves = gridDim.x * kWavesPerWg;
  for (int row = blockIdx.x * kWavesPerWg + (threadIdx.x >> 6); row < n_rows; row += total_waves) {
    const int beg = uni(rowptr[row]), end = uni(rowptr[row + 1]);
    if (beg == end) continue;
    int pos[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      const int c = lane + 64 * k;
      const int a = (c < C) ? argmax[static_cast<int64_t>(row) * C + c] : -1;
      int p = -1;
      if (a >= 0) {
        if (!eperm) {
          p = a;                                   // destination-sorted input: CSR position == original edge id
        } else {
          int lo = beg, hi = end - 1;              // eperm[beg..end) ascends (stable sort); a is one of them
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (eperm[mid] < a) lo = mid + 1; else hi = mid;
          }
          p = lo;
        }
      }
      pos[k] = p;
    }
    for (int chunk = beg; chunk < end; chunk += kWave) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < W; ++q) tile[lane * W + q] = 0u;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < KC; ++k) {
        const int c = lane + 64 * k;
        const int rel = pos[k] - chunk;
        if (pos[k] >= 0 && rel >= 0 && rel < kWave) atomicOr(&tile[rel * W + (c >> 5)], 1u << (c & 31));
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (chunk + lane < end) {
        uint32_t* o = mask + static_cast<int64_t>(chunk + lane) * W;
        if constexpr (W >= 4) {
#pragma unroll
          for (int q = 0; q < W; q += 4) {
            *reinterpret_cast<uint4*>(o + q) =
                make_uint4(tile[lane * W + q], tile[lane * W + q + 1], tile[lane * W + q + 2], tile[lane * W + q + 3]);
          }
        } else {
#pragma unroll
          for (int q = 0; q < W; ++q) o[q] = tile[lane * W + q];
        }
      }
    }
  }
}

inline int max_mask_words(int channels) {
  const int need = (channels + 31) / 32;
  int w = 1;
  while (w < need) w <<= 1;
  return w;   // 1, 2, 4 or 8 (channels <= 256)
}

int gen_aggr_bwd_impl(const dgcn_graph* g, const float* x, int64_t x_stride,
                      const float* edge_attr, const EncArgs* enc, float* enc_gpart, int32_t channels, int32_t mode,
                      int32_t msg, int32_t flags, float t, float p, float eps,
                      const float* t_dev, const float* p_dev, const float* gcoef,
                      const void* aux1, const float* out, const float* gshift,
                      const float* kshift, const int32_t* shift_ok, const float* groot,
                      float* grad_x, float* grad_edge_attr, void* workspace,
                      size_t workspace_bytes, void* stream, const uint32_t* maxmask = nullptr,
                      const int32_t* t_cpos = nullptr) {
  const bool ea_is_z = (flags & DGCN_FLAG_EA_IS_Z) != 0;
  // max needs no pre-activations at all (the forward's arg-max ids carry the relu mask): edge_attr may be NULL there
  if (ea_is_z && (enc || (!edge_attr && mode != DGCN_AGGR_MAX))) return DGCN_E_MODE;
  if (ea_is_z && !x) { x = gcoef; x_stride = channels; }        // never read: the rows of edge_attr are z_e itself
  if (!g || !x || !gcoef || !grad_x) return DGCN_E_NULL;
  if (const int rc = enc_check(enc, channels)) return rc;
  if (enc && !enc_gpart) return DGCN_E_NULL;
  if (g->n_src < 0 || g->n_edges < 0 || channels <= 0 || x_stride < channels) return DGCN_E_SHAPE;
  if (mode < DGCN_AGGR_ADD || mode > DGCN_AGGR_POWER) return DGCN_E_MODE;
  if (msg != DGCN_MSG_IDENTITY && msg != DGCN_MSG_RELU_EPS) return DGCN_E_MODE;
  if ((mode == DGCN_AGGR_SOFTMAX || mode == DGCN_AGGR_MAX) && !aux1) return DGCN_E_NULL;
  if (mode == DGCN_AGGR_SOFTMAX && (flags & DGCN_FLAG_LEARN_T) && !out) return DGCN_E_NULL;
  if (g->n_src == 0) return DGCN_OK;
  if (!g->t_rowptr || (g->n_edges > 0 && (!g->t_col || !g->t_eperm))) return DGCN_E_NULL;
  if (g->t_n_work && (!g->t_work_row || !g->t_work_beg || !g->t_work_end || !g->t_work_slot)) return DGCN_E_NULL;
  if (g->t_n_work && g->t_n_split > 0 && !g->t_split_item) return DGCN_E_NULL;
  {
    const size_t full = dgcn_gen_aggr_bwd_workspace_bytes(g, channels);
    const bool walk = enc && channels % 4 == 0 && channels >= 64 && channels <= 256;
    const size_t slots = g->t_n_work ? static_cast<size_t>(g->t_n_slots) * static_cast<size_t>(channels) * sizeof(float) : 0;
    if (workspace_bytes < (walk ? full : slots)) return DGCN_E_WORKSPACE;
  }
  if (g->t_n_work && g->t_n_slots > 0 && !workspace) return DGCN_E_NULL;

  const bool vec4 = (channels % 4 == 0) && (x_stride % 4 == 0) && aligned16(x) && aligned16(gcoef) &&
                    aligned16(grad_x) && (!edge_attr || aligned16(edge_attr)) && (!groot || aligned16(groot)) &&
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
  P.ea_is_z = ea_is_z ? 1 : 0;
  P.t = t; P.p = p; P.eps = eps; P.t_dev = t_dev; P.p_dev = p_dev;
  P.gcoef = gcoef; P.aux1 = aux1; P.out = out; P.grad_x = grad_x; P.grad_ea = grad_edge_attr;
  P.gshift = nullptr; P.kshift = nullptr; P.shift_ok = nullptr; P.shift_bad = 0;
  P.groot = groot;
  P.enc_feat = enc ? enc->feat : nullptr;
  P.enc_w = enc ? enc->w : nullptr;
  P.enc_b = enc ? enc->b : nullptr;
  P.enc_gpart = enc_gpart;
  P.n_edges_hint = g->n_edges;
  P.maxmask = nullptr; P.mask_words = 0;
  if (maxmask) {
    if (mode != DGCN_AGGR_MAX || edge_attr || enc || ea_is_z || !t_cpos || channels > 256) return DGCN_E_MODE;
    P.maxmask = maxmask; P.mask_words = max_mask_words(channels);
    P.g.eperm = t_cpos;            // the walk's "edge id" is the CSR position: the index of the mask rows
  }
  if (enc && !vec4) return DGCN_E_ALIGN;
  if (mode == DGCN_AGGR_SOFTMAX && gshift && kshift && shift_ok && vec4 && aligned16(gshift) && aligned16(kshift)) {
    P.gshift = gshift; P.kshift = kshift; P.shift_ok = shift_ok;
    P.shift_bad = (flags & DGCN_FLAG_SHIFT_FLAG_IS_RANGE) ? 1 : 0;
  }
  P.ws = static_cast<float*>(workspace);
  P.ticket = nullptr;
  if (enc && vec4 && channels >= 64 && channels <= 256 && !(flags & DGCN_FLAG_STATIC_ITEMS)) {       // wave-uniform encoder walk: work-item counter
    if (!workspace) return DGCN_E_NULL;
    P.ticket = reinterpret_cast<int32_t*>(static_cast<char*>(workspace) +
                                          dgcn_gen_aggr_bwd_workspace_bytes(g, channels) - kTicketBytes);
    if (const int zrc = zero_async(P.ticket, kTicketBytes, static_cast<hipStream_t>(stream))) return zrc;   // a kernel, not a memset node: dgcn_common.h
  }
  const int grid = bwd_grid(g, channels, vec4, enc != nullptr);
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


// dW' | db' of the per-edge encoder under MAX from the arg-max winners: dz has one non-zero per (destination row,
// channel) -- g[r][c] at the edge the forward stored (original id; -1 = relu floor or empty row, see max_id_for_bwd in
// gen_aggr_fwd.hip) -- so  dW'[c][f] = sum_r g[r][c] feat[arg[r][c]][f],  db'[c] = sum_{r: arg >= 0} g[r][c]:
// n_dst * C gathers of 32 bytes instead of a per-edge pass.  Thread = channel, kEwRows destination rows per workgroup,
// four gathers in flight per thread; partials [grid][C][kEncF + 1], every block fully written, fixed order.
constexpr int kEwRows = 16;

#ifdef DGCN_DEBUG_IDS
// investigation build only (python -m deep_gcns_torch_amd.build --debug-ids -> libdgcn_dbg.so): ids that are neither -1
// nor an edge of the graph are counted and the first 64 recorded as (row, channel, id)
__device__ int dgcn_dbg_bad[1 + 3 * 64 + 8 + 16];   // [201..216] = bad ids per row 0..15;   // [193..] = bad ids in row 0 / rows 1-15 / rows 16-1023 / rows >= 1024
#endif

__global__ __launch_bounds__(kWgThreads) void enc_max_bwd_weight_kernel(const float* __restrict__ g,
                                                                        const int32_t* __restrict__ arg, int n_rows, int C,
                                                                        int n_edges, const float* __restrict__ feat,
                                                                        float* __restrict__ part) {
  const int c = threadIdx.x;
  if (c >= C) return;
  const int r0 = blockIdx.x * kEwRows, r1 = min(r0 + kEwRows, n_rows);
  float acc[kEncF + 1];
#pragma unroll
  for (int f = 0; f <= kEncF; ++f) acc[f] = 0.f;
  for (int r = r0; r < r1; r += 4) {
    int id[4];
    float gv[4];
    float4 fa[4], fb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool ok = r + u < r1;
      id[u] = ok ? arg[static_cast<int64_t>(r + u) * C + c] : -1;
#ifdef DGCN_DEBUG_IDS
      if (id[u] != -1 && static_cast<uint32_t>(id[u]) >= static_cast<uint32_t>(n_edges)) {
        const int k = atomicAdd(&dgcn_dbg_bad[0], 1);
        if (k < 64) { dgcn_dbg_bad[1 + 3 * k] = r + u; dgcn_dbg_bad[2 + 3 * k] = c; dgcn_dbg_bad[3 + 3 * k] = id[u]; }
        const int rr = r + u;
        atomicAdd(&dgcn_dbg_bad[193 + (rr == 0 ? 0 : (rr < 16 ? 1 : (rr < 1024 ? 2 : 3)))], 1);
        if (rr < 16) atomicAdd(&dgcn_dbg_bad[201 + rr], 1);
      }
#endif
      if (static_cast<uint32_t>(id[u]) >= static_cast<uint32_t>(n_edges)) id[u] = -1;   // never an address outside feat
      gv[u] = ok ? g[static_cast<int64_t>(r + u) * C + c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      fa[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      fb[u] = fa[u];
      if (id[u] >= 0) {
        const float4* fp = reinterpret_cast<const float4*>(feat + static_cast<int64_t>(id[u]) * kEncF);
        fa[u] = fp[0];
        fb[u] = fp[1];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (id[u] >= 0) {
        const float fe[kEncF] = {fa[u].x, fa[u].y, fa[u].z, fa[u].w, fb[u].x, fb[u].y, fb[u].z, fb[u].w};
#pragma unroll
        for (int f = 0; f < kEncF; ++f) acc[f] = fmaf(gv[u], fe[f], acc[f]);
        acc[kEncF] += gv[u];
      }
    }
  }
  float* o = part + (static_cast<int64_t>(blockIdx.x) * C + c) * (kEncF + 1);
#pragma unroll
  for (int f = 0; f <= kEncF; ++f) o[f] = acc[f];
}

}  // namespace
}  // namespace dgcn

using namespace dgcn;

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


extern "C" int dgcn_power_bwd_prep_f32(const dgcn_graph* g, const float* grad_out, const float* q,
                                       const float* p_dev, float p, float* out, int32_t channels, void* stream) {
  if (!g || !grad_out || !out || !g->rowptr) return DGCN_E_NULL;
  if (g->n_dst < 0 || channels <= 0) return DGCN_E_SHAPE;
  if (g->n_dst == 0) return DGCN_OK;
  const int64_t n = static_cast<int64_t>(g->n_dst) * channels;
  int64_t blocks = (n + kWgThreads - 1) / kWgThreads;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(power_bwd_prep_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kWgThreads), 0,
                     static_cast<hipStream_t>(stream), grad_out, q, g->rowptr, p_dev, p, out, n, channels);
  return launch_status();
}

extern "C" size_t dgcn_gen_aggr_bwd_workspace_bytes(const dgcn_graph* g, int32_t channels) {
  if (!g) return 0;
  const size_t slots = g->t_n_work ? static_cast<size_t>(g->t_n_slots) * static_cast<size_t>(channels) * sizeof(float) : 0;
  return (slots + 255u) / 256u * 256u + kTicketBytes;     // + the work-item counter of the encoder walk
}


extern "C" int dgcn_gen_aggr_bwd_f32(const dgcn_graph* g, const float* x, int64_t x_stride,
                                     const float* edge_attr, int32_t channels, int32_t mode,
                                     int32_t msg, int32_t flags, float t, float p, float eps,
                                     const float* t_dev, const float* p_dev, const float* gcoef,
                                     const void* aux1, const float* out, const float* gshift,
                                     const float* kshift, const int32_t* shift_ok, const float* groot,
                                     float* grad_x, float* grad_edge_attr, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  return gen_aggr_bwd_impl(g, x, x_stride, edge_attr, nullptr, nullptr, channels, mode, msg, flags, t, p, eps, t_dev,
                           p_dev, gcoef, aux1, out, gshift, kshift, shift_ok, groot, grad_x, grad_edge_attr,
                           workspace, workspace_bytes, stream);
}

extern "C" size_t dgcn_gen_aggr_max_mask_bytes(int32_t n_edges, int32_t channels) {
  if (n_edges <= 0 || channels <= 0 || channels > 256) return 0;
  return static_cast<size_t>(n_edges) * max_mask_words(channels) * sizeof(uint32_t);
}

extern "C" int dgcn_gen_aggr_max_bwd_f32(const dgcn_graph* g, const int32_t* t_cpos, const float* x, int64_t x_stride,
                                         int32_t channels, int32_t msg, int32_t flags, float eps, const float* gcoef,
                                         const int32_t* argmax, const float* groot, float* grad_x, void* mask,
                                         size_t mask_bytes, void* workspace, size_t workspace_bytes, void* stream) {
  if (!g || !t_cpos || !argmax || !mask || !g->rowptr) return DGCN_E_NULL;
  if (channels <= 0 || channels > 256 || g->n_edges <= 0 || g->n_dst <= 0) return DGCN_E_SHAPE;
  if (mask_bytes < dgcn_gen_aggr_max_mask_bytes(g->n_edges, channels)) return DGCN_E_WORKSPACE;
  if (!aligned16(mask)) return DGCN_E_ALIGN;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int W = max_mask_words(channels);
  uint32_t* m = static_cast<uint32_t*>(mask);
  const dim3 grid(static_cast<unsigned>(grid_for_waves(g->n_dst))), wg(kWgThreads);
  switch (W) {
    case 1: hipLaunchKernelGGL(max_mask_build_kernel<1>, grid, wg, 0, s, argmax, g->rowptr, g->eperm, g->n_dst, channels, m); break;
    case 2: hipLaunchKernelGGL(max_mask_build_kernel<2>, grid, wg, 0, s, argmax, g->rowptr, g->eperm, g->n_dst, channels, m); break;
    case 4: hipLaunchKernelGGL(max_mask_build_kernel<4>, grid, wg, 0, s, argmax, g->rowptr, g->eperm, g->n_dst, channels, m); break;
    default: hipLaunchKernelGGL(max_mask_build_kernel<8>, grid, wg, 0, s, argmax, g->rowptr, g->eperm, g->n_dst, channels, m); break;
  }
  if (const int rc = launch_status()) return rc;
  return gen_aggr_bwd_impl(g, x, x_stride, nullptr, nullptr, nullptr, channels, DGCN_AGGR_MAX, msg, flags, 1.f, 1.f, eps,
                           nullptr, nullptr, gcoef, argmax, nullptr, nullptr, nullptr, nullptr, groot, grad_x, nullptr,
                           workspace, workspace_bytes, stream, m, t_cpos);
}

extern "C" int32_t dgcn_gen_aggr_enc_bwd_num_partials(const dgcn_graph* g, int32_t channels) {
  if (!g || channels <= 0 || channels % 4 != 0 || g->n_src <= 0) return 0;
  return bwd_grid(g, channels, true, true);
}

extern "C" int dgcn_gen_aggr_enc_bwd_f32(const dgcn_graph* g, const float* x, int64_t x_stride,
                                         const float* enc_feat, const float* enc_weight, const float* enc_bias,
                                         int32_t n_feat, int32_t channels, int32_t mode, int32_t msg,
                                         int32_t flags, float t, float p, float eps, const float* t_dev,
                                         const float* p_dev, const float* gcoef, const void* aux1,
                                         const float* out, const float* gshift, const float* kshift,
                                         const int32_t* shift_ok, const float* groot, float* grad_x,
                                         float* enc_grad_partials, void* workspace, size_t workspace_bytes,
                                         void* stream) {
  const EncArgs enc{enc_feat, enc_weight, enc_bias, n_feat};
  return gen_aggr_bwd_impl(g, x, x_stride, nullptr, &enc, enc_grad_partials, channels, mode, msg, flags, t, p, eps,
                           t_dev, p_dev, gcoef, aux1, out, gshift, kshift, shift_ok, groot, grad_x, nullptr,
                           workspace, workspace_bytes, stream);
}

#ifdef DGCN_DEBUG_IDS
extern "C" int dgcn_debug_bad_ids(int32_t* host, int32_t n_ints) {
  return static_cast<int>(hipMemcpyFromSymbol(host, HIP_SYMBOL(dgcn_dbg_bad), sizeof(int32_t) * n_ints, 0,
                                              hipMemcpyDeviceToHost));
}
#endif

extern "C" int32_t dgcn_enc_max_bwd_num_partials(int32_t n_dst) {
  return n_dst > 0 ? (n_dst + kEwRows - 1) / kEwRows : 0;
}

extern "C" int dgcn_enc_max_bwd_weight_f32(const float* gcoef, const int32_t* argmax, int32_t n_dst, int32_t n_edges,
                                           const float* enc_feat, int32_t n_feat, int32_t channels,
                                           float* enc_grad_partials, void* stream) {
  if (!gcoef || !argmax || !enc_feat || !enc_grad_partials) return DGCN_E_NULL;
  if (n_feat != kEncF || n_dst < 0 || n_edges < 0 || channels <= 0 || channels > kWgThreads) return DGCN_E_SHAPE;
  if (!aligned16(enc_feat)) return DGCN_E_ALIGN;
  if (n_dst == 0) return DGCN_OK;
  hipLaunchKernelGGL(enc_max_bwd_weight_kernel, dim3(dgcn_enc_max_bwd_num_partials(n_dst)), dim3(kWgThreads), 0,
                     static_cast<hipStream_t>(stream), gcoef, argmax, n_dst, channels, n_edges, enc_feat, enc_grad_partials);
  return launch_status();
}
