// Per-row reduction state of the generalized aggregators and its fold / merge rules, shared by the row-walk kernel
// (gen_aggr_fwd.hip) and the fused edge-GEMM kernel (gen_aggr_egemm.hip).  Anonymous namespace: one copy per
// translation unit.
#pragma once

#include "gen_aggr_common.h"

namespace dgcn {
namespace {

// Per-channel reduction state.  Meaning by mode:
//   SOFTMAX: a = running max M' of s' = t*log2(e)*m, b = sum 2^(s'-M'), c = sum 2^(s'-M')*m, d = sum 2^(s'-M')*m^2
//            (inside the edge loop c, d hold the sums over r = m - eps; see softmax_fold)
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
      const float s1 = (s.a[j] == DGCN_NEG_INF) ? 0.f : fast_exp2(s.a[j] - nm);   // a is in the log2 domain
      const float s2 = (o.a[j] == DGCN_NEG_INF) ? 0.f : fast_exp2(o.a[j] - nm);
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

// ---- softmax fold, written for VALU economy ------------------------------------------------------------
// The forward kernel is co-limited by HBM and VALU issue (16 G channel-visits per launch at the products
// shape), so the per-element work is kept minimal:
//   * everything in the log2 domain: s' = t*log2(e)*m, weights exp2(s' - max') -> one v_exp_f32 per element, no
//     extra multiply;
//   * m = relu(z) + eps is never formed: s' = fma(t2, relu(z), t2*eps), and sum(e*m) = sum(e*relu(z)) + eps*sum(e)
//     is fixed up once per row;
//   * channel PAIRS as 2-vectors so mul/add/fma become v_pk_*_f32 (two channels per instruction);
//   * full batches (all U*G edge slots valid) take a variant without any masking.
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float vrelu(float v) {
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(v));   // one instruction (fmaxf adds a canonicalising max)
  return r;
}
__device__ __forceinline__ f2 vrelu(f2 v) { return f2{vrelu(v.x), vrelu(v.y)}; }
__device__ __forceinline__ float vmax(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ f2 vmax(f2 a, f2 b) { return f2{fmaxf(a.x, b.x), fmaxf(a.y, b.y)}; }
__device__ __forceinline__ float vexp2(float a) { return fast_exp2(a); }
__device__ __forceinline__ f2 vexp2(f2 a) { return f2{fast_exp2(a.x), fast_exp2(a.y)}; }
__device__ __forceinline__ float vsplat(float, float v) { return v; }
__device__ __forceinline__ f2 vsplat(f2, float v) { return f2{v, v}; }

// T = float or f2.  (a, D, A, A2) = running max' / sum e / sum e*r / sum e*r^2 with r = relu(z) (or z).
template <typename T, int U, bool RELU, bool WITH_D, bool FULL>
__device__ __forceinline__ void softmax_fold(T& a, T& D, T& A, T& A2, const T (&z)[U], const bool (&ok)[U],
                                             float t2, float c0) {
  T r[U], s[U];
  const T vt2 = vsplat(a, t2), vc0 = vsplat(a, c0);
  T nm = a;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    r[u] = RELU ? vrelu(z[u]) : z[u];
    s[u] = r[u] * vt2 + vc0;
    if constexpr (!FULL) {
      if (u > 0) s[u] = ok[u] ? s[u] : vsplat(a, DGCN_NEG_INF);   // ok[0] holds (caller's guard)
    }
    nm = vmax(nm, s[u]);
  }
  const T sc = vexp2(a - nm);   // exp2(-inf) = 0 on the first batch
  D = D * sc;
  A = A * sc;
  if constexpr (WITH_D) A2 = A2 * sc;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const T e = vexp2(s[u] - nm);   // 0 for masked edges
    D = D + e;
    if constexpr (WITH_D) {
      const T er = e * r[u];
      A = A + er;
      A2 = er * r[u] + A2;
    } else {
      A = e * r[u] + A;
    }
  }
  a = nm;
}

// Fold U gathered rows (this lane's VEC channels of each) into the state.
template <int MODE, int VEC, int U, bool RELU, bool WITH_D, bool FULL>
__device__ __forceinline__ void accumulate(State<VEC>& st, const float (&v)[U][VEC],
                                           const bool (&ok)[U], const int (&eid)[U], float eps,
                                           float t2, float c0, float p) {
  if constexpr (!FULL) {
    if (!ok[0]) return;  // ok[] is monotone in u: nothing valid for this edge group
  }
  if constexpr (MODE == DGCN_AGGR_SOFTMAX) {
    if constexpr (VEC % 2 == 0) {
#pragma unroll
      for (int j = 0; j < VEC; j += 2) {
        f2 a = {st.a[j], st.a[j + 1]}, D = {st.b[j], st.b[j + 1]}, A = {st.c[j], st.c[j + 1]};
        f2 A2 = {st.d[j], st.d[j + 1]};
        f2 z[U];
#pragma unroll
        for (int u = 0; u < U; ++u) z[u] = f2{v[u][j], v[u][j + 1]};
        softmax_fold<f2, U, RELU, WITH_D, FULL>(a, D, A, A2, z, ok, t2, c0);
        st.a[j] = a.x; st.a[j + 1] = a.y;
        st.b[j] = D.x; st.b[j + 1] = D.y;
        st.c[j] = A.x; st.c[j + 1] = A.y;
        st.d[j] = A2.x; st.d[j + 1] = A2.y;
      }
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float z[U];
#pragma unroll
        for (int u = 0; u < U; ++u) z[u] = v[u][j];
        softmax_fold<float, U, RELU, WITH_D, FULL>(st.a[j], st.b[j], st.c[j], st.d[j], z, ok, t2, c0);
      }
    }
    return;
  }
  if constexpr (MODE == DGCN_AGGR_POWER && !WITH_D) {
    // p == 1, the reference's default (args: --p 1.0 without --learn_p): u^1 = u exactly, no log2 / exp2 per element
    if (p == 1.f) {   // wave-uniform
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (FULL || ok[u]) {
            const float m = RELU ? vrelu(v[u][j]) + eps : v[u][j];
            st.b[j] += fminf(fmaxf(m, kPowLo), kPowHi);
          }
        }
      }
      return;
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    if constexpr (MODE == DGCN_AGGR_POWER) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (FULL || ok[u]) {
          const float m = RELU ? vrelu(v[u][j]) + eps : v[u][j];
          const float uu = fminf(fmaxf(m, kPowLo), kPowHi);
          const float l2 = fast_log2(uu);
          const float up = fast_exp2(p * l2);
          st.b[j] += up;
          if constexpr (WITH_D) st.d[j] = fmaf(up, l2 * 0.6931471805599453f, st.d[j]);
        }
      }
    } else if constexpr (MODE == DGCN_AGGR_MAX) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (FULL || ok[u]) {
          const float m = RELU ? vrelu(v[u][j]) + eps : v[u][j];
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
        if (FULL || ok[u]) st.b[j] += RELU ? vrelu(v[u][j]) + eps : v[u][j];
      }
    }
  }
}

}  // namespace
}  // namespace dgcn
