// fp32-faithful GEMM products on the bf16 matrix pipe ("bf16x6"), used by the fused edge-GEMM aggregation
// (gen_aggr_egemm.hip).
//
// Every fp32 operand is split EXACTLY into three bf16 values by truncation (hi = top 16 bits, mid = top 16 bits of the
// remainder, lo = top 16 bits of what is left: 3 x 8 = 24 significand bits, f == hi + mid + lo bit for bit) and a*b is
// accumulated in fp32 from the six largest cross terms  a1 b1 + a1 b2 + a2 b1 + a1 b3 + a3 b1 + a2 b2  (the dropped
// ones are <= 3 * 2^-24 |a b|: the level of fp32 rounding; measured max error / sum|a||b| = 1.7e-7 against 3.3e-7 for a
// plain fp32 GEMM).  v_mfma_f32_16x16x32_bf16 runs on the matrix pipe, 16x the fp32-MFMA rate, and overlaps with VALU work.
#pragma once

#include "dgcn_common.h"

namespace dgcn {

typedef float f4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
typedef int i4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned eg_pack_hi16(float e0, float e1) {   // (top 16 bits of e0) | (top 16 bits of e1) << 16
  return __builtin_amdgcn_perm(__float_as_uint(e1), __float_as_uint(e0), 0x07060302u);
}
__device__ __forceinline__ float eg_top16(float v) { return __uint_as_float(__float_as_uint(v) & 0xffff0000u); }

// eight fp32 values -> three bf16x8 MFMA fragments (element e in the low/high half of register e / 2)
__device__ __forceinline__ void eg_split3(const f4v& f0, const f4v& f1, i4v& h, i4v& m, i4v& l) {
  const float v[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
  float r[8], r2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    r[e] = v[e] - eg_top16(v[e]);
    r2[e] = r[e] - eg_top16(r[e]);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    h[e] = static_cast<int>(eg_pack_hi16(v[2 * e], v[2 * e + 1]));
    m[e] = static_cast<int>(eg_pack_hi16(r[2 * e], r[2 * e + 1]));
    l[e] = static_cast<int>(eg_pack_hi16(r2[2 * e], r2[2 * e + 1]));
  }
}

__device__ __forceinline__ f4v eg_mfma_bf16(const i4v& a, const i4v& b, const f4v& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, a), __builtin_bit_cast(bf8v, b), c, 0, 0, 0);
}

}  // namespace dgcn
