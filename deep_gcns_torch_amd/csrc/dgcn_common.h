// Shared device/host helpers for libdgcn (gfx950 only; wave = 64 lanes).
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "dgcn.h"

namespace dgcn {

constexpr int kWave = 64;
constexpr int kWgThreads = 256;           // 4 waves, one per SIMD
constexpr int kWavesPerWg = kWgThreads / kWave;
constexpr int kNumCU = 256;               // MI355X (SPX mode): the fallback of num_cus() below.  GRID CAPS and partial-buffer sizing only: every kernel
                                          // strides over its work, so on a partitioned (CPX: 32 CUs) device the grids are merely
                                          // larger than needed -- results are unaffected.  A compile-time constant on purpose: the
                                          // *_num_partials sizing calls and the launches must agree whatever device is current.
constexpr int kNumXCD = 8;                // XCDs of the SPX device: sizes the per-XCD work queues in DEVICE code (arrays), so it stays a
                                          // constant; on a CPX partition (one XCD) the eight queues are merely drained by one die

// Compute units of the device the process runs on, read ONCE (first use; function-local static = thread-safe) from
// hipGetDeviceProperties of the then-current device -- SURVEY.md 8(b)'s call_once cache.  Grid caps and partial-buffer
// sizes use it, so that the *_num_partials sizing calls and the launches agree whatever device is current later; without
// a device (the CPU-only build container: tests/test_abi.py calls the sizing functions there) it is kNumCU.
inline int num_cus() {
  static const int n = [] {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
        prop.multiProcessorCount <= 0) {
      (void)hipGetLastError();                 // (no device: not an error of the caller's launch)
      return kNumCU;
    }
    return prop.multiProcessorCount;
  }();
  return n;
}

#define DGCN_NEG_INF (-__builtin_inff())

// Launch-error helper: kernel launches are asynchronous; hipGetLastError() reports
// configuration errors (bad grid, missing code object) without synchronising.
inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? DGCN_OK : static_cast<int>(e);
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

// Force a wave-uniform value into an SGPR so address math on it is scalar.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

template <int VEC>
__device__ __forceinline__ void load_vec(float (&r)[VEC], const float* __restrict__ p) {
  if constexpr (VEC == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
  } else if constexpr (VEC == 2) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    r[0] = t.x; r[1] = t.y;
  } else {
    r[0] = *p;
  }
}

template <int VEC>
__device__ __forceinline__ void store_vec(float* __restrict__ p, const float (&r)[VEC]) {
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(r[0], r[1], r[2], r[3]);
  } else if constexpr (VEC == 2) {
    *reinterpret_cast<float2*>(p) = make_float2(r[0], r[1]);
  } else {
    *p = r[0];
  }
}

template <int VEC>
__device__ __forceinline__ void store_vec_i(int32_t* __restrict__ p, const int (&r)[VEC]) {
  if constexpr (VEC == 4) {
    *reinterpret_cast<int4*>(p) = make_int4(r[0], r[1], r[2], r[3]);
  } else if constexpr (VEC == 2) {
    *reinterpret_cast<int2*>(p) = make_int2(r[0], r[1]);
  } else {
    *p = r[0];
  }
}

template <int VEC>
__device__ __forceinline__ void load_vec_i(int (&r)[VEC], const int32_t* __restrict__ p) {
  if constexpr (VEC == 4) {
    const int4 t = *reinterpret_cast<const int4*>(p);
    r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
  } else if constexpr (VEC == 2) {
    const int2 t = *reinterpret_cast<const int2*>(p);
    r[0] = t.x; r[1] = t.y;
  } else {
    r[0] = *p;
  }
}

// Hardware transcendentals (v_exp_f32 / v_log_f32 are base-2, ~1 ulp).  Used instead of libm
// calls on the hot paths; accuracy is far inside the 1e-4 relative parity budget.
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }

// Zero-fill of counters the next kernel of the same call claims work from / accumulates into -- as a KERNEL, never
// hipMemsetAsync.  Round 5 finding: with hipMemsetAsync (a memset NODE in a captured graph) ~10 % of the
// REPLAYED launches of the per-edge encoder kernels found the work-item counters of a few queues non-zero at their first
// claims (ROCm 7.2, MI355X; consecutive layers get the same workspace block, i.e. the same counters): item 0 of those
// queues -- rows 0 .. 2 of the ogbn-proteins cluster -- was never processed, its arg-max ids were whatever the block held
// before, and the max backward that uses the ids as addresses faulted.  8,026 stale ids per bench run with the memset
// node, 0 with this kernel; eager launches were never affected.
static __global__ __launch_bounds__(256) void zero_words_kernel(uint32_t* __restrict__ p, size_t n_words) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n_words; i += stride) p[i] = 0u;
}

static inline int zero_async(void* p, size_t bytes, hipStream_t s) {        // bytes % 4 == 0, p 4-byte aligned
  const size_t n = bytes / 4;
  if (n == 0) return DGCN_OK;
  size_t blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(zero_words_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, static_cast<uint32_t*>(p), n);
  return launch_status();                      // (the hipMemsetAsync this replaced returned its error too: ADVICE r5)
}

// Grid size for a wave-per-item kernel: enough workgroups to keep every CU's 32 wave
// slots busy, capped so very large inputs grid-stride instead of launching millions
// of tiny workgroups.
inline int grid_for_waves(int64_t n_items, int waves_per_cu_target = 32) {
  int64_t wgs = (n_items + kWavesPerWg - 1) / kWavesPerWg;
  const int64_t cap = static_cast<int64_t>(num_cus()) * waves_per_cu_target / kWavesPerWg * 4;
  if (wgs > cap) wgs = cap;
  if (wgs < 1) wgs = 1;
  return static_cast<int>(wgs);
}

}  // namespace dgcn
