// Composition of the model-level and the per-layer edge encoder into one Linear(F -> C), and its backward.
//
// Reference: edge_emb = self.edge_encoder(edge_attr) -- Linear(F = 8 -> hidden), examples/ogb_eff/ogbn_proteins/
// model_rev.py:98 -- followed in every GENConv by edge_encoder = Linear(hidden -> C), gcn_lib/sparse/torch_vertex.py:62-66,
// with nothing in between:  W' = W_l We  (C x F),  b' = W_l b_e + b_l  (C).  blocks.ComposedEdgeEmbedding forms (W', b')
// per layer and the aggregation kernels evaluate W' f_e + b' per edge (dgcn_gen_aggr_enc_*).  As torch ops that is a
// matmul + an addmv in the forward (twice per step in the reversible pattern) and four small GEMM / GEMV launches plus
// their accumulations in the backward, per coupling function: 128 of the 880 launches of an eager RevGCN-8 step
// (benchmarks/launch_census.py).  Here: one launch each way.  The products are 200 k multiply-adds: plain fp32 FMAs,
// one wave per output row, fixed summation order (bit-reproducible).
#include "dgcn_common.h"

namespace dgcn {
namespace {

constexpr int kCmpF = 16;      // raw edge features supported (F <= 16; the reference's models have 8)

// v where ok, +0 otherwise, as integer bits: a select the compiler cannot turn back into a branch around the load that
// produced v.  (`if (f < F) acc = fma(w, We[..], acc)` compiled to a branch per element with a full wait behind each
// load -- 64 dependent L2 round trips per wave, 9 us for 200 k multiply-adds; with clamped addresses the 17 loads of an
// iteration are in flight together.)
__device__ __forceinline__ float cmp_keep(float v, bool ok) {
  return __uint_as_float(__float_as_uint(v) & (ok ? 0xFFFFFFFFu : 0u));
}

// W'[c][f] = sum_h Wl[c][h] We[h][f],  b'[c] = sum_h Wl[c][h] be[h] + bl[c].  One wave per row c, lanes over h.
__global__ __launch_bounds__(kWave) void enc_compose_fwd_kernel(const float* __restrict__ Wl, const float* __restrict__ bl,
                                                                const float* __restrict__ We, const float* __restrict__ be,
                                                                int C, int H, int F, float* __restrict__ Wc,
                                                                float* __restrict__ bc) {
  const int c = blockIdx.x, lane = threadIdx.x;
  float acc[kCmpF + 1];
#pragma unroll
  for (int f = 0; f <= kCmpF; ++f) acc[f] = 0.f;
  for (int h = lane; h < H; h += kWave) {
    const float w = Wl[static_cast<int64_t>(c) * H + h];
    float e[kCmpF + 1];
#pragma unroll
    for (int f = 0; f < kCmpF; ++f) e[f] = cmp_keep(We[static_cast<int64_t>(h) * F + min(f, F - 1)], f < F);
    e[kCmpF] = cmp_keep((be ? be : Wl)[h], be != nullptr);       // (Wl: any readable address of >= H floats)
#pragma unroll
    for (int f = 0; f <= kCmpF; ++f) acc[f] = fmaf(w, e[f], acc[f]);
  }
#pragma unroll
  for (int f = 0; f <= kCmpF; ++f) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc[f] += __shfl_xor(acc[f], off);
  }
  if (lane == 0) {
#pragma unroll
    for (int f = 0; f < kCmpF; ++f) {
      if (f < F) Wc[static_cast<int64_t>(c) * F + f] = acc[f];
    }
    if (bc) bc[c] = acc[kCmpF] + (bl ? bl[c] : 0.f);
  }
}

// blocks [0, C):      dWl[c][h] = sum_f dWc[c][f] We[h][f] + dbc[c] be[h]            (threads over h)
// blocks [C, C + H):  dWe[h][f] = sum_c Wl[c][h] dWc[c][f],  dbe[h] = sum_c Wl[c][h] dbc[c]   (one wave per h, lanes over c)
__global__ __launch_bounds__(kWave) void enc_compose_bwd_kernel(const float* __restrict__ Wl, const float* __restrict__ We,
                                                                const float* __restrict__ be, const float* __restrict__ dWc,
                                                                const float* __restrict__ dbc, int C, int H, int F,
                                                                float* __restrict__ dWl, float* __restrict__ dWe,
                                                                float* __restrict__ dbe) {
  const int lane = threadIdx.x;
  if (static_cast<int>(blockIdx.x) < C) {
    const int c = blockIdx.x;
    if (!dWl) return;
    float g[kCmpF];
#pragma unroll
    for (int f = 0; f < kCmpF; ++f) g[f] = cmp_keep(dWc[static_cast<int64_t>(c) * F + min(f, F - 1)], f < F);
    const float gb = (dbc && be) ? dbc[c] : 0.f;
    for (int h = lane; h < H; h += kWave) {
      float e[kCmpF];
#pragma unroll
      for (int f = 0; f < kCmpF; ++f) e[f] = We[static_cast<int64_t>(h) * F + min(f, F - 1)];   // (g[f] = 0 past F)
      float a = (be ? gb * be[h] : 0.f);
#pragma unroll
      for (int f = 0; f < kCmpF; ++f) a = fmaf(g[f], e[f], a);
      dWl[static_cast<int64_t>(c) * H + h] = a;
    }
    return;
  }
  const int h = blockIdx.x - C;
  if (!dWe) return;
  float acc[kCmpF + 1];
#pragma unroll
  for (int f = 0; f <= kCmpF; ++f) acc[f] = 0.f;
  for (int c = lane; c < C; c += kWave) {
    const float w = Wl[static_cast<int64_t>(c) * H + h];
    float e[kCmpF + 1];
#pragma unroll
    for (int f = 0; f < kCmpF; ++f) e[f] = cmp_keep(dWc[static_cast<int64_t>(c) * F + min(f, F - 1)], f < F);
    e[kCmpF] = cmp_keep((dbc ? dbc : dWc)[c], dbc != nullptr);    // (dWc: any readable address of >= C floats)
#pragma unroll
    for (int f = 0; f <= kCmpF; ++f) acc[f] = fmaf(w, e[f], acc[f]);
  }
#pragma unroll
  for (int f = 0; f <= kCmpF; ++f) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc[f] += __shfl_xor(acc[f], off);
  }
  if (lane == 0) {
#pragma unroll
    for (int f = 0; f < kCmpF; ++f) {
      if (f < F) dWe[static_cast<int64_t>(h) * F + f] = acc[f];
    }
    if (dbe) dbe[h] = acc[kCmpF];
  }
}

}  // namespace
}  // namespace dgcn

using namespace dgcn;

extern "C" int dgcn_enc_compose_fwd_f32(const float* layer_w, const float* layer_b, const float* enc_w, const float* enc_b,
                                        int32_t channels, int32_t hidden, int32_t n_feat, float* out_w, float* out_b,
                                        void* stream) {
  if (!layer_w || !enc_w || !out_w) return DGCN_E_NULL;
  if (channels <= 0 || hidden <= 0 || n_feat <= 0 || n_feat > kCmpF) return DGCN_E_SHAPE;
  if ((layer_b || enc_b) && !out_b) return DGCN_E_NULL;
  hipLaunchKernelGGL(enc_compose_fwd_kernel, dim3(channels), dim3(kWave), 0, static_cast<hipStream_t>(stream), layer_w,
                     layer_b, enc_w, enc_b, channels, hidden, n_feat, out_w, out_b);
  return launch_status();
}

extern "C" int dgcn_enc_compose_bwd_f32(const float* layer_w, const float* enc_w, const float* enc_b, const float* grad_w,
                                        const float* grad_b, int32_t channels, int32_t hidden, int32_t n_feat,
                                        float* grad_layer_w, float* grad_enc_w, float* grad_enc_b, void* stream) {
  if (!layer_w || !enc_w || !grad_w) return DGCN_E_NULL;
  if (channels <= 0 || hidden <= 0 || n_feat <= 0 || n_feat > kCmpF) return DGCN_E_SHAPE;
  if (grad_enc_b && !grad_b) return DGCN_E_NULL;
  if (!grad_layer_w && !grad_enc_w) return DGCN_OK;
  hipLaunchKernelGGL(enc_compose_bwd_kernel, dim3(channels + hidden), dim3(kWave), 0, static_cast<hipStream_t>(stream),
                     layer_w, enc_w, enc_b, grad_w, grad_b, channels, hidden, n_feat, grad_layer_w, grad_enc_w, grad_enc_b);
  return launch_status();
}
