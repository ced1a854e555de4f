// Backward of the fused edge encoder under MAX aggregation, without the (E, C) gradient.
//
// Reference: GENConv.forward with encode_edge=True, gcn_lib/sparse/torch_vertex.py:62-66 (edge_emb = edge_encoder(f),
// message relu(x_j + edge_emb) + eps, :78-85) aggregated with scatter(reduce='max') (gcn_lib/sparse/torch_message.py:46-47);
// autograd then forms dz = dL/d(edge_emb) (E, C), dF = dz W (E, K) and dW = dz^T F (C, K).
//
// Under max, dz is one number per (destination row, channel): dz[e][c] = g[r][c] iff e is the arg-max edge of (r, c)
// (the forward stores the ORIGINAL edge id, -1 where no neighbour passed the relu), i.e. n_dst * C non-zeros in an
// E x C matrix (1.7 % at the ogbn-proteins cluster shape).  So instead of writing dz and running two dense
// E x C x K products over it (the round-2/3 path: 354 MB written, read twice, 2 x 39.7 GFLOP on the matrix pipe), two
// kernels walk the winners, n_dst * C * K plain fp32 FMAs each:
//     dF[e][:] += sum_{c won by e} g[r][c] W[c][:]      egemm_max_bwd_feat_kernel: one read-modify-write of the edge's
//                                                       4K-byte row (e belongs to exactly one destination row: one
//                                                       wave, one visit); edges that win nothing are never touched
//     dW[c][:]  = sum_r g[r][c] F[arg[r][c]][:]         egemm_max_bwd_weight_kernel: per-workgroup partials [grid][C][K]
// Lane l owns features 4l .. 4l+3 (K <= 256).  Work is balanced by rows (C winners each) whatever the degree
// distribution.  Both are deterministic (no atomics, fixed summation orders).
#include "dgcn_common.h"

namespace dgcn {
namespace {

typedef float f4v __attribute__((ext_vector_type(4)));

constexpr int kMbWaves = 16;       // waves per workgroup: the C x K array takes most of the LDS, one workgroup per CU
constexpr int kMbEdges = 8;        // winning edges whose rows are in flight per wave
constexpr int kMbMinRows = 16;     // destination rows per workgroup at least (one per wave)

struct MaxBwdParams {
  const float* g;            // [n_rows][C]
  const int32_t* arg;        // [n_rows][C] original edge id or -1
  int n_rows, C, K, rows_per_wg;
  int n_edges;               // ids outside [0, n_edges) are treated as -1 (a stale id must not become an address)
  const float* feat;         // [E][K], row stride feat_stride
  int64_t feat_stride;
  const float* w;            // [C][K]
  float* gfeat;              // [E][K] accumulated in place, row stride gfeat_stride; or null
  int64_t gfeat_stride;
  float* wpart;              // [grid][C][K]; or null
};

__device__ __forceinline__ int first_bit(uint64_t m) { return __builtin_ctzll(m); }
__device__ __forceinline__ float lane_value(float v, int l) {          // l wave-uniform
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// dF[e][:] += sum_{c in M_e} g[r][c] W[c][:].  W staged in LDS (element (c, 4l + j) at c K + j (K/4) + l: lane l reads
// K/4-strided words, consecutive lanes consecutive banks), the distinct winning edges of a destination row peeled off
// the two id registers per lane, kMbEdges gradient rows in flight per wave.  No atomics: deterministic.
__global__ __launch_bounds__(kMbWaves * kWave) void egemm_max_bwd_feat_kernel(const MaxBwdParams P) {
  extern __shared__ __attribute__((aligned(16))) float lds[];     // [C][4][K/4]
  const int C = P.C, K = P.K, KQ = P.K >> 2;
  for (int i = threadIdx.x; i < C * KQ; i += blockDim.x) {
    const int c = i / KQ, l = i - c * KQ;
    const f4v v = *reinterpret_cast<const f4v*>(P.w + c * K + 4 * l);
#pragma unroll
    for (int j = 0; j < 4; ++j) lds[c * K + j * KQ + l] = v[j];
  }
  __syncthreads();
  const int lane = lane_id();
  const int wave = uni(static_cast<int>(threadIdx.x >> 6));
  const int k0 = lane * 4;
  const bool kact = lane < KQ;
  const int r_beg = blockIdx.x * P.rows_per_wg;
  const int r_end = min(r_beg + P.rows_per_wg, P.n_rows);
  for (int r = r_beg + wave; r < r_end; r += kMbWaves) {
    const int64_t ro = static_cast<int64_t>(r) * C;
    const bool c0ok = lane < C, c1ok = lane + kWave < C;
    int a0 = c0ok ? P.arg[ro + lane] : -1;
    int a1 = c1ok ? P.arg[ro + lane + kWave] : -1;
    if (static_cast<uint32_t>(a0) >= static_cast<uint32_t>(P.n_edges)) a0 = -1;
    if (static_cast<uint32_t>(a1) >= static_cast<uint32_t>(P.n_edges)) a1 = -1;
    const float g0 = c0ok ? P.g[ro + lane] : 0.f;
    const float g1 = c1ok ? P.g[ro + lane + kWave] : 0.f;
    uint64_t p0 = __ballot(a0 >= 0), p1 = __ballot(a1 >= 0);      // channels whose winner is still to be visited
    while (p0 | p1) {
      int e[kMbEdges];
      uint64_t m0[kMbEdges], m1[kMbEdges];
      f4v row[kMbEdges];
#pragma unroll
      for (int u = 0; u < kMbEdges; ++u) {
        e[u] = -1;
        m0[u] = 0; m1[u] = 0;
        row[u] = f4v{0.f, 0.f, 0.f, 0.f};
        if (p0 | p1) {
          const int id = p0 ? __builtin_amdgcn_readlane(a0, first_bit(p0)) : __builtin_amdgcn_readlane(a1, first_bit(p1));
          m0[u] = __ballot(a0 == id);        // (visited channels hold other ids, dead channels -1: no false hits)
          m1[u] = __ballot(a1 == id);
          p0 &= ~m0[u];
          p1 &= ~m1[u];
          e[u] = id;
          if (kact) row[u] = *reinterpret_cast<const f4v*>(P.gfeat + static_cast<int64_t>(id) * P.gfeat_stride + k0);
        }
      }
#pragma unroll
      for (int u = 0; u < kMbEdges; ++u) {
        if (e[u] < 0) continue;                      // wave-uniform
        f4v acc = row[u];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          uint64_t mm = half ? m1[u] : m0[u];
          while (mm) {
            const int l = first_bit(mm);
            mm &= mm - 1;
            const float gc = lane_value(half ? g1 : g0, l);
            const float* wc = lds + (half * kWave + l) * K + lane;
            if (kact) {
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[j] += gc * wc[j * KQ];
            }
          }
        }
        if (kact) *reinterpret_cast<f4v*>(P.gfeat + static_cast<int64_t>(e[u]) * P.gfeat_stride + k0) = acc;
      }
    }
  }
}

// dW[c][:] = sum_r g[r][c] F[arg[r][c]][:], owner-computes: wave w of the workgroup owns the channels w, w + 16, ...
// (at most kMbChan of them: C <= 128) and keeps their K-float sums in registers (lane l: features 4l .. 4l+3).  The
// workgroup's 16 waves walk its destination rows TOGETHER, one row at a time: a winning edge wins ~3 channels of its
// row, i.e. its feature row is wanted by ~3 waves -- with the whole workgroup inside the same row those requests fall
// into one L2 residency.  Measured fetch per launch at the cluster shape (FETCH_SIZE x 2): 0.47 GB = the distinct
// winning rows; every wave sweeping 64 rows per channel: 1.21 GB (0.200 ms); two rows per step: 0.86 GB; this form
// 0.147 ms, bound by the latency of the kMbChan rows a wave has in flight (a rolling reload of the registers pair by
// pair would keep them in flight across rows; the compiler's waitcnt placement drained it, not pursued).
// Per row a wave has kMbChan (row, channel) pairs: lane p loads id / gradient of pair p (the next row's are requested
// before this row's feature rows).  No LDS, no atomics; every workgroup writes its whole [C][K] block: fixed summation
// order, bit-reproducible.
constexpr int kMbChan = 8;
constexpr int kMbChunk = 1;

__global__ __launch_bounds__(kMbWaves * kWave) void egemm_max_bwd_weight_kernel(const MaxBwdParams P) {
  constexpr int NP = kMbChan * kMbChunk;          // pairs per chunk: pair p = ci * kMbChunk + j  (channel slot ci, row j)
  const int C = P.C, K = P.K;
  const int lane = lane_id();
  const int wave = uni(static_cast<int>(threadIdx.x >> 6));
  const int k0 = lane * 4;
  const bool kact = k0 < K;
  const int r_beg = blockIdx.x * P.rows_per_wg;
  const int r_end = min(r_beg + P.rows_per_wg, P.n_rows);
  f4v acc[kMbChan];
#pragma unroll
  for (int ci = 0; ci < kMbChan; ++ci) acc[ci] = f4v{0.f, 0.f, 0.f, 0.f};
  const int pc = wave + kMbWaves * (lane / kMbChunk);      // this lane's pair: channel ...
  const int pj = lane % kMbChunk;                          // ... and row within the chunk
  auto load_pair = [&](int rc, int& id, float& gv) {
    const int r = rc + pj;
    const bool ok = lane < NP && pc < C && r < r_end;
    id = ok ? P.arg[static_cast<int64_t>(r) * C + pc] : -1;
    if (static_cast<uint32_t>(id) >= static_cast<uint32_t>(P.n_edges)) id = -1;
    gv = ok ? P.g[static_cast<int64_t>(r) * C + pc] : 0.f;
  };
  int id, idn;
  float gv, gvn;
  load_pair(r_beg, id, gv);
  for (int rc = r_beg; rc < r_end; rc += kMbChunk) {
    load_pair(rc + kMbChunk, idn, gvn);                    // (past the end: all lanes off)
    f4v row[NP];
    float gg[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int e = __builtin_amdgcn_readlane(id, p);
      gg[p] = lane_value(gv, p);
      row[p] = f4v{0.f, 0.f, 0.f, 0.f};
      if (e >= 0 && kact) {                                // e wave-uniform
        row[p] = *reinterpret_cast<const f4v*>(P.feat + static_cast<int64_t>(e) * P.feat_stride + k0);
      }
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) acc[p / kMbChunk] += gg[p] * row[p];
    id = idn;
    gv = gvn;
  }
  float* out = P.wpart + static_cast<int64_t>(blockIdx.x) * C * K;
#pragma unroll
  for (int ci = 0; ci < kMbChan; ++ci) {
    const int c = wave + kMbWaves * ci;
    if (c < C && kact) *reinterpret_cast<f4v*>(out + c * K + k0) = acc[ci];
  }
}

inline int mb_rows_per_wg(int n_dst) {
  const int even = (n_dst + num_cus() - 1) / num_cus();
  return even > kMbMinRows ? even : kMbMinRows;
}

inline bool aligned16p(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace
}  // namespace dgcn

using namespace dgcn;

extern "C" int32_t dgcn_egemm_max_bwd_num_partials(int32_t n_dst) {
  if (n_dst <= 0) return 0;
  const int rpw = mb_rows_per_wg(n_dst);
  return (n_dst + rpw - 1) / rpw;
}

extern "C" int dgcn_egemm_max_bwd_f32(const float* gcoef, const int32_t* argmax, int32_t n_dst, int32_t n_edges,
                                      const float* edge_feat, int64_t feat_stride, const float* enc_weight,
                                      int32_t n_feat, int32_t channels, float* grad_feat, int64_t grad_feat_stride,
                                      float* grad_w_partials, void* stream) {
  if (!gcoef || !argmax || !enc_weight) return DGCN_E_NULL;
  if (grad_w_partials && !edge_feat) return DGCN_E_NULL;
  if (n_dst < 0 || n_edges < 0 || channels <= 0 || channels > 2 * kWave) return DGCN_E_SHAPE;
  if (n_feat <= 0 || n_feat % 4 != 0 || n_feat > 4 * kWave) return DGCN_E_SHAPE;
  if (grad_w_partials && (feat_stride < n_feat || feat_stride % 4 != 0)) return DGCN_E_SHAPE;
  if (grad_feat && (grad_feat_stride < n_feat || grad_feat_stride % 4 != 0)) return DGCN_E_SHAPE;
  if (!aligned16p(enc_weight) || (edge_feat && !aligned16p(edge_feat)) || (grad_feat && !aligned16p(grad_feat)) ||
      (grad_w_partials && !aligned16p(grad_w_partials))) {
    return DGCN_E_ALIGN;
  }
  if (n_dst == 0 || (!grad_feat && !grad_w_partials)) return DGCN_OK;
  MaxBwdParams P;
  P.g = gcoef; P.arg = argmax; P.n_rows = n_dst; P.C = channels; P.K = n_feat; P.n_edges = n_edges;
  P.rows_per_wg = mb_rows_per_wg(n_dst);
  P.feat = edge_feat; P.feat_stride = feat_stride; P.w = enc_weight;
  P.gfeat = grad_feat; P.gfeat_stride = grad_feat_stride; P.wpart = grad_w_partials;
  const int grid = dgcn_egemm_max_bwd_num_partials(n_dst);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (grad_feat) {
    const size_t lds = static_cast<size_t>(channels) * n_feat * sizeof(float);
    if (lds > 64 * 1024) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(egemm_max_bwd_feat_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
      if (e != hipSuccess) return static_cast<int>(e);
    }
    hipLaunchKernelGGL(egemm_max_bwd_feat_kernel, dim3(grid), dim3(kMbWaves * kWave), lds, s, P);
  }
  if (grad_w_partials) {
    hipLaunchKernelGGL(egemm_max_bwd_weight_kernel, dim3(grid), dim3(kMbWaves * kWave), 0, s, P);
  }
  return launch_status();
}
