// Shared pieces of the sparse generalized aggregation kernels (gen_aggr_fwd.hip / gen_aggr_bwd.hip): walk
// descriptors, the per-item software pipeline helpers, the fused edge encoder and the host-side layout choices.
// Everything lives in an anonymous namespace: each translation unit gets its own copy.
// Tuning-build switches (compile-time, not defined in the shipped build): DGCN_G (edge groups per row), DGCN_FWD_U
// (load batches in flight), DGCN_FWD_WPE / DGCN_FWD_WAVES_PER_CU (occupancy / grid), DGCN_NO_PREFETCH / DGCN_NO_PREFETCH_BWD (item
// look-ahead off; measured neutral on the products shape).  The defaults are the measured optimum on the products and arxiv shapes.
#pragma once

#include <stdlib.h>

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
  int add_root;         // out_i += x_i (the GENConv residual h = x + m fused into the epilogue)
  const float* enc_feat;  // EA == 2: raw edge features [E, kEncF] in original edge order; e_e = enc_w f_e + enc_b
  const float* enc_w;     // [C, kEncF] (nn.Linear weight)
  const float* enc_b;     // [C] or null
  int n_edges_hint;     // edges of the walk (layout choice only)
  float* ws;  // partial slots: [slot][4][C]
  int32_t* ticket;      // wave-uniform encoder walk: work-item counter (zeroed by the entry point before the launch)
};

struct BwdParams {
  WalkGraph g;      // transposed walk: rows = sources, col = destinations
  const float* x;
  int64_t x_stride;
  const float* ea;
  int C;
  int msg;
  int learn_t;
  int ea_is_z;      // EA == 1: the edge rows ARE z_e (saved by the fused edge-GEMM forward), x is not added
  float t, p, eps;
  const float* t_dev;
  const float* p_dev;
  const float* gcoef;
  const void* aux1;
  const float* out;
  const float* gshift;    // [n_dst, C] g_i * exp(kshift_c - L_i)  (single-gather softmax backward) or null
  const float* kshift;    // [C] per-channel shift
  const int32_t* shift_ok;  // device flag: the shifted form is numerically safe for this call iff *shift_ok != shift_bad
  int shift_bad;            // 0: flag means "ok"; 1: flag is the forward's range_flag (nonzero = NOT safe)
  const float* groot;     // [n_src, C] upstream gradient added to grad_x (backward of add_root) or null
  const float* enc_feat;  // EA == 2: see FwdParams
  const float* enc_w;
  const float* enc_b;
  float* enc_gpart;       // EA == 2: [gridDim.x][C][kEncF + 1] per-workgroup partial (dW | db)
  float* grad_x;
  float* grad_ea;
  int n_edges_hint;       // edges of the walk (layout choice only)
  // MAX fast path: per-edge arg-max bit masks [E][mask_words] indexed by CSR position (bit c of edge p set iff p is
  // the arg-max of its destination row in channel c); g.eperm then holds the CSR position of every CSC position
  const uint32_t* maxmask;
  int mask_words;
  float* ws;  // partial slots: [slot][C]
  int32_t* ticket;      // wave-uniform encoder walk: work-item counter (zeroed by the entry point before the launch)
};

// Work items are claimed from a counter instead of being dealt out by wave index: the items of a power-law graph differ
// by two orders of magnitude (1 .. 512 edges), a workgroup of four dealt waves lives as long as its longest item and the
// chip was 35 % occupied (SQ counters, profiles/r04_enc_*): with tickets every wave slot stays busy until the list is
// empty.  The next ticket is claimed while the current item is processed (its row bounds and first block are prefetched).
// Eight counters, one per XCD (blockIdx % 8), each handing out its own interleaved eighth of the items (item = t * 8 +
// xcd): same-address device atomics complete at ~10 ns each, and ONE counter serving the 27 k items of an ogbn-proteins
// cluster serialised the whole launch at 0.27 ms; eight counters on separate cache lines run side by side (3.4 k atomics
// each, spread over the launch), and an interleaved eighth of the items is as heavy as any other to a percent or two.
constexpr int kQueues = kNumXCD;
constexpr int kQueueStride = 32;          // ints between two counters (128 bytes)
struct ItemQueue {
  int32_t* cnt;
  int q, nq, t_static;
  __device__ __forceinline__ int claim() {
    if (cnt == nullptr) {                                // DGCN_FLAG_STATIC_ITEMS: wave w takes items w, w + W, ...
      const int it = t_static;
      t_static += nq;
      return it;
    }
    int t = 0;
    if (lane_id() == 0) t = atomicAdd(cnt, 1);
    return __builtin_amdgcn_readfirstlane(t) * nq + q;
  }
  __device__ __forceinline__ int first(int32_t* ticket) {
    if (ticket == nullptr) {
      cnt = nullptr;
      nq = static_cast<int>(gridDim.x) * kWavesPerWg;
      t_static = static_cast<int>(blockIdx.x) * kWavesPerWg + (threadIdx.x >> 6);
      q = 0;
      return claim();
    }
    nq = min(kQueues, static_cast<int>(gridDim.x));      // (a tiny launch: every queue needs a workgroup)
    q = blockIdx.x % nq;
    cnt = ticket + q * kQueueStride;
    return claim();
  }
  __device__ __forceinline__ int next(int32_t*) { return claim(); }
};
constexpr size_t kTicketBytes = kQueues * kQueueStride * sizeof(int32_t);

struct Work {
  int row, beg, end, slot;
};

// SW = lanes that share one work item.  SW == 64: the whole wave, everything wave-uniform (SGPRs).
// SW < 64 (narrow rows): 64/SW sub-groups walk different items side by side; an out-of-range item is empty.
template <int SW>
__device__ __forceinline__ int maybe_uni(int v) {
  if constexpr (SW == kWave) return uni(v);
  return v;
}

template <int SW>
__device__ __forceinline__ bool any_sub(bool c) {
  if constexpr (SW == kWave) return c;
  return __any(c);
}

template <int SW>
__device__ __forceinline__ bool all_sub(bool c) {
  if constexpr (SW == kWave) return c;
  return __all(c);
}

template <int SW>
__device__ __forceinline__ Work fetch_work(const WalkGraph& g, int item, int n_items) {
  Work w;
  if (item >= n_items) {   // past the end (sub-group tail, or the look-ahead of the last items): empty
    w.row = -1; w.beg = 0; w.end = 0; w.slot = -1;
    return w;
  }
  if (g.n_work) {
    w.row = maybe_uni<SW>(g.work_row[item]);
    w.beg = maybe_uni<SW>(g.work_beg[item]);
    w.end = maybe_uni<SW>(g.work_end[item]);
    w.slot = maybe_uni<SW>(g.work_slot[item]);
  } else {
    w.row = item;
    w.beg = maybe_uni<SW>(g.rowptr[item]);
    w.end = maybe_uni<SW>(g.rowptr[item + 1]);
    w.slot = -1;
  }
  return w;
}

// First/next block of <= SW column ids (and original edge ids) of an item, one per lane of the sub-group.
template <int SW, bool NEED_EID>
__device__ __forceinline__ void load_cols(const WalkGraph& g, const Work& w, int blk, int sl, int& col, int& eid) {
  col = 0;
  eid = 0;
  if (sl < w.end - blk) {
    col = g.col[blk + sl];
    if constexpr (NEED_EID) eid = g.eperm ? g.eperm[blk + sl] : blk + sl;
  }
}

// Block b runs on XCD b % 8 (observed dispatch order; used for L2 affinity only).  Remap so
// that, within one grid-stride sweep, each XCD covers contiguous RUNS of rows: rows that
// are adjacent in a locality-ordered graph then share their neighbours' lines in one L2.
// The runs (>= 64 blocks, at most 64 per XCD) are dealt to the XCDs in turn: with ONE range per XCD a graph whose
// heavy rows sit together (a degree-sorted power-law graph: the first eighth of the rows holds most of the edges)
// gives one XCD most of the work.  Measured: arxiv-shaped uniform graph forward 0.207 -> 0.183 ms, the degree-sorted
// ogbn-proteins-cluster-shaped graph 0.26 -> 0.25 ms (its time is not set by the XCD balance), products unchanged.
__device__ __forceinline__ int virtual_block() {
  const int per = gridDim.x / kNumXCD;  // gridDim.x is a multiple of 8
  const int xcd = blockIdx.x % kNumXCD, j = blockIdx.x / kNumXCD;
  int run = per / 64;
  if (run < 64) run = 64;
  const int whole = per / run * run;    // blocks of an XCD that fall into whole runs
  if (j < whole) return ((j / run) * kNumXCD + xcd) * run + (j % run);
  return whole * kNumXCD + (j - whole) * kNumXCD + xcd;
}

__device__ __forceinline__ float msg_apply(float z, int msg, float eps) {
  return msg == DGCN_MSG_RELU_EPS ? fmaxf(z, 0.f) + eps : z;
}

__device__ __forceinline__ float fast_pow(float u, float p) {  // u > 0
  return fast_exp2(p * fast_log2(u));
}

// ---- fused edge encoder (EA == 2): e_e = W f_e + b with kEncF raw features per edge -------------------------
// GENConv(encode_edge=True) builds edge_emb = Linear(edge_feat_dim -> C)(edge_attr), an (E, C) tensor written by
// a GEMM and read back by the aggregation (gcn_lib/sparse/torch_vertex.py:56-66).  With 8 raw features per edge
// (ogbn-proteins) the row is cheaper to recompute per edge from 32 bytes than to load as 4C bytes.
constexpr int kEncF = 8;

template <int VEC>
struct EncW {
  float w[VEC][kEncF];
  float b[VEC];
};

template <int VEC>
__device__ __forceinline__ void enc_load(EncW<VEC>& e, const float* __restrict__ W, const float* __restrict__ b,
                                         int c0, bool act) {
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    e.b[j] = (act && b) ? b[c0 + j] : 0.f;
#pragma unroll
    for (int f = 0; f < kEncF; ++f) e.w[j][f] = act ? W[(c0 + j) * kEncF + f] : 0.f;
  }
}

__device__ __forceinline__ void enc_feat_row(float (&fe)[kEncF], const float* __restrict__ feat, int eid) {
  const float4* p = reinterpret_cast<const float4*>(feat + static_cast<int64_t>(eid) * kEncF);
  const float4 a = p[0], b = p[1];
  fe[0] = a.x; fe[1] = a.y; fe[2] = a.z; fe[3] = a.w;
  fe[4] = b.x; fe[5] = b.y; fe[6] = b.z; fe[7] = b.w;
}

template <int VEC>
__device__ __forceinline__ void enc_apply(float (&out)[VEC], const EncW<VEC>& e, const float (&fe)[kEncF]) {
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    float a = e.b[j];
#pragma unroll
    for (int f = 0; f < kEncF; ++f) a = fmaf(e.w[j][f], fe[f], a);
    out[j] = a;
  }
}

// Address of row `src`: a 32x32->64 multiply (one v_mad_u64_u32); the host checks 0 <= stride < 2^31.
__device__ __forceinline__ const float* row_ptr(const float* base, int src, uint32_t stride) {
  return base + static_cast<uint64_t>(static_cast<uint32_t>(src)) * stride;
}

// EA: 0 = no edge features, 1 = dense (E, C) edge features, 2 = encoded on the fly from kEncF raw features
// ---------------------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------------------
int lanes_per_row(int C, int vec) {
  const int need = (C + vec - 1) / vec;
  int lpr = 4;
  while (lpr < need && lpr < kWave) lpr <<= 1;
  return lpr;
}

int round_up8(int v) { return (v + 7) / 8 * 8; }

// Edge groups per item (G): how many edges of ONE row a wave walks side by side; the remaining 64 / (LPR * G)
// sub-groups walk other rows.  DGCN_G overrides it at compile time (tuning builds).
#ifdef DGCN_G
constexpr int kEdgeGroups(int) { return DGCN_G; }
#else
constexpr int kEdgeGroups(int lpr) { return lpr >= 8 ? 2 : 4; }   // measured: products + arxiv shapes, C = 16..128
#endif
constexpr int kSubWidth(int lpr) { return lpr * kEdgeGroups(lpr) < kWave ? lpr * kEdgeGroups(lpr) : kWave; }

// Lanes per work item actually used for a walk over n_items rows: the narrow layout (several rows per wave) only
// when it still leaves >= kMinWaves waves (two per wave slot of the chip); a small graph (ogbn-proteins clusters,
// PPI) needs every row as its own wave.
constexpr int kMinWaves = 12288;
inline int subgroup_width(int lpr, int64_t n_items, int64_t n_edges = -1) {
  // 128-channel rows on a LOW-DEGREE graph (ogbn-arxiv: 14.7 edges per row): two rows per wave, one edge group each.
  // There the kernel is VALU-issue bound by per-row work (SQ_INSTS_VALU = 35 per edge against ~12 of fold; the
  // cross-group state merge, the epilogue and the item bookkeeping are paid per wave), so sharing a wave between two
  // rows and dropping the merge beats walking two edges of one row at once.  High-degree graphs keep one row per wave.
  if (lpr == 32 && n_edges >= 0 && n_edges < 32 * n_items && n_items * 32 / kWave >= kMinWaves) return 32;
  const int sw = kSubWidth(lpr);
  return (n_items * sw / kWave >= kMinWaves) ? sw : kWave;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }


struct EncArgs {
  const float* feat;
  const float* w;
  const float* b;
  int n_feat;
};

constexpr int kEncMaxParts = 1024;   // workgroups (= partial dW|db blocks) of the encoded backward

inline int enc_check(const EncArgs* enc, int channels) {
  if (!enc) return DGCN_OK;
  if (!enc->feat || !enc->w) return DGCN_E_NULL;
  if (enc->n_feat != kEncF || channels % 4 != 0 || channels > 256) return DGCN_E_SHAPE;
  if (!aligned16(enc->feat) || !aligned16(enc->w)) return DGCN_E_ALIGN;
  return DGCN_OK;
}

}  // namespace
}  // namespace dgcn
