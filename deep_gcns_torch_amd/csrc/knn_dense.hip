// Dense kNN graph build for gfx950: fused pairwise distance + exact top-K select + dilation.
//
// Replaces  pairwise_distance / dense_knn_matrix / DenseDilated
//           (gcn_lib/dense/torch_edge.py:32-42, 45-58, 19-29)  and the ATen bmm + topk under them,
// without ever materialising the (B,N,N) distance matrix (537 MB/layer at B=8, N=4096).
//
// One workgroup (8 waves) owns TM query points of one sample:
//   phase 1  all 512 threads: D[r][j] = (|x_r|^2 + (-2 <x_r,x_j>)) + |x_j|^2  for every candidate j,
//            fp32, same association as the reference; the inner product is a channel-ordered fma
//            chain.  x is channel-major (B,C,N): for a fixed channel, consecutive lanes read
//            consecutive points (coalesced), each loaded value is reused for TM rows x 1 column
//            and JJ columns are register-blocked per thread.  The TM x N distance strip stays in LDS.
//   phase 2  wave r owns row r: exact K-th smallest key by 32-step bitwise bisection over the row
//            (64 keys per lane live in registers; one ballot-free wave reduction per step);
//   phase 3  ordered compaction of the K winners (ties at the threshold: lowest index first);
//   phase 4  in-register bitonic sort of the <=512 winners as u64 (key<<32 | index);
//   phase 5  emit positions 0, d, 2d, ... as int64 (dilation fused), plus the centre ids.
// Selection is exact: the output is the ascending-distance order of torch.topk(-D, K) wherever
// distances are distinct; equal distances are ordered by index (torch leaves that order open).

#include "bf16x6.h"

namespace dgcn {
namespace {

constexpr int kKnnThreads = 512;
constexpr int kKnnWaves = kKnnThreads / kWave;  // 8
constexpr int kMaxPerLane = 64;                 // N <= 64 * 64 = 4096 points per cloud
constexpr int kLdsBudget = 160 * 1024 - 1024;

struct KnnParams {
  const float* x;
  int64_t sb, sc, sn;  // strides in floats of (B, C, N)
  int B, C, N, K, dilation, Kout;
  int* redo;           // null, or [1 + B*N]: redo[0] = number of rows the filter kernel could not finish, redo[1..] =
                       // their flat ids b*N+i (any order).  Exact kernel: when non-null, one listed row per tile.
  const i4v* planes;   // bf16 filter path: [3][B*N][C/8] units of 8 bf16 -- the three exact bf16 planes of every point,
                       // point-major (written by knn_planes_kernel), or null
  float* sqnorm;       // [B*N] |x_j|^2 (fma chain over channels), written by knn_prep_kernel for the filter pass
  uint32_t* tau;       // [B*N] per-row sample threshold (ordered-uint key), written by knn_prep_kernel
  uint2* lists;        // knn_filter2_kernel: [B*N][kF2Cap] (key, id) candidate lists in global memory, or null
  int* list_cnt;       // [B*N] candidates appended per row (may exceed the capacity: the row is then redone)
  int exclude_self;    // 1: the query point itself is never a neighbour (torch_cluster.knn_graph, loop=False)
  int sample_rank;     // rank of the sample threshold used by the candidate pre-filter (0 = disabled)
  int64_t* nn_out;     // [B, N, Kout] neighbour ids
  int64_t* ctr_out;    // [B, N, Kout] centre ids (may be null)
};

__device__ __forceinline__ uint32_t key_of(float d) {
  const uint32_t b = __float_as_uint(d);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);  // monotone: float order == unsigned order
}

// wave-wide sum of a per-lane int, returned wave-uniform.  DPP row shifts + row broadcasts (VALU data
// path, ~8 cycles per step) instead of ds_bpermute shuffles (~100+ cycles each on a dependent chain):
// the 32-step bisection below calls this once per step.
__device__ __forceinline__ int wave_sum(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);  // row_shr:8  -> inclusive scan per 16-lane row
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);  // row_bcast:15 into rows 1,3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);  // row_bcast:31 into rows 2,3
  return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int off) {
  uint32_t lo = static_cast<uint32_t>(v), hi = static_cast<uint32_t>(v >> 32);
  lo = __shfl_xor(static_cast<int>(lo), off);
  hi = __shfl_xor(static_cast<int>(hi), off);
  return (static_cast<unsigned long long>(hi) << 32) | lo;
}

// Bitonic sort of R*64 u64 values, BLOCKED layout: lane l holds elements e = l*R + r, r < R.
// Compare-exchange partners at distance < R live in the same lane (pure register work); only distances
// >= R cross lanes (one 64-bit shuffle with lane distance stride/R).  For R = 8 that is 21 shuffling stages
// out of 45 instead of 39 with the striped layout.
template <int R>
__device__ __forceinline__ void bitonic_sort_wave(unsigned long long (&v)[R], int lane) {
  constexpr int TOTAL = R * kWave;
#pragma unroll
  for (int size = 2; size <= TOTAL; size <<= 1) {
#pragma unroll
    for (int stride = size >> 1; stride >= 1; stride >>= 1) {
      if (stride < R) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int pr = r ^ stride;
          if (pr > r) {
            const int e = lane * R + r;
            const bool up = (e & size) == 0;  // ascending block
            const unsigned long long a = v[r], b = v[pr];
            const bool swap = up ? (a > b) : (a < b);
            if (swap) { v[r] = b; v[pr] = a; }
          }
        }
      } else {
        const int lstride = stride / R;            // lane distance of the partner
        const bool lower = (lane & lstride) == 0;  // this lane holds the lower-indexed element
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int e = lane * R + r;
          const unsigned long long other = shfl_xor_u64(v[r], lstride);
          const bool up = (e & size) == 0;
          // keep the smaller of the pair iff (ascending block) == (lower position): one compare, one select
          const bool keep = (v[r] < other) == (up == lower);
          v[r] = keep ? v[r] : other;
        }
      }
    }
  }
}

template <int R>
__device__ __forceinline__ void sort_and_emit(const KnnParams& P, const uint32_t* __restrict__ selkey,
                                              const uint32_t* __restrict__ selidx, int lane, int K,
                                              int64_t out_base, int i_global) {
  unsigned long long v[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int e = lane * R + r;
    v[r] = (e < K) ? ((static_cast<unsigned long long>(selkey[e]) << 32) | selidx[e]) : ~0ull;
  }
  bitonic_sort_wave<R>(v, lane);
  const int d = P.dilation;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int e = lane * R + r;
    if (e < K && (e % d) == 0) {
      const int pos = e / d;
      if (pos < P.Kout) {
        P.nn_out[out_base + pos] = static_cast<int64_t>(static_cast<uint32_t>(v[r]));
        if (P.ctr_out) P.ctr_out[out_base + pos] = i_global;
      }
    }
  }
}

// Exact K-th smallest (0-based K-1) of the R*64 keys held as k[r] per lane: largest tau with
// count(key < tau) < K.  32 steps of R compare+add and one DPP wave reduction.
// LOW > 0 resolves only bits 31..LOW and fills the rest with ones: an upper bound of the K-th key that is
// good enough for a pre-filter threshold (never for the final selection).
template <int R, int LOW = 0>
__device__ __forceinline__ uint32_t kth_smallest(const uint32_t (&k)[R], int K) {
  uint32_t tau = 0;
#pragma unroll 1
  for (int bit = 31; bit >= LOW; --bit) {
    const uint32_t cand = tau | (1u << bit);
    int c4[4] = {0, 0, 0, 0};  // independent chains: a single add-carry chain serialises on its latency
#pragma unroll
    for (int s = 0; s < R; ++s) c4[s & 3] += (k[s] < cand) ? 1 : 0;
    if (wave_sum((c4[0] + c4[1]) + (c4[2] + c4[3])) < K) tau = cand;
  }
  return tau | ((1u << LOW) - 1u);
}

template <int R>
__device__ __forceinline__ int count_below(const uint32_t (&k)[R], uint32_t tau, bool inclusive) {
  int c = 0;
#pragma unroll
  for (int s = 0; s < R; ++s) c += (inclusive ? (k[s] <= tau) : (k[s] < tau)) ? 1 : 0;
  return wave_sum(c);
}

// Ordered compaction of the winners {key < tau} + the first need_eq of {key == tau} out of R*64 keys
// (element e = r*64 + lane, increasing e = increasing point index) into skey/sidx.
template <int R>
__device__ __forceinline__ void compact_winners(const uint32_t (&k)[R], const uint32_t (&id)[R], int slots,
                                                uint32_t tau, int need_eq, uint32_t* skey, uint32_t* sidx,
                                                int lane) {
  int n_sel = 0, n_eq = 0;
  const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
  for (int s = 0; s < R; ++s) {
    if (s < slots) {  // uniform
      const bool lt = k[s] < tau;
      const bool eq = k[s] == tau;
      const unsigned long long m_eq = __ballot(eq);
      const int eq_rank = n_eq + __popcll(m_eq & below);
      const bool take = lt || (eq && eq_rank < need_eq);
      const unsigned long long m_take = __ballot(take);
      if (take) {
        const int pos = n_sel + __popcll(m_take & below);
        skey[pos] = k[s];
        sidx[pos] = id[s];
      }
      n_sel += __popcll(m_take);
      n_eq += __popcll(m_eq);
    }
  }
}

// Candidate stage of the pre-filtered select: R candidates per lane from LDS -> exact threshold -> winners.
template <int R>
__device__ __forceinline__ void select_from_candidates(const uint32_t* ckey, const uint32_t* cidx, int cnt, int K,
                                                       uint32_t* skey, uint32_t* sidx, int lane) {
  uint32_t ck[R], ci[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int e = r * kWave + lane;
    ck[r] = (e < cnt) ? ckey[e] : 0xFFFFFFFFu;
    ci[r] = (e < cnt) ? cidx[e] : 0u;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();   // candidates are in registers before sidx (aliasing cidx) is rewritten
  const uint32_t tau = kth_smallest<R>(ck, K);
  const int need_eq = K - count_below<R>(ck, tau, false);
  compact_winners<R>(ck, ci, (cnt + kWave - 1) / kWave, tau, need_eq, skey, sidx, lane);
}

// Phases 2-5 for one query row, executed by ONE wave holding SLOTS keys per lane (N <= 64 SLOTS).  Serves clouds of up to
// 2048 points (SLOTS = 16 / 32: register-resident without spills); larger rows take select_rows_coop below.
template <int SLOTS>
__device__ __forceinline__ void select_row(const KnnParams& P, float* drow, uint32_t* skey, int b, int i) {
  const int lane = lane_id();
  const int N = P.N, K = P.K;
  uint32_t* sidx = reinterpret_cast<uint32_t*>(drow);  // in-place compaction target (position <= index)

  // keys of this lane: element j = s*64 + lane
  uint32_t key[SLOTS];
  const int slots = (N + kWave - 1) / kWave;  // <= SLOTS
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int j = s * kWave + lane;
    key[s] = (s < slots && j < N && !(P.exclude_self && j == i)) ? key_of(drow[j]) : 0xFFFFFFFFu;
  }

  // phases 2-3.  Fast path: threshold a 256-key SAMPLE (4 keys per lane) at a rank chosen so that, with
  // overwhelming probability, at least K and at most ~2K+100 of the row's keys lie below it; compact those
  // candidates (index order) and run the exact 32-step bisection on <= 16 keys per lane instead of 64.
  // Any miss (fewer than K candidates, or too many) falls through to the exact full-row path: the
  // result never depends on the sample.
  bool done = false;
  if (P.sample_rank > 0 && slots >= 4) {
    const int st = slots / 4;
    uint32_t ks[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint32_t v = 0xFFFFFFFFu;
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) v = (s == i * st) ? key[s] : v;   // key[] stays in registers
      ks[i] = v;
    }
    const uint32_t ts = kth_smallest<4>(ks, P.sample_rank);
    const int cnt = count_below<SLOTS>(key, ts, true);
    const int cap = min(16 * kWave, (N / 2) & ~3);
    if (cnt >= K && cnt <= cap) {
      uint32_t* cidx = reinterpret_cast<uint32_t*>(drow);          // the row is dead: its keys are in registers
      uint32_t* ckey = cidx + cap;
      int n_c = 0;
      const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) {
        if (s < slots) {
          const bool in = key[s] <= ts;
          const unsigned long long m = __ballot(in);
          if (in) {
            const int pos = n_c + __popcll(m & below);
            ckey[pos] = key[s];
            cidx[pos] = static_cast<uint32_t>(s * kWave + lane);
          }
          n_c += __popcll(m);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (cnt <= 2 * kWave) select_from_candidates<2>(ckey, cidx, cnt, K, skey, sidx, lane);
      else if (cnt <= 4 * kWave) select_from_candidates<4>(ckey, cidx, cnt, K, skey, sidx, lane);
      else if (cnt <= 8 * kWave) select_from_candidates<8>(ckey, cidx, cnt, K, skey, sidx, lane);
      else select_from_candidates<16>(ckey, cidx, cnt, K, skey, sidx, lane);
      done = true;
    }
  }
  if (!done) {
    // exact full-row path
    const uint32_t tau = kth_smallest<SLOTS>(key, K);
    const int need_eq = K - count_below<SLOTS>(key, tau, false);  // >= 1 ties at the threshold
    uint32_t id[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) id[s] = static_cast<uint32_t>(s * kWave + lane);
    compact_winners<SLOTS>(key, id, slots, tau, need_eq, skey, sidx, lane);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  // phases 4-5
  const int64_t out_base = (static_cast<int64_t>(b) * N + i) * P.Kout;
  if (K <= 64) sort_and_emit<1>(P, skey, sidx, lane, K, out_base, i);
  else if (K <= 128) sort_and_emit<2>(P, skey, sidx, lane, K, out_base, i);
  else if (K <= 256) sort_and_emit<4>(P, skey, sidx, lane, K, out_base, i);
  else if (K <= 512) sort_and_emit<8>(P, skey, sidx, lane, K, out_base, i);
  else sort_and_emit<16>(P, skey, sidx, lane, K, out_base, i);     // deeper stacks than ResGCN-28 (k * d up to 1024)
}

// ---- cooperative select: all eight waves of the workgroup work on ONE row (N up to 4096: 8 keys per thread) ---------
// The wave-per-row form needs 64 keys per lane at N = 4096 (256 VGPRs + spills); here a thread holds element
// j = s * 512 + tid, s < 8, and the 32-step bisection counts across the workgroup: per step one DPP wave sum, one LDS
// atomic per wave and one barrier (three rotating counters: a slot is cleared two steps before it is used again).
// Winners are compacted in ANY order (the sort that follows orders them by (key, id)); among keys equal to the
// threshold the lowest point ids are taken, found by a 12-bit bisection on the id only when the tie is real.
constexpr int kCoopSlots = 8;

struct CoopCtx {
  int* cnt3;      // LDS int[4]: three rotating counters + the append cursor
  int phase;
};

__device__ __forceinline__ int coop_total(CoopCtx& c, int lane_count) {
  const int w = wave_sum(lane_count);
  if (lane_id() == 0 && w) atomicAdd(&c.cnt3[c.phase], w);
  __syncthreads();
  const int total = c.cnt3[c.phase];
  if (threadIdx.x == 0) c.cnt3[(c.phase + 2) % 3] = 0;
  c.phase = (c.phase + 1) % 3;
  return total;
}

__device__ __forceinline__ void select_row_coop(const KnnParams& P, float* drow, uint32_t* skey, int i, CoopCtx& c) {
  const int N = P.N, K = P.K;
  const int tid = threadIdx.x, lane = lane_id();
  uint32_t key[kCoopSlots];
#pragma unroll
  for (int s = 0; s < kCoopSlots; ++s) {
    const int j = s * kKnnThreads + tid;
    key[s] = (j < N && !(P.exclude_self && j == i)) ? key_of(drow[j]) : 0xFFFFFFFFu;
  }
  uint32_t tau = 0;
#pragma unroll 1
  for (int bit = 31; bit >= 0; --bit) {
    const uint32_t cand = tau | (1u << bit);
    int n = 0;
#pragma unroll
    for (int s = 0; s < kCoopSlots; ++s) n += (key[s] < cand) ? 1 : 0;
    if (coop_total(c, n) < K) tau = cand;
  }
  int n_lt = 0, n_eq = 0;
#pragma unroll
  for (int s = 0; s < kCoopSlots; ++s) { n_lt += (key[s] < tau) ? 1 : 0; n_eq += (key[s] == tau) ? 1 : 0; }
  const int need_eq = K - coop_total(c, n_lt);
  const int have_eq = coop_total(c, n_eq);
  uint32_t id_thr = 0xFFFFFFFFu;          // ids <= id_thr among the keys equal to tau are taken
  if (have_eq != need_eq) {               // block-uniform
    id_thr = 0;
#pragma unroll 1
    for (int bit = 12; bit >= 0; --bit) {
      const uint32_t cand = id_thr | (1u << bit);
      int n = 0;
#pragma unroll
      for (int s = 0; s < kCoopSlots; ++s) n += (key[s] == tau && static_cast<uint32_t>(s * kKnnThreads + tid) < cand) ? 1 : 0;
      if (coop_total(c, n) < need_eq) id_thr = cand;
    }
  }
  // unordered compaction: the row's distances are dead (every key is in a register and a barrier has passed), so the
  // ids go to the front of the row itself
  uint32_t* sidx = reinterpret_cast<uint32_t*>(drow);
  const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
  for (int s = 0; s < kCoopSlots; ++s) {
    const uint32_t j = static_cast<uint32_t>(s * kKnnThreads + tid);
    const bool take = key[s] < tau || (key[s] == tau && j <= id_thr);
    const unsigned long long m = __ballot(take);
    int base = 0;
    if (lane == 0 && m) base = atomicAdd(&c.cnt3[3], __popcll(m));
    base = __builtin_amdgcn_readfirstlane(base);
    if (take) {
      const int pos = base + __popcll(m & below);
      skey[pos] = key[s];
      sidx[pos] = j;
    }
  }
  __syncthreads();
  if (tid == 0) c.cnt3[3] = 0;            // cursor for the next row (its first append is many barriers away)
}

template <int R>
__device__ __forceinline__ void emit_sorted(const KnnParams& P, const float* drow, const uint32_t* skey, int b, int i) {
  const int64_t out_base = (static_cast<int64_t>(b) * P.N + i) * P.Kout;
  sort_and_emit<R>(P, skey, reinterpret_cast<const uint32_t*>(drow), lane_id(), P.K, out_base, i);
}

// LDS layout (dynamic): q[C][TM] | sq[TM] | dist[TM][Npad] | selkey[TM][Kpad]
// (the winners' indices are compacted IN PLACE at the front of each dist row: position <= index)
// SLOTS: keys per lane of the wave-per-row select (16: N <= 1024, 32: N <= 2048); 0: the cooperative select (N <= 4096)
template <int TM, bool VEC4, int SLOTS>
__global__ __launch_bounds__(kKnnThreads) void knn_dense_kernel(const KnnParams P, int Npad, int Kpad) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int C = P.C, N = P.N;
  float* q = reinterpret_cast<float*>(smem);                 // [C][TM]
  float* sq = q + static_cast<size_t>(C) * TM;               // [TM] (padded to 16 floats)
  float* dist = sq + 16;                                     // [TM][Npad]
  uint32_t* selkey = reinterpret_cast<uint32_t*>(dist + static_cast<size_t>(TM) * Npad);  // [TM][Kpad]

  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const int tiles_per_b = (N + TM - 1) / TM;
  // redo pass (TM == 1): the tiles are the rows the filter kernel listed, a fixed grid strides over the list
  const int n_tiles = P.redo ? P.redo[0] : static_cast<int>(gridDim.x);
  for (int tile_id = blockIdx.x; tile_id < n_tiles; tile_id += gridDim.x) {
  int b, i0;
  if (P.redo) {
    const int row = P.redo[1 + tile_id];
    b = row / N;
    i0 = row % N;
  } else {
    b = tile_id / tiles_per_b;
    i0 = (tile_id % tiles_per_b) * TM;
  }
  const float* xb = P.x + static_cast<int64_t>(b) * P.sb;
  __syncthreads();   // the previous tile's LDS contents are dead

  // ---- phase 0: stage the TM query points, channel-major [c][r] ----
  for (int e = tid; e < C * TM; e += kKnnThreads) {
    const int c = e / TM, r = e % TM;
    const int i = min(i0 + r, N - 1);
    q[e] = xb[c * P.sc + i * P.sn];
  }
  __syncthreads();
  if (tid < TM) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s = fmaf(q[c * TM + tid], q[c * TM + tid], s);
    sq[tid] = s;
  }
  __syncthreads();

  // ---- phase 1: distance strip ----
  constexpr int JJ = 4;
  const int n_phase1 = N;
  if (VEC4) {
    // contiguous, 16B-aligned points: each thread owns 4 CONSECUTIVE columns -> one dwordx4 load per
    // channel (1 KiB per wave instruction) and one ds_write_b128 per row.
    for (int j0 = 0; j0 < n_phase1; j0 += kKnnThreads * JJ) {
      const int cbase = j0 + tid * JJ;          // N % 4 == 0 in this mode: a block is all-in or all-out
      const bool in = cbase < N;
      const float* xc = xb + (in ? cbase : 0);
      float acc[JJ][TM], sj[JJ];
#pragma unroll
      for (int jj = 0; jj < JJ; ++jj) {
        sj[jj] = 0.f;
#pragma unroll
        for (int r = 0; r < TM; ++r) acc[jj][r] = 0.f;
      }
      // software pipeline: the next CH channels are in flight while the current CH are consumed
      // (only 2 waves per SIMD are resident, so latency must be hidden by ILP, not by occupancy)
      constexpr int CH = TM >= 8 ? 4 : 8;      // eight rows: 32 accumulators, so half the loads in flight (no spills)
      float4 nxt[CH];
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        nxt[u] = (u < C) ? *reinterpret_cast<const float4*>(xc + static_cast<int64_t>(u) * P.sc)
                         : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      for (int c0 = 0; c0 < C; c0 += CH) {
        float4 cur[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) cur[u] = nxt[u];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          const int c = c0 + CH + u;
          nxt[u] = (c < C) ? *reinterpret_cast<const float4*>(xc + static_cast<int64_t>(c) * P.sc)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          const int c = min(c0 + u, C - 1);      // channels past C carry zeros: no contribution
          const float xv[JJ] = {cur[u].x, cur[u].y, cur[u].z, cur[u].w};
          float qv[TM];
#pragma unroll
          for (int r = 0; r < TM; ++r) qv[r] = q[c * TM + r];
#pragma unroll
          for (int jj = 0; jj < JJ; ++jj) {
            sj[jj] = fmaf(xv[jj], xv[jj], sj[jj]);
#pragma unroll
            for (int r = 0; r < TM; ++r) acc[jj][r] = fmaf(qv[r], xv[jj], acc[jj][r]);
          }
        }
      }
      if (in) {
#pragma unroll
        for (int r = 0; r < TM; ++r) {
          float4 o;
          o.x = (sq[r] + (-2.f * acc[0][r])) + sj[0];
          o.y = (sq[r] + (-2.f * acc[1][r])) + sj[1];
          o.z = (sq[r] + (-2.f * acc[2][r])) + sj[2];
          o.w = (sq[r] + (-2.f * acc[3][r])) + sj[3];
          *reinterpret_cast<float4*>(dist + r * Npad + cbase) = o;
        }
      }
    }
  } else {
    for (int j0 = 0; j0 < n_phase1; j0 += kKnnThreads * JJ) {
      float acc[JJ][TM], sj[JJ];
      int col[JJ];
#pragma unroll
      for (int jj = 0; jj < JJ; ++jj) {
        col[jj] = j0 + jj * kKnnThreads + tid;
        sj[jj] = 0.f;
#pragma unroll
        for (int r = 0; r < TM; ++r) acc[jj][r] = 0.f;
      }
#pragma unroll 2
      for (int c = 0; c < C; ++c) {
        float xv[JJ];
#pragma unroll
        for (int jj = 0; jj < JJ; ++jj) {
          const int j = min(col[jj], N - 1);
          xv[jj] = xb[c * P.sc + j * P.sn];
        }
        float qv[TM];
#pragma unroll
        for (int r = 0; r < TM; ++r) qv[r] = q[c * TM + r];
#pragma unroll
        for (int jj = 0; jj < JJ; ++jj) {
          sj[jj] = fmaf(xv[jj], xv[jj], sj[jj]);
#pragma unroll
          for (int r = 0; r < TM; ++r) acc[jj][r] = fmaf(qv[r], xv[jj], acc[jj][r]);
        }
      }
#pragma unroll
      for (int jj = 0; jj < JJ; ++jj) {
        if (col[jj] < N) {
#pragma unroll
          for (int r = 0; r < TM; ++r) {
            // reference association: (x_square + x_inner) + x_square^T, x_inner = -2 * <x_i, x_j>
            dist[r * Npad + col[jj]] = (sq[r] + (-2.f * acc[jj][r])) + sj[jj];
          }
        }
      }
    }
  }
  __syncthreads();

  if constexpr (SLOTS > 0) {
    // ---- phases 2-5: one wave per query row ----
    for (int r = wave; r < TM; r += kKnnWaves) {
      const int i = i0 + r;
      if (i >= N) continue;  // wave-uniform
      select_row<SLOTS>(P, dist + static_cast<size_t>(r) * Npad, selkey + static_cast<size_t>(r) * Kpad, b, i);
    }
  } else {
    // ---- phases 2-3 with the whole workgroup per row, then phases 4-5 one wave per row ----
    __shared__ int coop_cnt[4];
    if (tid < 4) coop_cnt[tid] = 0;
    __syncthreads();
    CoopCtx cc{coop_cnt, 0};
    for (int r = 0; r < TM; ++r) {
      if (i0 + r >= N) break;  // block-uniform
      select_row_coop(P, dist + static_cast<size_t>(r) * Npad, selkey + static_cast<size_t>(r) * Kpad, i0 + r, cc);
    }
    __syncthreads();
    for (int r = wave; r < TM; r += kKnnWaves) {
      const int i = i0 + r;
      if (i >= N) continue;
      const float* dr = dist + static_cast<size_t>(r) * Npad;
      const uint32_t* sk = selkey + static_cast<size_t>(r) * Kpad;
      const int K = P.K;
      if (K <= 64) emit_sorted<1>(P, dr, sk, b, i);
      else if (K <= 128) emit_sorted<2>(P, dr, sk, b, i);
      else if (K <= 256) emit_sorted<4>(P, dr, sk, b, i);
      else if (K <= 512) emit_sorted<8>(P, dr, sk, b, i);
      else emit_sorted<16>(P, dr, sk, b, i);
    }
  }
  }  // tiles
}


// =======================================================================================
// Candidate-filter kNN (N >= 1024, contiguous points): 16 query rows per workgroup.
//
// The exact kernel above is bound by L2->L1 traffic: every 8-row workgroup streams the sample's whole
// feature block, and the 8 x N distance strip fills the LDS.  Here a per-row threshold tau_r is first
// estimated from the distances to 256 SAMPLED candidates (rank chosen so that, with overwhelming
// probability, between K and ~1000 of the N candidates fall below it); the full distance pass then keeps
// only candidates with key <= tau_r, appended to a 1024-entry per-row list.  16 rows fit (128 KB), so the
// feature block is streamed half as often, and the select runs on <= 16 candidates per lane without
// touching the full row again.  Exactness does not depend on the sample: a row whose list holds fewer
// than K or more than 1024 candidates is flagged and redone by the exact kernel (second launch, which
// exits immediately for unflagged tiles).  Distances use the same fma chain and association as above.
// =======================================================================================
constexpr int kRedoGrid = 1024;   // workgroups of the list-driven exact pass (4 per CU)
constexpr int kFTM = 16;
constexpr int kFSamples = 256;
constexpr int kFThreads = 1024;   // 16 waves: one per query row in the select phase, 4 per SIMD to hide latency
constexpr int kFWaves = kFThreads / kWave;
#ifndef DGCN_KNN_FB_WAVES
#define DGCN_KNN_FB_WAVES 8
#endif
constexpr int kFbWaves = DGCN_KNN_FB_WAVES;   // waves per workgroup of the bf16 filter kernel: 8 = two workgroups per CU whose phases
                                               // (matrix pass / per-row select) overlap; 16 = round 4's single resident workgroup

constexpr int kPrepThreads = kFSamples;   // one thread per sampled candidate

// Pre-pass of the filter path, 16 query rows per (small, high-occupancy) workgroup:
//   sqnorm[b,i] = |x_i|^2 as the channel-ordered fma chain used everywhere else in this file;
//   tau[b,i]    = the sample_rank-th smallest key among the distances to 256 sampled candidates (4 windows of 64
//                 consecutive points, rotated per tile) -- the per-row threshold of the candidate filter;
//   redo[0]     = 0 (the filter kernel appends the rows it cannot finish).
// Kept out of the filter kernel because it is pure latency (64 dependent-ish loads per thread) and that kernel
// runs one 1024-thread workgroup per CU: there it cost 65-85 us per call, here ~15.
__global__ __launch_bounds__(kPrepThreads) void knn_prep_kernel(const KnnParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int TM = kFTM;
  const int C = P.C, N = P.N;
  float* q = reinterpret_cast<float*>(smem);                        // [C][16]
  float* sq = q + static_cast<size_t>(C) * TM;                      // [16]
  uint32_t* skeys = reinterpret_cast<uint32_t*>(sq + TM);           // [16][256]
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = tid >> 6;
  const int tiles_per_b = (N + TM - 1) / TM;
  const int b = blockIdx.x / tiles_per_b;
  const int tile = blockIdx.x % tiles_per_b;
  const int i0 = tile * TM;
  const float* xb = P.x + static_cast<int64_t>(b) * P.sb;
  if (blockIdx.x == 0 && tid == 0) P.redo[0] = 0;

  for (int e = tid; e < C * TM; e += kPrepThreads) {
    const int c = e / TM, r = e % TM;
    q[e] = xb[static_cast<int64_t>(c) * P.sc + min(i0 + r, N - 1)];
  }
  __syncthreads();
  if (tid < TM) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s = fmaf(q[c * TM + tid], q[c * TM + tid], s);
    sq[tid] = s;
    if (i0 + tid < N) P.sqnorm[static_cast<int64_t>(b) * N + i0 + tid] = s;
  }
  __syncthreads();
  {
    const int s = tid;
    const int quarter = N / 4;
    const int j = ((s / 64) * quarter + (tile * 64) % quarter + (s % 64)) % N;
    float acc[TM], sj = 0.f;
#pragma unroll
    for (int r = 0; r < TM; ++r) acc[r] = 0.f;
    constexpr int SCH = 16;  // channel loads in flight per thread
    for (int c0 = 0; c0 < C; c0 += SCH) {
      float xv[SCH];
#pragma unroll
      for (int u = 0; u < SCH; ++u) xv[u] = (c0 + u < C) ? xb[static_cast<int64_t>(c0 + u) * P.sc + j] : 0.f;
#pragma unroll
      for (int u = 0; u < SCH; ++u) {
        const int c = min(c0 + u, C - 1);      // channels past C carry zeros
        sj = fmaf(xv[u], xv[u], sj);
        const float4* qc = reinterpret_cast<const float4*>(q + c * TM);   // wave-wide broadcast reads
#pragma unroll
        for (int r4 = 0; r4 < TM / 4; ++r4) {
          const float4 qq = qc[r4];
          acc[4 * r4 + 0] = fmaf(qq.x, xv[u], acc[4 * r4 + 0]);
          acc[4 * r4 + 1] = fmaf(qq.y, xv[u], acc[4 * r4 + 1]);
          acc[4 * r4 + 2] = fmaf(qq.z, xv[u], acc[4 * r4 + 2]);
          acc[4 * r4 + 3] = fmaf(qq.w, xv[u], acc[4 * r4 + 3]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < TM; ++r) skeys[r * kFSamples + s] = key_of((sq[r] + (-2.f * acc[r])) + sj);
  }
  __syncthreads();
  for (int rr = wave; rr < TM; rr += kPrepThreads / kWave) {
    uint32_t ks[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) ks[u] = skeys[rr * kFSamples + u * kWave + lane];
    const uint32_t t = kth_smallest<4, 12>(ks, P.sample_rank);   // 12 low mantissa bits of a threshold do not matter
    if (lane == 0 && i0 + rr < N) P.tau[static_cast<int64_t>(b) * N + i0 + rr] = t;
  }
}

// ---- wave-wide helpers of the bucket select ---------------------------------------------------------------------------
// inclusive prefix sum over the 64 lanes (the DPP network of wave_sum, every lane keeps its partial)
__device__ __forceinline__ int wave_scan_incl(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);
  return v;
}

// wave-wide minimum, returned wave-uniform (lanes a DPP step does not write keep the identity)
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  constexpr int kId = static_cast<int>(0xFFFFFFFFu);
  v = min(v, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(kId, static_cast<int>(v), 0x111, 0xf, 0xf, false)));
  v = min(v, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(kId, static_cast<int>(v), 0x112, 0xf, 0xf, false)));
  v = min(v, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(kId, static_cast<int>(v), 0x114, 0xf, 0xf, false)));
  v = min(v, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(kId, static_cast<int>(v), 0x118, 0xf, 0xf, false)));
  v = min(v, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(kId, static_cast<int>(v), 0x142, 0xa, 0xf, false)));
  v = min(v, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(kId, static_cast<int>(v), 0x143, 0xc, 0xf, false)));
  return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 63));
}

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- bucket select: the Kout order statistics of a candidate list, no threshold search, no sort ----------------------
// A dilated graph emits the neighbours of rank 0, d, 2d, ... only (gcn_lib/dense/torch_edge.py:26-28): k = K / d order
// statistics of the candidate list, not its K sorted winners.  The 32-step bisection for the K-th key, the compaction of
// the winners and the bitonic sort of up to 512 (key, id) pairs (~3,000 of the kernel's ~5,900 vector instructions per
// wave at K = 432, profiles/r03_knn_filter_bf16_counters.md) are replaced by one counting pass:
//   1. the keys are mapped monotonically onto NB = CAP buckets between the second-smallest key and the largest (the
//      smallest is the query point itself at distance ~0, an outlier of the key range; it gets bucket 0);
//   2. an LDS histogram and its prefix sum give every bucket its rank range [p_b, p_b + n_b);
//   3. lane j < Kout finds the bucket that holds rank j d by binary search in the prefix array; the few candidates of
//      those <= Kout buckets are copied into per-bucket lists;
//   4. the lane picks the (j d - p_b)-th smallest (key, id) of its bucket's list (n_b is 1-3 on real data) and writes
//      output position j.
// Ranks are ranks in (key, id) order -- ties by lowest point id, as in the exact path -- and the list holds every key
// <= tau_r, so the element of rank r < K of the list is the element of rank r of the row.  Anything unusual (more than
// 32 outputs per row, a bucket holding more than 16 candidates -- duplicate points --, more than TCAP list entries)
// returns false and the caller runs the bisection + sort path on the same registers.
// scratch: the row's own list storage, CAP words at ``sa`` (prefix array) and CAP words at ``sb`` (byte map bucket ->
// slot, slot tables, lists); the candidates are in registers by now.
template <int R, int CAP>
__device__ __forceinline__ bool bucket_select(const KnnParams& P, const uint32_t (&ck)[R], const uint32_t (&ci)[R], int cnt,
                                              uint32_t* sa, uint32_t* sb, int b, int i, int lane) {
  constexpr int NB = CAP;                     // buckets
  constexpr int W = NB / kWave;               // prefix entries per lane in the scan
  constexpr int TCAP = CAP / 4;               // list entries over all target buckets
  constexpr int kMaxPerBucket = 16;
  constexpr int kMaxOut = 32;
  const int Kout = P.Kout, d = P.dilation;
  if (Kout > kMaxOut) return false;           // wave-uniform
  uint32_t* pref = sa;                                              // [NB]
  unsigned char* slot_of = reinterpret_cast<unsigned char*>(sb);    // [NB] bytes
  uint32_t* soff = sb + NB / 4;                                     // [32]
  uint32_t* scnt = soff + kMaxOut;                                  // [32]
  unsigned long long* list = reinterpret_cast<unsigned long long*>(scnt + kMaxOut);   // [TCAP] (key << 32 | id)

  // 1. key range: smallest, second-smallest distinct, largest
  uint32_t mn = 0xFFFFFFFFu, mx = 0u;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const bool valid = r * kWave + lane < cnt;
    mn = min(mn, ck[r]);                      // (padding keys are 0xFFFFFFFF)
    mx = max(mx, valid ? ck[r] : 0u);
  }
  const uint32_t kmin = wave_min_u32(mn);
  const uint32_t hi = ~wave_min_u32(~mx);
  uint32_t m2 = 0xFFFFFFFFu;
#pragma unroll
  for (int r = 0; r < R; ++r) m2 = min(m2, ck[r] > kmin ? ck[r] : 0xFFFFFFFFu);
  uint32_t lo = wave_min_u32(m2);
  if (lo > hi) lo = kmin;                     // every valid key equals kmin
  const uint32_t span = hi - lo;
  // bucket(key) = key < lo ? 0 : 1 + floor((key - lo) * inv / 2^32) <= NB - 1, monotone in key
  float invf = static_cast<float>(NB - 2) * 4294967296.f / (static_cast<float>(span) + 1.f) * (1.f - 1.f / 4194304.f);
  invf = fminf(invf, 4294967040.f);
  const uint32_t inv = static_cast<uint32_t>(invf);

  // 2. histogram
  {
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int q = 0; q < W; q += 4) *reinterpret_cast<uint4*>(pref + lane * W + q) = z;
    if (NB / 4 >= 4 * kWave) *reinterpret_cast<uint4*>(sb + lane * 4) = make_uint4(~0u, ~0u, ~0u, ~0u);
    else *reinterpret_cast<uint2*>(sb + lane * 2) = make_uint2(~0u, ~0u);
  }
  wave_lds_sync();
  uint32_t bk[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const bool valid = r * kWave + lane < cnt;
    const uint32_t off = ck[r] - lo;
    bk[r] = ck[r] < lo ? 0u : min(1u + __umulhi(off, inv), static_cast<uint32_t>(NB - 1));
    if (valid) atomicAdd(&pref[bk[r]], 1u);
  }
  wave_lds_sync();
  // prefix sum: lane l owns buckets [l W, l W + W)
  {
    uint32_t c[W];
#pragma unroll
    for (int q = 0; q < W; q += 4) {
      const uint4 v = *reinterpret_cast<const uint4*>(pref + lane * W + q);
      c[q] = v.x; c[q + 1] = v.y; c[q + 2] = v.z; c[q + 3] = v.w;
    }
    uint32_t tot = 0;
#pragma unroll
    for (int q = 0; q < W; ++q) { const uint32_t t = c[q]; c[q] = tot; tot += t; }
    const uint32_t base = static_cast<uint32_t>(wave_scan_incl(static_cast<int>(tot))) - tot;
#pragma unroll
    for (int q = 0; q < W; q += 4)
      *reinterpret_cast<uint4*>(pref + lane * W + q) = make_uint4(base + c[q], base + c[q + 1], base + c[q + 2], base + c[q + 3]);
  }
  wave_lds_sync();

  // 3. lane j: the bucket of rank j d
  const bool active = lane < Kout;
  const uint32_t rank = static_cast<uint32_t>(active ? lane * d : 0);
  int bj = 0;
#pragma unroll
  for (int step = NB / 2; step >= 1; step >>= 1)
    if (pref[bj + step] <= rank) bj += step;
  const uint32_t pj = pref[bj];
  const uint32_t nxt = (bj + 1 < NB) ? pref[min(bj + 1, NB - 1)] : static_cast<uint32_t>(cnt);
  const uint32_t nj = nxt - pj;
  if (!active) bj = -1 - lane;                // distinct from every bucket and from each other
  const int prev = __shfl_up(bj, 1);
  const bool first = active && (lane == 0 || bj != prev);
  if (__ballot(active && nj > static_cast<uint32_t>(kMaxPerBucket))) return false;
  const unsigned long long mfirst = __ballot(first);
  const int slot = __popcll(mfirst & ((2ull << lane) - 1ull)) - 1;        // slot of my bucket (first lanes: their own)
  const int vcnt = first ? static_cast<int>(nj) : 0;
  const int incl = wave_scan_incl(vcnt);
  if (__builtin_amdgcn_readlane(incl, 63) > TCAP) return false;
  if (first) {
    slot_of[bj] = static_cast<unsigned char>(slot);
    soff[slot] = static_cast<uint32_t>(incl - vcnt);
    scnt[slot] = 0u;
  }
  wave_lds_sync();
  // the candidates of the target buckets -> their bucket's list
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const bool valid = r * kWave + lane < cnt;
    if (valid) {
      const uint32_t sl = slot_of[bk[r]];
      if (sl != 0xFFu) {
        const uint32_t pos = soff[sl] + atomicAdd(&scnt[sl], 1u);
        list[pos] = (static_cast<unsigned long long>(ck[r]) << 32) | ci[r];
      }
    }
  }
  wave_lds_sync();
  // 4. the (rank - p_b)-th smallest of the bucket's list
  if (active) {
    const uint32_t off = soff[slot];
    const uint32_t q = rank - pj;
    unsigned long long ans = 0ull;
    for (uint32_t u = 0; u < nj; ++u) {
      const unsigned long long xu = list[off + u];
      uint32_t below = 0;
      for (uint32_t v = 0; v < nj; ++v) below += list[off + v] < xu ? 1u : 0u;
      if (below == q) ans = xu;
    }
    const int64_t o = (static_cast<int64_t>(b) * P.N + i) * Kout + lane;
    P.nn_out[o] = static_cast<int64_t>(static_cast<uint32_t>(ans));
    if (P.ctr_out) P.ctr_out[o] = i;
  }
  return true;
}

// The select of one row on candidates held in registers (element e = r * 64 + lane, padding keys 0xFFFFFFFF);
// ckey / cidx: CAP words of LDS scratch each (the LDS-list kernels pass the row's own list storage).
template <int R, int CAP>
__device__ __forceinline__ void filter_select_regs(const KnnParams& P, const uint32_t (&ck)[R], const uint32_t (&ci)[R],
                                                   uint32_t* ckey, uint32_t* cidx, int cnt, int b, int i, int lane) {
  const int K = P.K;
#ifndef KNNF_NO_BUCKET_SELECT
  if (bucket_select<R, CAP>(P, ck, ci, cnt, ckey, cidx, b, i, lane)) return;
  wave_lds_sync();
#endif
  const uint32_t tau = kth_smallest<R>(ck, K);
  const int n_lt = count_below<R>(ck, tau, false);
  const int need_eq = K - n_lt;
  // candidates arrive in arbitrary order: among keys == tau take the need_eq LOWEST point ids (12-bit bisection
  // on the id, ids are < 4096) -- only when the tie at tau is real, i.e. more keys equal tau than are needed
  uint32_t tid_thr = 0xFFFFFFFFu;
  if (count_below<R>(ck, tau, true) - n_lt != need_eq) {
    tid_thr = 0;
#pragma unroll 1
    for (int bit = 11; bit >= 0; --bit) {
      const uint32_t cand = tid_thr | (1u << bit);
      int c = 0;
#pragma unroll
      for (int r = 0; r < R; ++r) c += (ck[r] == tau && ci[r] < cand) ? 1 : 0;
      if (wave_sum(c) < need_eq) tid_thr = cand;
    }
  }
  int n_sel = 0;
  const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const bool take = (ck[r] < tau) || (ck[r] == tau && ci[r] <= tid_thr);
    const unsigned long long m = __ballot(take);
    if (take) {
      const int pos = n_sel + __popcll(m & below);
      ckey[pos] = ck[r];
      cidx[pos] = ci[r];
    }
    n_sel += __popcll(m);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int64_t out_base = (static_cast<int64_t>(b) * P.N + i) * P.Kout;
  if (K <= 64) sort_and_emit<1>(P, ckey, cidx, lane, K, out_base, i);
  else if (K <= 128) sort_and_emit<2>(P, ckey, cidx, lane, K, out_base, i);
  else if (K <= 256) sort_and_emit<4>(P, ckey, cidx, lane, K, out_base, i);
  else sort_and_emit<8>(P, ckey, cidx, lane, K, out_base, i);
}

template <int R, int CAP>
__device__ __forceinline__ void filter_select_row(const KnnParams& P, uint32_t* ckey, uint32_t* cidx, int cnt,
                                                  int b, int i, int lane) {
  uint32_t ck[R], ci[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int e = r * kWave + lane;
    ck[r] = (e < cnt) ? ckey[e] : 0xFFFFFFFFu;
    ci[r] = (e < cnt) ? cidx[e] : 0xFFFFFFFFu;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();  // candidates live in registers before their LDS slots are reused
  filter_select_regs<R, CAP>(P, ck, ci, ckey, cidx, cnt, b, i, lane);
}

// the same for a candidate list in global memory (knn_filter2_kernel): (key, id) pairs, LDS scratch sa / sb
template <int R, int CAP>
__device__ __forceinline__ void filter_select_list(const KnnParams& P, const uint2* __restrict__ list, uint32_t* sa,
                                                   uint32_t* sb, int cnt, int b, int i, int lane) {
  uint32_t ck[R], ci[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int e = r * kWave + lane;
    uint2 v = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
    if (e < cnt) v = list[e];
    ck[r] = v.x;
    ci[r] = v.y;
  }
  filter_select_regs<R, CAP>(P, ck, ci, sa, sb, cnt, b, i, lane);
}

// Profiling-only compile-time switches (never defined in the shipped build; results are WRONG with them):
//   KNNF_STOP_AFTER=1|2  return after the threshold load / after the distance + append pass
//   KNNF_NO_MFMA, KNNF_NO_APPEND  drop the matrix-core work / the candidate appends
// They produced the phase decomposition quoted in DESIGN.md section 4.3 (rocprofv3 per-kernel times of variant builds).
// CAP = per-row candidate list capacity: 512 (64 KB of lists -> two workgroups per CU, their phases overlap)
// when K leaves enough room below it, else 1024.
template <int kFCap, int KS>
__global__ __launch_bounds__(kFThreads, 4) void knn_filter_kernel(const KnnParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int TM = kFTM;
  const int C = P.C, N = P.N, K = P.K;
  float* q = reinterpret_cast<float*>(smem);                        // [C][16]
  float* sq = q + static_cast<size_t>(C) * TM;                      // [16]
  uint32_t* tau = reinterpret_cast<uint32_t*>(sq + TM);             // [16]
  int* cnt = reinterpret_cast<int*>(tau + TM);                      // [16]
  uint32_t* ckey = reinterpret_cast<uint32_t*>(cnt + TM);           // [16][kFCap]
  uint32_t* cidx = ckey + TM * kFCap;                               // [16][kFCap]

  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = tid >> 6;
  const int tiles_per_b = (N + TM - 1) / TM;
  const int b = blockIdx.x / tiles_per_b;
  const int tile = blockIdx.x % tiles_per_b;
  const int i0 = tile * TM;
  const float* xb = P.x + static_cast<int64_t>(b) * P.sb;

  // ---- stage the 16 query points ----
  for (int e = tid; e < C * TM; e += kFThreads) {
    const int c = e / TM, r = e % TM;
    q[e] = xb[static_cast<int64_t>(c) * P.sc + min(i0 + r, N - 1)];
  }
  if (tid < TM) {   // row norms and sample thresholds come from knn_prep_kernel
    const int64_t row = static_cast<int64_t>(b) * N + min(i0 + tid, N - 1);
    sq[tid] = P.sqnorm[row];
    tau[tid] = P.tau[row];
    cnt[tid] = 0;
  }
  __syncthreads();

#if defined(KNNF_STOP_AFTER) && KNNF_STOP_AFTER == 1
  return;
#endif
  // ---- full distance pass on the matrix cores, keep only candidates with key <= tau_r ----
  // v_mfma_f32_16x16x4_f32: D(16 rows x 16 cols) += A(16 x 4 ch) * B(4 ch x 16 cols), an exact k-ordered f32
  // fma chain (bitwise the VALU fmaf chain of the exact kernel).  A = the 16 staged query rows, kept in
  // registers for a 64-channel chunk; B = candidate features straight from memory: lane l loads the float4
  // of channel 4*ks + (l>>4), columns col0 + 4*(l&15) .. +3, and component t feeds column tile t, i.e. MFMA
  // column (l&15) of tile t is point col0 + 4*(l&15) + t.  One wave owns 64 consecutive candidates.
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int li = lane & 15, lk = lane >> 4;
  const unsigned long long below = (1ull << lane) - 1ull;
  const float* sqn = P.sqnorm + static_cast<int64_t>(b) * N;
  // Channel chunks of 4*KS channels are DOUBLE BUFFERED (A/B): the loads of the next chunk -- or of the next
  // column block's first chunk -- are in flight while the current chunk's MFMAs (and the append code below) run.
  // Without this the 4 waves of a SIMD convoy: all wait for L2 together, then queue on the MFMA pipe together, and
  // load time, MFMA time and append time simply add up (measured 95 + 110 + 68 us per call).
  constexpr int CHUNK = 4 * KS;
  const int cpb = ((C + CHUNK - 1) / CHUNK + 1) & ~1;   // chunks per column block, rounded up to even (zero-padded)
  // Loads are UNCONDITIONAL (clamped addresses) so that they sit in straight-line code and the compiler can wait
  // with an exact vmcnt instead of vmcnt(0): a channel past C gets a zero A operand (the B value, real data from
  // the clamped address, is multiplied away), a column block past N is discarded by `in` in the append code.
  auto load_chunk = [&](int col0, int cc, float (&a)[KS], float4 (&bx)[KS]) {
    const int cb = min(col0 + 4 * li, N - 4);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int c = cc + 4 * ks + lk;
      const int cl = min(c, C - 1);
      const float qa = q[cl * TM + li];
      a[ks] = (c < C) ? qa : 0.f;
      bx[ks] = *reinterpret_cast<const float4*>(xb + static_cast<int64_t>(cl) * P.sc + cb);
    }
  };
  auto mfma_chunk = [&](f32x4 (&acc)[4], const float (&a)[KS], const float4 (&bx)[KS]) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#ifdef KNNF_NO_MFMA
      acc[0][0] += a[ks] * bx[ks].x; acc[1][0] += a[ks] * bx[ks].y; acc[2][0] += a[ks] * bx[ks].z; acc[3][0] += a[ks] * bx[ks].w;
#else
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], bx[ks].x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], bx[ks].y, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], bx[ks].z, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], bx[ks].w, acc[3], 0, 0, 0);
#endif
    }
  };
  constexpr int kColStride = kFWaves * 64;
  float aA[KS], aB[KS];
  float4 bA[KS], bB[KS];
  load_chunk(wave * 64, 0, aA, bA);
  for (int col0 = wave * 64; col0 < N; col0 += kColStride) {
    const int cbase = col0 + 4 * li;
    const bool in = cbase < N;  // N % 4 == 0
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ch = 0; ch < cpb; ch += 2) {
      // sched_barrier: keep "issue the next chunk's loads, THEN run this chunk's MFMAs" -- left alone, the
      // scheduler sinks every load next to its first use (lower register pressure, no overlap at all)
      load_chunk(col0, (ch + 1) * CHUNK, aB, bB);
      __builtin_amdgcn_sched_barrier(0);
      mfma_chunk(acc, aA, bA);
      __builtin_amdgcn_sched_barrier(0);
      // next A chunk: this block's, or the first of the next block (clamped past the end: loaded, never used)
      const bool more = ch + 2 < cpb;
      load_chunk(more ? col0 : col0 + kColStride, more ? (ch + 2) * CHUNK : 0, aA, bA);
      __builtin_amdgcn_sched_barrier(0);
      mfma_chunk(acc, aB, bB);
      __builtin_amdgcn_sched_barrier(0);
    }
#ifdef KNNF_NO_APPEND
    {
      float tsum = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) tsum += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
      if (tsum == 123.456f) cnt[0] = 1;
      continue;
    }
#endif
    // acc[t][reg] = <x_row, x_col> for row = lk*4 + reg, col = cbase + t
    float sj[4] = {0.f, 0.f, 0.f, 0.f};
    if (in) load_vec<4>(sj, sqn + cbase);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int r = lk * 4 + reg;  // the four 16-lane groups hold four different rows
      const uint32_t tr = tau[r];
      const int self = P.exclude_self ? i0 + r : -1;
      uint32_t key[4];
      bool hit[4];
      unsigned long long m[4];
      const unsigned long long grp = 0xFFFFull << (16 * lk);  // this row's lanes
      int tot = 0;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        key[t] = key_of((sq[r] + (-2.f * acc[t][reg])) + sj[t]);
        hit[t] = in && key[t] <= tr && (cbase + t) != self;
        m[t] = __ballot(hit[t]) & grp;
        tot += __popcll(m[t]);
      }
      // one LDS atomic per (row, wave, reg): the group leader reserves the slots of its row
      int base = 0;
      if (li == 0 && tot) base = atomicAdd(&cnt[r], tot);
      base = __shfl(base, lk * 16);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int pos = base + __popcll(m[t] & below);
        if (hit[t] && pos < kFCap) {
          ckey[r * kFCap + pos] = key[t];
          cidx[r * kFCap + pos] = static_cast<uint32_t>(cbase + t);
        }
        base += __popcll(m[t]);
      }
    }
  }
  __syncthreads();
#if defined(KNNF_STOP_AFTER) && KNNF_STOP_AFTER == 2
  return;
#endif

  // ---- per-row select on the candidate lists ----
  for (int rr = wave; rr < TM; rr += kFWaves) {
    const int i = i0 + rr;
    if (i >= N) continue;  // wave-uniform
    const int c = cnt[rr];
    const bool ok = c >= K && c <= kFCap;
    if (!ok) {   // hand the row to the exact kernel
      if (lane == 0) P.redo[1 + atomicAdd(&P.redo[0], 1)] = b * N + i;
      continue;
    }
    uint32_t* ck = ckey + rr * kFCap;
    uint32_t* ci = cidx + rr * kFCap;
    if (c <= 2 * kWave) filter_select_row<2, kFCap>(P, ck, ci, c, b, i, lane);
    else if (c <= 4 * kWave) filter_select_row<4, kFCap>(P, ck, ci, c, b, i, lane);
    else if (kFCap <= 8 * kWave || c <= 8 * kWave) filter_select_row<8, kFCap>(P, ck, ci, c, b, i, lane);
    else if (c <= 12 * kWave) filter_select_row<12, kFCap>(P, ck, ci, c, b, i, lane);
    else filter_select_row<16, kFCap>(P, ck, ci, c, b, i, lane);
  }
}

// ---- bf16 planes of the points: every fp32 coordinate split exactly into three bf16 values (csrc/bf16x6.h), stored in
// the order the MFMA fragments are read: per sample and plane [tile of 16 points][unit = 8 channels][point in tile], 16
// bytes each -- a wave's fragment load (lane = (point in tile, unit within the 32-channel block)) is then ONE contiguous
// kilobyte.  A 256-thread block owns 256 / units consecutive points x all units: reads are coalesced along the points
// of a channel row, writes along the points of a tile; 6 bytes per coordinate (12.6 MB for B = 8, N = 4096, C = 64).
// The same pass leaves |x_j|^2 (8-channel fma chains, summed over the units in order: deterministic) and resets the
// redo counter: the filter path needs no other pre-pass.  Points past N (the last tile) repeat point N - 1: never
// selected (their columns are masked), but finite.
__global__ __launch_bounds__(256) void knn_planes_kernel(const float* __restrict__ x, int64_t sb, int64_t sc, int B,
                                                         int C, int N, int Np, i4v* __restrict__ planes,
                                                         float* __restrict__ sqnorm, int* __restrict__ redo) {
  __shared__ float part[256];
  const int units = C / 8;                       // 4 or 8
  const int ppb = 256 / units;                   // points per block
  const int blocks_per_sample = (Np + ppb - 1) / ppb;
  const int b = blockIdx.x / blocks_per_sample;
  const int p0 = (blockIdx.x % blocks_per_sample) * ppb;
  const int u = threadIdx.x / ppb, pl = threadIdx.x % ppb;
  const int pt = p0 + pl;
  if (blockIdx.x == 0 && threadIdx.x == 0) redo[0] = 0;
  const int64_t per_sample = static_cast<int64_t>(Np) * units;
  const int64_t total = per_sample * B;
  float sq = 0.f;
  if (pt < Np) {
    const float* xp = x + b * sb + min(pt, N - 1);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = xp[static_cast<int64_t>(8 * u + e) * sc];
#pragma unroll
    for (int e = 0; e < 8; ++e) sq = fmaf(v[e], v[e], sq);
    i4v h, m, l;
    eg_split3(f4v{v[0], v[1], v[2], v[3]}, f4v{v[4], v[5], v[6], v[7]}, h, m, l);
    const int64_t o = b * per_sample + (static_cast<int64_t>(pt >> 4) * units + u) * 16 + (pt & 15);
    planes[o] = h;
    planes[total + o] = m;
    planes[2 * total + o] = l;
  }
  part[threadIdx.x] = sq;
  __syncthreads();
  if (threadIdx.x < ppb && pt < N) {
    float t = 0.f;
    for (int q = 0; q < units; ++q) t += part[q * ppb + threadIdx.x];
    sqnorm[static_cast<int64_t>(b) * N + pt] = t;
  }
}

// ---- the filter kernel with the distance pass on the bf16 matrix pipe --------------------------------------------------
// Same structure as knn_filter_kernel (16 query rows per 1024-thread workgroup, per-row candidate lists in LDS, exact
// select per row, rows that miss [K, CAP] go to the exact kernel), but
//   * the inner products come from v_mfma_f32_16x16x32_bf16 on pre-split operands: six products per 16x16x32 block
//     (bf16x6.h: fp32-faithful, max error / sum|a||b| = 1.7e-7).  The fp32 MFMA of the kernel above issues on the
//     VECTOR ALU port at the vector rate (SQ_VALU_MFMA_BUSY 20 % of the kernel, additive with the 65 % of VALU work of
//     the append / select code, profiles/r01_knn_filter_counters.md); the bf16 pipe is 16x faster per flop and runs
//     beside the other waves' VALU instructions;
//   * operands are read as ready-made 16-byte fragments (no per-chunk LDS reads of the query rows, no fp32 split in the
//     loop): the 16 query rows' fragments live in 24 registers for the whole kernel;
//   * blocks are mapped sample-minor (sample = blockIdx % B): block v runs on XCD v % 8, so with B = 8 every XCD's L2
//     holds ONE sample's planes (1.5 MB) instead of all eight (12.6 MB against 4 MB of L2).
// Distances: D = (|x_i|^2 + (-2 ip)) + |x_j|^2 with ip from the six-product sum instead of the channel-ordered fma chain:
// the same value up to fp32 rounding (not bit for bit); a row is ranked entirely by ONE of the two evaluations (this
// kernel's, or the exact kernel's chain when the row is redone), never by a mixture.
template <int kFCap, int KC, int NW>
__global__ __launch_bounds__(NW * kWave, 4) void knn_filter_bf16_kernel(const KnnParams P) {
#ifdef DGCN_KNN_AGG_ALWAYS
  constexpr bool kFAggAppend = true;
#else
  constexpr bool kFAggAppend = kFCap > 512;      // wave-aggregated appends where a quarter of the candidates are hits
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int TM = kFTM;
  const int N = P.N, K = P.K;
  float* sq = reinterpret_cast<float*>(smem);                       // [16]
  float* tauf = sq + TM;                                            // [16] threshold as a distance
  int* cnt = reinterpret_cast<int*>(tauf + TM);                     // [16]
  uint32_t* ckey = reinterpret_cast<uint32_t*>(cnt + TM);           // [16][kFCap]
  uint32_t* cidx = ckey + TM * kFCap;                               // [16][kFCap]

  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = tid >> 6;
  const int b = blockIdx.x % P.B;                                   // sample-minor: see above
  const int tile = blockIdx.x / P.B;
  const int i0 = tile * TM;
  constexpr int UNITS = 4 * KC;                                     // 16-byte units per point and plane (C = 32 KC)
  const int Np = (N + 15) & ~15;                                    // points per sample in the planes (whole tiles)
  const int64_t plane = static_cast<int64_t>(P.B) * Np * UNITS;     // units per plane
  const i4v* pb = P.planes + static_cast<int64_t>(b) * Np * UNITS;  // plane 0 of this sample; + p * plane

  if (tid < TM) {
    const int64_t row = static_cast<int64_t>(b) * N + min(i0 + tid, N - 1);
    sq[tid] = P.sqnorm[row];
    cnt[tid] = 0;
  }
  const int li = lane & 15, lk = lane >> 4;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const unsigned long long below = (1ull << lane) - 1ull;
  const float* sqn = P.sqnorm + static_cast<int64_t>(b) * N;
  // A fragments: lane (m = li, kq = lk) holds channels 32 kb + 8 lk .. + 7 of query row i0 + li, three planes
  i4v a[KC][3];
  // tile layout [tile][unit][point in tile]: lane (li, lk) of k-block kb reads unit 4 kb + lk of point li
  auto frag = [&](int ctile, int kb, int p) -> i4v {
    return pb[p * plane + static_cast<int64_t>(ctile) * UNITS * 16 + (4 * kb + lk) * 16 + li];
  };
  constexpr int kColStride = NW * 64;
  // one step = the three plane fragments of TWO 16-candidate tiles for one 32-channel block: 6 loads, 12 MFMAs that
  // alternate between the two accumulators (no back-to-back dependent MFMAs); the next step's loads are in flight
  // while this step's MFMAs run.  Four steps (2 tile pairs x KC) per 64-candidate block when KC = 2.
  auto load_step = [&](int col0, int pair, int kb, i4v (&f)[2][3]) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int ct = min((col0 >> 4) + 2 * pair + t, (Np >> 4) - 1);      // candidate tile (clamped past the end)
#pragma unroll
      for (int p = 0; p < 3; ++p) f[t][p] = frag(ct, kb, p);
    }
  };
  constexpr int pa[6] = {0, 0, 1, 0, 2, 1};       // a1 b1, a1 b2, a2 b1, a1 b3, a3 b1, a2 b2
  constexpr int pbb[6] = {0, 1, 0, 2, 0, 1};
  auto mfma_step = [&](f32x4 (&acc)[4], int pair, int kb, const i4v (&f)[2][3]) {
#pragma unroll
    for (int s6 = 0; s6 < 6; ++s6) {
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[2 * pair + t] = eg_mfma_bf16(a[kb][pa[s6]], f[t][pbb[s6]], acc[2 * pair + t]);
    }
  };
  constexpr int NSTEP = 2 * KC;
  i4v fA[2][3], fB[2][3];

  // ---- prologue: ONE memory latency for everything the workgroup needs before its first append -- the query
  // fragments, the fragments of this wave's two SAMPLE tiles and the first step of the main loop are requested together.
  // Per-row threshold tau_r = the sample_rank-th smallest of row r's distances to 512 sampled candidates: 32 tiles
  // spread over the sample (rotated per query tile), two per wave, on the same matrix-pipe path.  (The fp32 kernel
  // above takes tau from knn_prep_kernel, a separate latency-bound launch of ~43 us; twice the samples also halve the
  // spread of the candidate count, so the short lists serve K = 224 and fewer rows need the exact pass.)
  {
    uint32_t* skeys = ckey;                       // [16][512], aliasing the (still empty) candidate lists
    const int NT = Np >> 4;
    const int rot = (tile * 7) % max(NT / 32, 1);
    constexpr int NPASS = 16 / NW;                // sample tiles: 32 per workgroup, two per wave and pass
#pragma unroll
    for (int kb = 0; kb < KC; ++kb) {
#pragma unroll
      for (int p = 0; p < 3; ++p) a[kb][p] = frag(tile, kb, p);
    }
    load_step(wave * 64, 0, 0, fA);
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      int sct[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        sct[t] = static_cast<int>((static_cast<int64_t>(2 * (wave + pass * NW) + t) * NT / 32 + rot) % NT);
      }
      i4v sf[KC][2][3];
#pragma unroll
      for (int kb = 0; kb < KC; ++kb) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          sf[kb][0][p] = frag(sct[0], kb, p);
          sf[kb][1][p] = frag(sct[1], kb, p);
        }
      }
      float sjv[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) sjv[t] = (sct[t] * 16 + li) < N ? sqn[sct[t] * 16 + li] : 0.f;
      if (pass == 0) __syncthreads();             // sq[], cnt[] visible
      f32x4 sacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int kb = 0; kb < KC; ++kb) {
#pragma unroll
        for (int s6 = 0; s6 < 6; ++s6) {
#pragma unroll
          for (int t = 0; t < 2; ++t) sacc[t] = eg_mfma_bf16(a[kb][pa[s6]], sf[kb][t][pbb[s6]], sacc[t]);
        }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bool inn = (sct[t] * 16 + li) < N;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int r = lk * 4 + reg;
          skeys[r * 512 + (2 * (wave + pass * NW) + t) * 16 + li] =
              inn ? key_of((sq[r] + (-2.f * sacc[t][reg])) + sjv[t]) : 0xFFFFFFFFu;
        }
      }
    }
    __syncthreads();
    for (int trow = wave; trow < TM; trow += NW) {
      // tau_r = an upper bound of the sample_rank-th smallest of the 512 sample keys, one bucket of a 256-bucket
      // histogram wide (a threshold may overshoot by a couple of samples; the 20-step bisection that used to find it
      // exactly cost 1,240 vector instructions per wave, 40 % of the K = 16 kernel): key range, histogram in LDS,
      // prefix sum, the lane whose eight buckets hold the rank.  The range starts at the SECOND-smallest key: one row
      // in eight has the query point itself among its samples (distance ~0: a key far below all others), which would
      // stretch the buckets to several units of distance; keys below the range share bucket 0.
      uint32_t ks[8];
      uint32_t mn = 0xFFFFFFFFu, mx = 0u;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        ks[q] = skeys[trow * 512 + q * kWave + lane];
        mn = min(mn, ks[q]);
        mx = max(mx, ks[q] == 0xFFFFFFFFu ? 0u : ks[q]);            // (padding of a last, partly filled tile)
      }
      mn = wave_min_u32(mn);
      mx = ~wave_min_u32(~mx);
      uint32_t lo = 0xFFFFFFFFu;
#pragma unroll
      for (int q = 0; q < 8; ++q) lo = min(lo, ks[q] > mn ? ks[q] : 0xFFFFFFFFu);
      lo = wave_min_u32(lo);
      if (lo > mx) lo = mn;                                         // every sample has the same key
      const uint32_t span = mx - lo;
      const int shift = max(0, 24 - static_cast<int>(__builtin_clz(span | 1u)));   // (span >> shift) < 256
      uint32_t* hist = cidx + trow * 512;                           // [512] in the (still unused) id lists
      *reinterpret_cast<uint4*>(hist + lane * 8) = make_uint4(0u, 0u, 0u, 0u);
      *reinterpret_cast<uint4*>(hist + lane * 8 + 4) = make_uint4(0u, 0u, 0u, 0u);
      wave_lds_sync();
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (ks[q] != 0xFFFFFFFFu) atomicAdd(&hist[ks[q] < lo ? 0u : 1u + ((ks[q] - lo) >> shift)], 1u);
      wave_lds_sync();
      uint32_t hc[8];
      {
        const uint4 h0 = *reinterpret_cast<const uint4*>(hist + lane * 8);
        const uint4 h1 = *reinterpret_cast<const uint4*>(hist + lane * 8 + 4);
        hc[0] = h0.x; hc[1] = h0.y; hc[2] = h0.z; hc[3] = h0.w; hc[4] = h1.x; hc[5] = h1.y; hc[6] = h1.z; hc[7] = h1.w;
      }
      uint32_t tot = 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) tot += hc[q];
      const uint32_t incl = static_cast<uint32_t>(wave_scan_incl(static_cast<int>(tot)));
      const uint32_t rank = static_cast<uint32_t>(P.sample_rank);
      const bool mine = incl - tot < rank && rank <= incl;          // exactly one lane (rank <= the valid samples)
      uint32_t run = incl - tot, qsel = 7u;
#pragma unroll
      for (int q = 7; q >= 0; --q) {                                // first bucket whose inclusive prefix reaches the rank
        uint32_t upto = run;
#pragma unroll
        for (int u = 0; u <= q; ++u) upto += hc[u];
        if (rank <= upto) qsel = static_cast<uint32_t>(q);
      }
      uint32_t bq = 8u * lane + qsel;     // (formed once: eight per-lane 8 lane + q constants would stay live, and spill)
      const unsigned long long mm = __ballot(mine);
      const int src = mm ? static_cast<int>(__builtin_ctzll(mm)) : 63;
      bq = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(bq), src));
      // bucket 0 = keys below lo; bucket b >= 1 = keys lo + ((b - 1) << shift) .. lo + (b << shift) - 1
      const unsigned long long edge = static_cast<unsigned long long>(lo) + (static_cast<unsigned long long>(bq) << shift) - 1ull;
      const uint32_t tv = mm ? static_cast<uint32_t>(min(edge, 0xFFFFFFFEull)) : 0xFFFFFFFEu;
      // the float with that key (key_of is monotone: distance <= tauf  <=>  key <= tv)
      if (lane == 0) tauf[trow] = __uint_as_float((tv & 0x80000000u) ? (tv & 0x7FFFFFFFu) : ~tv);
    }
    __syncthreads();
  }
#if defined(KNNF_STOP_AFTER) && KNNF_STOP_AFTER == 1
  return;
#endif

  for (int col0 = wave * 64; col0 < N; col0 += kColStride) {
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < NSTEP; st += 2) {
      // step st (in fA), prefetch st + 1 into fB; then step st + 1, prefetch st + 2 (or the next block's step 0) into fA
      load_step(col0, (st + 1) / KC, (st + 1) % KC, fB);
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(acc, st / KC, st % KC, fA);
      __builtin_amdgcn_sched_barrier(0);
      const bool more = st + 2 < NSTEP;
      load_step(more ? col0 : col0 + kColStride, more ? (st + 2) / KC : 0, more ? (st + 2) % KC : 0, fA);
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(acc, (st + 1) / KC, (st + 1) % KC, fB);
      __builtin_amdgcn_sched_barrier(0);
    }
#ifdef KNNF_NO_APPEND
    {
      float tsum = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) tsum += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
      if (tsum == 123.456f) cnt[0] = 1;
      continue;
    }
#endif
    // acc[t][reg] = <x_row, x_col> for row = lk*4 + reg, col = col0 + 16 t + li
    float sj[4];
    bool in[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int c = col0 + 16 * t + li;
      in[t] = c < N;
      sj[t] = in[t] ? sqn[c] : 0.f;
    }
    // Append the candidates below the row's threshold.  The compare runs on the distance itself (tauf = the float
    // whose key is tau: float order == key order).  Round 5: ONE LDS atomic per (row, 64-candidate block) instead of one
    // per hit -- the 16-lane group of a row counts its hits of the four tiles from four ballots, its first lane reserves
    // the positions with a single ds_add_rtn (four different counters per instruction: no same-address serialisation,
    // which cost 36 M bank-conflict cycles per launch at K = 432 where a quarter of the candidates are hits), the hits
    // take base + their rank among the group's hits.  A block without a hit in any of the four rows of this register
    // (K = 16: 98 % of the candidates miss) is skipped with one wave-uniform branch.  The lists are unordered sets, the
    // select ranks by (key, id): the emitted ids do not depend on the order of the appends.
    if constexpr (kFAggAppend) {
      const int seg = lane & ~15;                                       // first lane of this lane's 16-lane group
      const unsigned long long segbelow = ((1ull << (lane & 15)) - 1ull) << seg;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int r = lk * 4 + reg;  // the four 16-lane groups hold four different rows
        const float tf = tauf[r];
        const float sr = sq[r];
        const int self = P.exclude_self ? i0 + r : -1;
        float dist[4];
        bool hit[4];
        unsigned long long bal[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          dist[t] = (sr + (-2.f * acc[t][reg])) + sj[t];
          hit[t] = in[t] && dist[t] <= tf && (col0 + 16 * t + li) != self;
          bal[t] = __ballot(hit[t]);
        }
        if ((bal[0] | bal[1] | bal[2] | bal[3]) == 0ull) continue;     // wave-uniform
        int nseg[4], total = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          nseg[t] = __popcll((bal[t] >> seg) & 0xFFFFull);
          total += nseg[t];
        }
        int base = 0;
        if (li == 0 && total > 0) base = atomicAdd(&cnt[r], total);
        base = __shfl(base, seg);
        int before = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (hit[t]) {
            const int pos = base + before + __popcll(bal[t] & segbelow);
            if (pos < kFCap) {
              ckey[r * kFCap + pos] = key_of(dist[t]);
              cidx[r * kFCap + pos] = static_cast<uint32_t>(col0 + 16 * t + li);
            }
          }
          before += nseg[t];
        }
      }
    } else {
      // one LDS atomic per hit (round 4's form): cheaper in vector instructions when hits are rare (K = 16: 2 % of the
      // candidates), serialises on the row's counter when they are not
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int r = lk * 4 + reg;
        const float tf = tauf[r];
        const float sr = sq[r];
        const int self = P.exclude_self ? i0 + r : -1;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float dist = (sr + (-2.f * acc[t][reg])) + sj[t];
          const int c = col0 + 16 * t + li;
          if (in[t] && dist <= tf && c != self) {
            const int pos = atomicAdd(&cnt[r], 1);
            if (pos < kFCap) {
              ckey[r * kFCap + pos] = key_of(dist);
              cidx[r * kFCap + pos] = static_cast<uint32_t>(c);
            }
          }
        }
      }
    }
  }
  __syncthreads();
#if defined(KNNF_STOP_AFTER) && KNNF_STOP_AFTER == 2
  return;
#endif

  // ---- per-row select on the candidate lists (as in knn_filter_kernel) ----
  for (int rr = wave; rr < TM; rr += NW) {
    const int i = i0 + rr;
    if (i >= N) continue;  // wave-uniform
    const int c = cnt[rr];
    const bool ok = c >= K && c <= kFCap;
    if (!ok) {   // hand the row to the exact kernel
      if (lane == 0) P.redo[1 + atomicAdd(&P.redo[0], 1)] = b * N + i;
      continue;
    }
    uint32_t* ck = ckey + rr * kFCap;
    uint32_t* ci = cidx + rr * kFCap;
    if (c <= 2 * kWave) filter_select_row<2, kFCap>(P, ck, ci, c, b, i, lane);
    else if (c <= 4 * kWave) filter_select_row<4, kFCap>(P, ck, ci, c, b, i, lane);
    else if (kFCap <= 8 * kWave || c <= 8 * kWave) filter_select_row<8, kFCap>(P, ck, ci, c, b, i, lane);
    else if (c <= 12 * kWave) filter_select_row<12, kFCap>(P, ck, ci, c, b, i, lane);
    else filter_select_row<16, kFCap>(P, ck, ci, c, b, i, lane);
  }
}

// ---- round 6: 32 query rows per workgroup, candidate lists in GLOBAL memory -------------------------------------------
// What bounded knn_filter_bf16_kernel (profiles/r06_knn_phases.md: variant builds on one box, K = 16 / 432): the distance
// pass WITHOUT its appends takes 107 - 115 us for 46 us of matrix work -- every 16-row workgroup streams its sample's three
// planes (1.5 MB) out of the L2, 3.2 GB per launch = 30 TB/s, the L2's peak.  The lists of a 32-row workgroup do not fit
// the LDS (32 x 1024 x 8 B), so they move to global memory -- appends are 2 - 25 % of the candidates, 8 bytes each, and
// the workgroup that wrote a row's list is the one that reads it back (L2-resident) -- which frees the LDS for everything
// but the sample keys and gives every K the 1024-entry capacity (the 512-entry kernels sent K = 128 .. 240 through a
// 32 - 70 us exact redo per layer).  Per wave: the A fragments of BOTH 16-row tiles stay in registers (48 VGPRs at
// C = 64) and every candidate fragment fetched from the L2 feeds two MFMAs: 1.6 GB per launch.  One candidate tile per
// step (3 plane fragments per 32-channel block, 12 MFMAs alternating between the two row tiles' accumulators), the next
// tile's fragments in flight while this one's run.  Thresholds, append rule, select, redo list and therefore the emitted
// ids are those of knn_filter_bf16_kernel: same six-product distances, lists are unordered sets, the select ranks by
// (key, id).
constexpr int kF2Rows = 32;
constexpr int kF2Cap = 1024;
constexpr int kF2Waves = 8;

template <int KC, bool EXCL>
__global__ __launch_bounds__(kF2Waves * kWave, 4) void knn_filter2_kernel(const KnnParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int TM = kF2Rows, NW = kF2Waves, CAP = kF2Cap;
  const int N = P.N, K = P.K;
  float* sq = reinterpret_cast<float*>(smem);                       // [32]
  float* tauf = sq + TM;                                            // [32] threshold as a distance
  int* cnt = reinterpret_cast<int*>(tauf + TM);                     // [32]
  uint32_t* big = reinterpret_cast<uint32_t*>(cnt + TM + 32);       // [32][512] sample keys, then [8 waves][1024] hit stage

  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = tid >> 6;
  const int b = blockIdx.x % P.B;                                   // sample-minor: one sample's planes per XCD's L2
  const int tile2 = blockIdx.x / P.B;
  const int i0 = tile2 * TM;
  constexpr int UNITS = 4 * KC;
  const int Np = (N + 15) & ~15;
  const int NT = Np >> 4;
  const int64_t plane = static_cast<int64_t>(P.B) * Np * UNITS;
  const i4v* pb = P.planes + static_cast<int64_t>(b) * Np * UNITS;
  const float* sqn = P.sqnorm + static_cast<int64_t>(b) * N;

  if (tid < TM) {
    sq[tid] = sqn[min(i0 + tid, N - 1)];
    cnt[tid] = 0;
  }
  const int li = lane & 15, lk = lane >> 4;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  // fragment address = (uniform plane base of this sample) + (32-bit per-lane byte offset of the tile) + (immediate
  // 1 KiB per 32-channel block): one offset VGPR per tile in flight instead of a 64-bit address per load (the launcher
  // checks that a sample's plane stays below 4 GiB)
  const char* pbase[3] = {reinterpret_cast<const char*>(pb), reinterpret_cast<const char*>(pb + plane),
                          reinterpret_cast<const char*>(pb + 2 * plane)};
  const uint32_t lane_off = static_cast<uint32_t>(lk * 16 + li) * 16u;
  auto frag = [&](int ctile, int kb, int p) -> i4v {
    const uint32_t off = static_cast<uint32_t>(ctile) * (UNITS * 256u) + lane_off;
    return *reinterpret_cast<const i4v*>(pbase[p] + off + kb * 1024);
  };
  constexpr int pa[6] = {0, 0, 1, 0, 2, 1};       // a1 b1, a1 b2, a2 b1, a1 b3, a3 b1, a2 b2
  constexpr int pbb[6] = {0, 1, 0, 2, 0, 1};
  i4v a[2][KC][3];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const int t = min(2 * tile2 + rt, NT - 1);
#pragma unroll
    for (int kb = 0; kb < KC; ++kb) {
#pragma unroll
      for (int p = 0; p < 3; ++p) a[rt][kb][p] = frag(t, kb, p);
    }
  }

  // ---- per-row thresholds from 512 sampled candidates (32 tiles spread over the sample, four per wave) ----
  {
    uint32_t* skeys = big;                        // [32][512]
    const int rot = (tile2 * 7) % max(NT / 32, 1);
    // sample tile q of this wave = sample slot 4 wave + q of 32; two tiles' fragments in flight (the register budget of
    // the distance pass below: A fragments of both row tiles + two candidate tiles)
    i4v sf0[KC][3], sf1[KC][3];
    float sv0, sv1;
    auto sample_tile = [&](int q) -> int {
      return static_cast<int>((static_cast<int64_t>(4 * wave + q) * NT / 32 + rot) % NT);
    };
    auto load_sample = [&](int q, i4v (&f)[KC][3], float& sv) {
      const int ct = sample_tile(q);
#pragma unroll
      for (int kb = 0; kb < KC; ++kb) {
#pragma unroll
        for (int p = 0; p < 3; ++p) f[kb][p] = frag(ct, kb, p);
      }
      sv = (ct * 16 + li) < N ? sqn[ct * 16 + li] : 0.f;
    };
    auto do_sample = [&](int q, const i4v (&f)[KC][3], float sv) {
      f32x4 sacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int kb = 0; kb < KC; ++kb) {
#pragma unroll
        for (int s6 = 0; s6 < 6; ++s6) {
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) sacc[rt] = eg_mfma_bf16(a[rt][kb][pa[s6]], f[kb][pbb[s6]], sacc[rt]);
        }
      }
      const bool inn = (sample_tile(q) * 16 + li) < N;
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int r = rt * 16 + lk * 4 + reg;
          skeys[r * 512 + (4 * wave + q) * 16 + li] = inn ? key_of((sq[r] + (-2.f * sacc[rt][reg])) + sv) : 0xFFFFFFFFu;
        }
      }
    };
    load_sample(0, sf0, sv0);
    load_sample(1, sf1, sv1);
    __syncthreads();                              // sq[], cnt[] visible
    do_sample(0, sf0, sv0);
    load_sample(2, sf0, sv0);
    do_sample(1, sf1, sv1);
    load_sample(3, sf1, sv1);
    do_sample(2, sf0, sv0);
    do_sample(3, sf1, sv1);
    __syncthreads();
    for (int trow = wave; trow < TM; trow += NW) {
      // as in knn_filter_bf16_kernel: the upper edge of the bucket (256 buckets between the second-smallest and the
      // largest sample key) that holds the sample rank; the histogram reuses the row's own key storage
      uint32_t ks[8];
      uint32_t mn = 0xFFFFFFFFu, mx = 0u;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        ks[q] = skeys[trow * 512 + q * kWave + lane];
        mn = min(mn, ks[q]);
        mx = max(mx, ks[q] == 0xFFFFFFFFu ? 0u : ks[q]);
      }
      mn = wave_min_u32(mn);
      mx = ~wave_min_u32(~mx);
      uint32_t lo = 0xFFFFFFFFu;
#pragma unroll
      for (int q = 0; q < 8; ++q) lo = min(lo, ks[q] > mn ? ks[q] : 0xFFFFFFFFu);
      lo = wave_min_u32(lo);
      if (lo > mx) lo = mn;
      const uint32_t span = mx - lo;
      const int shift = max(0, 24 - static_cast<int>(__builtin_clz(span | 1u)));   // (span >> shift) < 256
      uint32_t* hist = skeys + trow * 512;                          // the keys are in registers
      wave_lds_sync();
      *reinterpret_cast<uint4*>(hist + lane * 8) = make_uint4(0u, 0u, 0u, 0u);
      *reinterpret_cast<uint4*>(hist + lane * 8 + 4) = make_uint4(0u, 0u, 0u, 0u);
      wave_lds_sync();
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (ks[q] != 0xFFFFFFFFu) atomicAdd(&hist[ks[q] < lo ? 0u : 1u + ((ks[q] - lo) >> shift)], 1u);
      wave_lds_sync();
      uint32_t hc[8];
      {
        const uint4 h0 = *reinterpret_cast<const uint4*>(hist + lane * 8);
        const uint4 h1 = *reinterpret_cast<const uint4*>(hist + lane * 8 + 4);
        hc[0] = h0.x; hc[1] = h0.y; hc[2] = h0.z; hc[3] = h0.w; hc[4] = h1.x; hc[5] = h1.y; hc[6] = h1.z; hc[7] = h1.w;
      }
      uint32_t tot = 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) tot += hc[q];
      const uint32_t incl = static_cast<uint32_t>(wave_scan_incl(static_cast<int>(tot)));
      const uint32_t rank = static_cast<uint32_t>(P.sample_rank);
      const bool mine = incl - tot < rank && rank <= incl;
      uint32_t run = incl - tot, qsel = 7u;
#pragma unroll
      for (int q = 7; q >= 0; --q) {
        uint32_t upto = run;
#pragma unroll
        for (int u = 0; u <= q; ++u) upto += hc[u];
        if (rank <= upto) qsel = static_cast<uint32_t>(q);
      }
      uint32_t bq = 8u * lane + qsel;
      const unsigned long long mm = __ballot(mine);
      const int src = mm ? static_cast<int>(__builtin_ctzll(mm)) : 63;
      bq = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(bq), src));
      const unsigned long long edge = static_cast<unsigned long long>(lo) + (static_cast<unsigned long long>(bq) << shift) - 1ull;
      const uint32_t tv = mm ? static_cast<uint32_t>(min(edge, 0xFFFFFFFEull)) : 0xFFFFFFFEu;
      // rows past the end of the cloud (last workgroup of a sample) never take a candidate
      if (lane == 0)
        tauf[trow] = (i0 + trow < N) ? __uint_as_float((tv & 0x80000000u) ? (tv & 0x7FFFFFFFu) : ~tv) : DGCN_NEG_INF;
    }
    __syncthreads();
  }
#if defined(KNNF_STOP_AFTER) && KNNF_STOP_AFTER == 1
  if (tid < TM && i0 + tid < N) P.list_cnt[static_cast<int64_t>(b) * N + i0 + tid] = -1;   // (select + redo skipped)
  return;
#endif

  // ---- distance pass.  A wave takes PAIRS of candidate tiles (pair q = tiles 2q, 2q + 1; pairs wave, wave + 8, ...) and
  // walks a pair one 32-channel block at a time: 6 fragment loads (two tiles x three planes), 24 MFMAs into FOUR
  // accumulators (2 row tiles x 2 candidate tiles) issued round-robin, so that an MFMA depends on the one four back --
  // with two accumulators (rounds 4 - 5, and this kernel's first version) every MFMA waits for the one two back and an
  // issue slot of another wave in between costs the dependent one ~43 cycles (MI355X_MICROARCH.md, per-instruction
  // constants): 210 cycles per MFMA and wave at four waves per SIMD, measured on both kernels
  // (profiles/r06_knn_filter_counters.md).  The next step's fragments are in flight while this step's MFMAs run.
  {
    const int NP = (NT + 1) >> 1;                                   // tile pairs of the sample
    const int npair = (NP - wave + NW - 1) / NW;                    // pairs of this wave
    constexpr int NSTEP = KC;                                       // steps per pair
    i4v fA[2][3], fB[2][3];
    auto load_step = [&](int q, int kb, i4v (&f)[2][3]) {
#ifdef KNNF_NO_LOADS
      if (q > 0) return;
#endif
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int ct = min(2 * (wave + q * NW) + t, NT - 1);
#pragma unroll
        for (int p = 0; p < 3; ++p) f[t][p] = frag(ct, kb, p);
      }
    };
    auto mfma_step = [&](f32x4 (&acc)[2][2], int kb, const i4v (&f)[2][3]) {
#ifdef KNNF_NO_MFMA2
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int p = 0; p < 3; ++p) acc[0][t][p] += __int_as_float(f[t][p][0] ^ f[t][p][3]);
      }
      return;
#endif
#pragma unroll
      for (int s6 = 0; s6 < 6; ++s6) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) acc[rt][t] = eg_mfma_bf16(a[rt][kb][pa[s6]], f[t][pbb[s6]], acc[rt][t]);
        }
      }
    };
    // Epilogue of a pair: 16 (row, candidate) distances per lane.  Hits are 2 - 25 % of them, so a branch per distance
    // (rounds 4 - 5's form, and this kernel's first) runs its body -- an LDS atomic, the key, a store -- for 1.4 active
    // lanes on average, 11 to 32 times per pair: 18 instructions per distance, 118 us per launch once the loads no longer
    // hide it (profiles/r06_knn_phases.md; a per-wave LDS stage in front of the stores did not help: the cost is
    // instruction issue, not the stores).  Here the 16 compares only set bits; the distances go to a lane-private LDS
    // column; then the wave loops over ROUNDS -- every lane with a bit left takes its lowest one -- so the body runs
    // 2 - 3 times per pair at K = 16 (9 at K = 432) with as many lanes as have hits.  Lists are unordered sets: the order
    // of the appends does not reach the output.
    float* dl = reinterpret_cast<float*>(big) + wave * (16 * kWave);  // [16 slots][64 lanes], where the sample keys were
    // (the list base is formed here, not at the top of the kernel: a 64-bit value live across the prologue was the one
    //  spilled VGPR pair of the C = 64 instantiations)
    char* const lbase = reinterpret_cast<char*>(P.lists + (static_cast<int64_t>(b) * N + i0) * CAP);
    auto epilogue = [&](int q, const f32x4 (&acc)[2][2], const float (&sj)[2]) {
#ifdef KNNF_NO_APPEND
      {
        float tsum = sj[0] + sj[1];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
          for (int t = 0; t < 2; ++t) tsum += acc[rt][t][0] + acc[rt][t][1] + acc[rt][t][2] + acc[rt][t][3];
        }
        if (tsum == 123.456f) cnt[0] = 1;
        return;
      }
#endif
      const int c0 = 2 * (wave + q * NW) * 16 + li;                 // candidate of t = 0; t = 1: + 16
      uint32_t bits = 0u;
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const float4 tf4 = *reinterpret_cast<const float4*>(tauf + rt * 16 + lk * 4);
        const float4 sr4 = *reinterpret_cast<const float4*>(sq + rt * 16 + lk * 4);
#ifdef KNNF_FULL_WAIT
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        const float tfv[4] = {tf4.x, tf4.y, tf4.z, tf4.w};
        const float srv[4] = {sr4.x, sr4.y, sr4.z, sr4.w};
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int k = rt * 8 + reg * 2 + t;                      // slot: row = rt * 16 + lk * 4 + reg, candidate c0 + 16 t
            const float dist = (srv[reg] + (-2.f * acc[rt][t][reg])) + sj[t];
            bool hit = dist <= tfv[reg];

            bits |= hit ? (1u << k) : 0u;
            dl[k * kWave + lane] = dist;
          }
        }
      }
      while (__ballot(bits != 0u) != 0ull) {                        // rounds: wave-uniform trip count
        if (bits != 0u) {
          const int k = __builtin_ctz(bits);
          bits &= bits - 1u;
          const float dist = dl[k * kWave + lane];                   // (written by this lane: no barrier needed)
          const int r = (k >> 3) * 16 + lk * 4 + ((k >> 1) & 3);
          const int c = c0 + ((k & 1) << 4);
          if (EXCL && c == i0 + r) continue;                          // exclude_self: the query point is not a candidate
          const int pos = atomicAdd(&cnt[r], 1);
          if (pos < CAP)
            *reinterpret_cast<uint2*>(lbase + static_cast<uint32_t>(r * CAP + pos) * 8u) =
                make_uint2(key_of(dist), static_cast<uint32_t>(c));
        }
      }
    };
    auto load_sj = [&](int q, float (&sj)[2]) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int c = (2 * (wave + q * NW) + t) * 16 + li;
        sj[t] = c < N ? sqn[min(c, N - 1)] : __builtin_inff();        // columns past N: distance +inf, never a hit
      }
    };
    load_step(0, 0, fA);
#pragma unroll 1
    for (int q = 0; q < npair; ++q) {
      f32x4 acc[2][2];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      float sj[2];
      load_sj(q, sj);                                                 // requested BEFORE the next step's fragments (in-order returns)
      if constexpr (NSTEP == 1) {
        // one step per pair: alternate the buffers by hand over two pairs
        load_step(q + 1, 0, fB);
        __builtin_amdgcn_sched_barrier(0);
        mfma_step(acc, 0, fA);
        __builtin_amdgcn_sched_barrier(0);
        epilogue(q, acc, sj);
        if (++q >= npair) break;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
          for (int t = 0; t < 2; ++t) acc[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        load_sj(q, sj);
        load_step(q + 1, 0, fA);
        __builtin_amdgcn_sched_barrier(0);
        mfma_step(acc, 0, fB);
        __builtin_amdgcn_sched_barrier(0);
        epilogue(q, acc, sj);
      } else {
#pragma unroll
        for (int st = 0; st < NSTEP; st += 2) {
          load_step(q, st + 1, fB);
          __builtin_amdgcn_sched_barrier(0);
          mfma_step(acc, st, fA);
          __builtin_amdgcn_sched_barrier(0);
          const bool more = st + 2 < NSTEP;
          load_step(more ? q : q + 1, more ? st + 2 : 0, fA);
          __builtin_amdgcn_sched_barrier(0);
          mfma_step(acc, st + 1, fB);
          __builtin_amdgcn_sched_barrier(0);
        }
        epilogue(q, acc, sj);
      }
    }
  }
  // the rows' candidate counts, for the select kernel (the end of a kernel is the only fence the lists need: an
  // agent-scope __threadfence() per wave in the first version of this kernel -- lists read back by the same launch --
  // wrote back and invalidated the XCD's L2 once per wave: 100 - 140 us per launch, profiles/r06_knn_phases.md)
  __syncthreads();
#if defined(KNNF_STOP_AFTER)
  if (tid < TM && i0 + tid < N) P.list_cnt[static_cast<int64_t>(b) * N + i0 + tid] = -1;   // (select + redo skipped)
  return;
#endif
  {
    int tr = tid;
    asm volatile("" : "+v"(tr));                // (recomputed here: kept from the kernel's first lines, i0 + tid was the one spilled VGPR)
    if (tr < TM && i0 + tr < N) P.list_cnt[static_cast<int64_t>(b) * N + i0 + tr] = cnt[tr];
  }
}

// ---- the select of knn_filter2_kernel's lists: one wave per query row, any row order, high occupancy --------------------
// (inside the filter kernel the four rows of a wave ran one after the other at 16 waves per CU: 76 - 195 us per launch;
// as its own launch 32,768 independent waves: the same code, profiles/r06_knn_phases.md)
constexpr int kSelWaves = 4;
__global__ __launch_bounds__(kSelWaves * kWave) void knn_select_lists_kernel(const KnnParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int CAP = kF2Cap;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int64_t row = static_cast<int64_t>(blockIdx.x) * kSelWaves + wave;
  if (row >= static_cast<int64_t>(P.B) * P.N) return;              // wave-uniform
  const int b = static_cast<int>(row / P.N), i = static_cast<int>(row % P.N);
  uint32_t* sa = reinterpret_cast<uint32_t*>(smem) + wave * (2 * CAP);
  uint32_t* sb = sa + CAP;
  const int c = P.list_cnt[row];
#if defined(KNNF_STOP_AFTER)
  if (c < 0) return;
#endif
  if (c < P.K || c > CAP) {   // hand the row to the exact kernel
#ifdef KNNF_DEBUG_REDO
    if (lane < P.Kout) P.nn_out[row * P.Kout + lane] = -7 - c;      // (debug builds: mark the row instead)
    return;
#endif
    if (lane == 0) P.redo[1 + atomicAdd(&P.redo[0], 1)] = static_cast<int>(row);
    return;
  }
  const uint2* list = P.lists + row * CAP;
#ifdef KNNF_DEBUG_REDO
  if (lane == 0 && P.ctr_out) P.ctr_out[row * P.Kout + P.Kout - 1] = c;       // (debug builds: the list length, read by the host)
  KnnParams Q = P;
  Q.ctr_out = nullptr;
  if (c <= 2 * kWave) filter_select_list<2, 512>(Q, list, sa, sb, c, b, i, lane);
  else if (c <= 4 * kWave) filter_select_list<4, 512>(Q, list, sa, sb, c, b, i, lane);
  else if (c <= 8 * kWave) filter_select_list<8, 512>(Q, list, sa, sb, c, b, i, lane);
  else if (c <= 12 * kWave) filter_select_list<12, 1024>(Q, list, sa, sb, c, b, i, lane);
  else filter_select_list<16, 1024>(Q, list, sa, sb, c, b, i, lane);
  return;
#endif
  if (c <= 2 * kWave) filter_select_list<2, 512>(P, list, sa, sb, c, b, i, lane);
  else if (c <= 4 * kWave) filter_select_list<4, 512>(P, list, sa, sb, c, b, i, lane);
  else if (c <= 8 * kWave) filter_select_list<8, 512>(P, list, sa, sb, c, b, i, lane);
  else if (c <= 12 * kWave) filter_select_list<12, 1024>(P, list, sa, sb, c, b, i, lane);
  else filter_select_list<16, 1024>(P, list, sa, sb, c, b, i, lane);
}

size_t knn_select_lds_bytes() { return static_cast<size_t>(kSelWaves) * 2u * kF2Cap * 4u; }

// sq / tauf / cnt + the per-wave stage counters, and the sample keys (later: the hit stages)
size_t knn_filter2_lds_bytes() { return (3u * kF2Rows + 32u) * 4u + static_cast<size_t>(kF2Rows) * 512u * 4u; }

size_t knn_filter_bf16_lds_bytes(int cap) { return 3u * kFTM * 4u + static_cast<size_t>(kFTM) * cap * 8u; }

size_t knn_filter_lds_bytes(int C, int cap) {
  return (static_cast<size_t>(C) * kFTM + 3 * kFTM) * 4 + static_cast<size_t>(kFTM) * cap * 8;
}

size_t knn_prep_lds_bytes(int C) {
  return (static_cast<size_t>(C) * kFTM + kFTM) * 4 + static_cast<size_t>(kFTM) * kFSamples * 4;
}

// rank of the ns-sample threshold for a list capacity `cap` (see the comment at the call site)
// Margin of the sample threshold above K, in units of sqrt(unit * K) (unit = N / samples).  The number of candidates
// below the r-th smallest of ns uniformly placed samples is r + BetaBinomial(N - ns, r, ns - r + 1) whatever the data:
// with the 3.2 of rounds 3 - 4 between 0.6 and 5 of the 32,768 rows of a config-2 layer were expected to end with fewer
// than K candidates, and ONE such row costs a launch of the exact kernel with real work (30 - 39 us instead of 4.6 us:
// measured per dilation, profiles/r05_knn_by_dilation.md).  4.0 (4.8 below eight sample ranks, where the relative spread
// is largest) brings the expectation under 0.25 (0.01) rows per layer; where the list capacity is too close for any
// rank to clear both ends (cap 512 from K = 128 on) nothing changes: the mid-point clamp below decides.
inline double knn_z(double ranks_of_k) { return ranks_of_k < 8.0 ? 4.8 : 4.0; }

int knn_sample_rank(int N, int K, int cap, int nsamples = kFSamples) {
  const double ns = nsamples, unit = N / ns;
  const double r0 = K / unit;
  const double sd0 = unit * sqrt(r0 > 1.0 ? r0 : 1.0);
  double target = K + knn_z(r0) * sd0 + 2.0 * unit;
  const double mid = 0.5 * (K + static_cast<double>(cap));
  if (target > mid) target = mid;
  const int r = static_cast<int>(ceil(target / unit));
  return (r >= 1 && r <= 200 * nsamples / kFSamples) ? r : 0;
}

size_t knn_lds_bytes(int TM, int C, int Npad, int Kpad) {
  return (static_cast<size_t>(C) * TM + 16 + static_cast<size_t>(TM) * Npad + static_cast<size_t>(TM) * Kpad) * 4;
}

}  // namespace
}  // namespace dgcn

using namespace dgcn;

// x: (B, C, N) fp32 with element strides (sb, sc, sn).  K = k*dilation neighbours are selected per
// point (self included, ascending distance); positions 0, d, 2d, ... are written, Kout = ceil(K/d).
// nn_out / ctr_out: [B, N, Kout] int64 contiguous (ctr_out may be NULL).
namespace dgcn {
namespace {
// channels the bf16 filter kernel is instantiated for (C = 32 KC; KC = 4 would need 48 registers of query fragments and
// spills under the 128-VGPR budget of 16 waves per workgroup: C = 128 stays on the fp32-MFMA kernel)
inline int knn_bf16_kc(int C) { return (C == 32 || C == 64) ? C / 32 : 0; }
inline size_t knn_ws_head(size_t pts) { return (pts * 12u + 4u + 255u) / 256u * 256u; }
inline size_t knn_lists_bytes(size_t pts) { return pts * static_cast<size_t>(kF2Cap) * 8u + (pts * 4u + 255u) / 256u * 256u; }
inline size_t knn_planes_bytes(int B, int N, int C) {
  return static_cast<size_t>(B) * ((static_cast<size_t>(N) + 15u) & ~static_cast<size_t>(15u)) * static_cast<size_t>(C) * 6u;
}
}  // namespace
}  // namespace dgcn

extern "C" size_t dgcn_knn_dense_workspace_bytes(int32_t B, int32_t N, int32_t C) {
  if (B <= 0 || N <= 0 || C < 0) return 0;      // C == 0: the head only (fp32-MFMA filter kernel, no planes)
  // |x_j|^2 (fp32) and the sample threshold (u32) per point, then the redo list: a counter + up to B*N row ids; then the
  // three bf16 planes of the points (6 bytes per coordinate) when the bf16 filter kernel serves this width
  const size_t pts = static_cast<size_t>(B) * static_cast<size_t>(N);
  const size_t ppts = static_cast<size_t>(B) * ((static_cast<size_t>(N) + 15u) & ~static_cast<size_t>(15u));
  // ... and, behind the planes, the candidate lists of knn_filter2_kernel: 1024 (key, id) pairs per point
  return knn_ws_head(pts) + (knn_bf16_kc(C) ? ppts * static_cast<size_t>(C) * 6u + knn_lists_bytes(pts) : 0u);
}

extern "C" int dgcn_knn_dense_f32(const float* x, int64_t sb, int64_t sc, int64_t sn, int32_t B,
                                  int32_t C, int32_t N, int32_t K, int32_t dilation, int32_t exclude_self,
                                  int64_t* nn_out, int64_t* ctr_out, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  if (!x || !nn_out) return DGCN_E_NULL;
  if (workspace && (reinterpret_cast<uintptr_t>(workspace) & 15u)) return DGCN_E_ALIGN;
  if (B < 0 || C <= 0 || N <= 0 || K <= 0 || dilation <= 0) return DGCN_E_SHAPE;
  if (K > N - (exclude_self ? 1 : 0) || K > 1024) return DGCN_E_SHAPE;            // sorted winners: <= 16 u64 per lane
  if (N > kMaxPerLane * kWave) return DGCN_E_SHAPE;      // 64 keys per lane in the select phase: N <= 4096
  if (B == 0) return DGCN_OK;

  int Kpad = 64;
  while (Kpad < K) Kpad <<= 1;
  const int Npad = (N + 3) / 4 * 4;
  int TM = 8;
  while (TM > 1 && knn_lds_bytes(TM, C, Npad, Kpad) > static_cast<size_t>(kLdsBudget)) TM >>= 1;
  size_t lds = knn_lds_bytes(TM, C, Npad, Kpad);
  if (lds > static_cast<size_t>(kLdsBudget)) return DGCN_E_SHAPE;

  KnnParams P;
  P.x = x; P.sb = sb; P.sc = sc; P.sn = sn;
  P.B = B; P.C = C; P.N = N; P.K = K; P.dilation = dilation;
  P.Kout = (K + dilation - 1) / dilation;
  P.nn_out = nn_out; P.ctr_out = ctr_out;
  P.redo = nullptr;
  P.planes = nullptr;
  P.lists = nullptr;
  P.list_cnt = nullptr;
  P.sqnorm = nullptr;
  P.tau = nullptr;
  P.exclude_self = exclude_self ? 1 : 0;
  P.sample_rank = 0;
  if (N >= 1024 && K <= 512) {    // K > 512 (ResGCN-56's deep blocks): exact full-row selection only
    // Sample rank: the number of candidates below the r-th of 256 sample keys has mean r*N/256 and standard
    // deviation ~ sqrt(r)*N/256.  Aim 3.2 sigma above K, but no higher than the middle of [K, 1024] so that
    // both "too few" and "too many" (list capacity) stay rare; either way the exact path catches the row.
    P.sample_rank = knn_sample_rank(N, K, 16 * kWave);
  }
  int tiles = (N + TM - 1) / TM;
  dim3 grid(static_cast<unsigned>(B) * tiles);
  const dim3 block(kKnnThreads);
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e = hipSuccess;
  const bool vec4 = (sn == 1) && (N % 4 == 0) && (sc % 4 == 0) && (sb % 4 == 0) &&
                    ((reinterpret_cast<uintptr_t>(x) & 15u) == 0);
  // Candidate-filter fast path: 16 rows per workgroup; rows it cannot finish are redone below.
  // small lists (two workgroups per CU) when the 3.2-sigma target fits under 512 with margin
  const bool small_lists = K + 3.2 * (N / 256.0) * sqrt(K / (N / 256.0) > 1.0 ? K / (N / 256.0) : 1.0) + 2.0 * (N / 256.0) + 96 <= 512;
  const int cap = small_lists ? 512 : 1024;
  if (vec4 && P.sample_rank > 0 && workspace && workspace_bytes >= dgcn_knn_dense_workspace_bytes(B, N, 0) &&
      knn_filter_lds_bytes(C, cap) <= static_cast<size_t>(kLdsBudget)) {
    KnnParams F = P;
    const size_t pts = static_cast<size_t>(B) * N;
    F.sqnorm = static_cast<float*>(workspace);
    F.tau = reinterpret_cast<uint32_t*>(F.sqnorm + pts);
    F.redo = reinterpret_cast<int*>(F.tau + pts);
    F.sample_rank = knn_sample_rank(N, K, cap);
    const size_t flds = knn_filter_lds_bytes(C, cap);
    const size_t plds = knn_prep_lds_bytes(C);
    const int ftiles = (N + kFTM - 1) / kFTM;
    const dim3 fgrid(static_cast<unsigned>(B) * ftiles);
    // the bf16 variant needs the planes behind the head of the workspace; a caller that passes the head only
    // (dgcn_knn_dense_workspace_bytes(B, N, 0)) gets the fp32-MFMA filter kernel
    // C in {32, 64}: the planes behind the head serve the bf16 filter kernels; with the candidate lists behind the planes
    // too (the size dgcn_knn_dense_workspace_bytes reports) the 32-row kernel with global lists runs, with the planes only
    // (rounds 4 - 5's size) the 16-row kernel with LDS lists -- kept for A/B measurements, same ids
    const size_t ws_planes = knn_ws_head(pts) + knn_planes_bytes(B, N, C);
    const int kc = (knn_bf16_kc(C) && workspace_bytes >= ws_planes) ? knn_bf16_kc(C) : 0;
    const bool lists32 = kc && workspace_bytes >= ws_planes + knn_lists_bytes(pts);
    int bcap = cap;
    if (lists32) {
      bcap = kF2Cap;
      F.sample_rank = knn_sample_rank(N, K, bcap, 512);
    } else if (kc) {
      // 512 samples (unit = N / 512): the candidate count spreads half as much, so the short lists serve larger K
      const double unit = N / 512.0, r0 = K / unit;
      const bool small = K + 3.2 * unit * sqrt(r0 > 1.0 ? r0 : 1.0) + 2.0 * unit + 96 <= 512;
      bcap = small ? 512 : 1024;
      F.sample_rank = knn_sample_rank(N, K, bcap, 512);
    }
    if (F.sample_rank > 0 && (kc || plds <= static_cast<size_t>(kLdsBudget))) {
      if (kc) {
        // distance pass on the bf16 matrix pipe from pre-split point-major planes; |x_j|^2, the redo-counter reset and
        // the sample thresholds come from the planes kernel / the filter kernel itself: no prep launch
        i4v* planes = reinterpret_cast<i4v*>(static_cast<char*>(workspace) + knn_ws_head(pts));
        const int Np = (N + 15) & ~15;
        const int ppb = 256 / (C / 8);
        const unsigned pblocks = static_cast<unsigned>(B) * static_cast<unsigned>((Np + ppb - 1) / ppb);
        hipLaunchKernelGGL(knn_planes_kernel, dim3(pblocks), dim3(256), 0, s, x, sb, sc, B, C, N, Np, planes, F.sqnorm,
                           F.redo);
        F.planes = planes;
        if (lists32) {
          F.lists = reinterpret_cast<uint2*>(static_cast<char*>(workspace) + ws_planes);
          F.list_cnt = reinterpret_cast<int*>(static_cast<char*>(workspace) + ws_planes + pts * static_cast<size_t>(kF2Cap) * 8u);
          const size_t l2 = knn_filter2_lds_bytes();
          const dim3 g2(static_cast<unsigned>(B) * static_cast<unsigned>((N + kF2Rows - 1) / kF2Rows));
#define DGCN_KNN2_LAUNCH(KCV, EX)                                                                            \
  do {                                                                                                        \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(knn_filter2_kernel<KCV, EX>),                       \
                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(l2));                \
    if (e != hipSuccess) return static_cast<int>(e);                                                          \
    hipLaunchKernelGGL((knn_filter2_kernel<KCV, EX>), g2, dim3(kF2Waves * kWave), l2, s, F);                  \
  } while (0)
          if (kc == 1) { if (exclude_self) DGCN_KNN2_LAUNCH(1, true); else DGCN_KNN2_LAUNCH(1, false); }
          else { if (exclude_self) DGCN_KNN2_LAUNCH(2, true); else DGCN_KNN2_LAUNCH(2, false); }
          const size_t l3 = knn_select_lds_bytes();
          const dim3 g3(static_cast<unsigned>((pts + kSelWaves - 1) / kSelWaves));
          hipLaunchKernelGGL(knn_select_lists_kernel, g3, dim3(kSelWaves * kWave), l3, s, F);
#undef DGCN_KNN2_LAUNCH
        } else {
        const size_t blds = knn_filter_bf16_lds_bytes(bcap);
#define DGCN_KNNB_LAUNCH(CAP, KCV, NWV)                                                                      \
  do {                                                                                                        \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(knn_filter_bf16_kernel<CAP, KCV, NWV>),             \
                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(blds));              \
    if (e != hipSuccess) return static_cast<int>(e);                                                          \
    hipLaunchKernelGGL((knn_filter_bf16_kernel<CAP, KCV, NWV>), fgrid, dim3(NWV * kWave), blds, s, F);        \
  } while (0)
        // 512-entry lists (64 KB of LDS): 8-wave workgroups, two per CU, so that one's per-row select overlaps the
        // other's matrix pass; 1024-entry lists (128 KB) leave room for one workgroup per CU anyway: 16 waves
        if (bcap == 512) {
          if (kc == 1) DGCN_KNNB_LAUNCH(512, 1, kFbWaves); else DGCN_KNNB_LAUNCH(512, 2, kFbWaves);
        } else {
          if (kc == 1) DGCN_KNNB_LAUNCH(1024, 1, 16); else DGCN_KNNB_LAUNCH(1024, 2, 16);
        }
#undef DGCN_KNNB_LAUNCH
        }
      } else {
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(knn_prep_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(plds));
      if (e != hipSuccess) return static_cast<int>(e);
      hipLaunchKernelGGL(knn_prep_kernel, fgrid, dim3(kPrepThreads), plds, s, F);
      // chunk = 4*KS channels, two chunks per column block: KS sized so that C fills both
      const int ks = C > 16 ? 4 : 2;   // (KS = 8 double-buffered needs > 128 VGPRs: spills)
#define DGCN_KNNF_LAUNCH(CAP, KSV)                                                                           \
  do {                                                                                                        \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(knn_filter_kernel<CAP, KSV>),                       \
                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(flds));              \
    if (e != hipSuccess) return static_cast<int>(e);                                                          \
    hipLaunchKernelGGL((knn_filter_kernel<CAP, KSV>), fgrid, dim3(kFThreads), flds, s, F);                    \
  } while (0)
      if (cap == 512) {
        if (ks == 4) DGCN_KNNF_LAUNCH(512, 4); else DGCN_KNNF_LAUNCH(512, 2);
      } else {
        if (ks == 4) DGCN_KNNF_LAUNCH(1024, 4); else DGCN_KNNF_LAUNCH(1024, 2);
      }
#undef DGCN_KNNF_LAUNCH
      }
      // The exact pass below only redoes the rows the filter pass listed: one row per tile (a redo then costs one
      // row's distance strip, not eight), a fixed grid striding over the device-side list.
      P.redo = F.redo;
      TM = 1;
      lds = knn_lds_bytes(TM, C, Npad, Kpad);
      tiles = N;
      grid = dim3(kRedoGrid);
    }
  }
#define DGCN_KNN_LAUNCH_S(TMV, V4, SL)                                                                     \
  do {                                                                                                      \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(knn_dense_kernel<TMV, V4, SL>),                   \
                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));             \
    if (e != hipSuccess) return static_cast<int>(e);                                                        \
    hipLaunchKernelGGL((knn_dense_kernel<TMV, V4, SL>), grid, block, lds, s, P, Npad, Kpad);                \
  } while (0)
  // keys per lane of the wave-per-row select, or the cooperative select for the big clouds
#define DGCN_KNN_LAUNCH(TMV, V4)                                                                           \
  do {                                                                                                      \
    if (N <= 16 * kWave) DGCN_KNN_LAUNCH_S(TMV, V4, 16);                                                    \
    else if (N <= 32 * kWave) DGCN_KNN_LAUNCH_S(TMV, V4, 32);                                               \
    else DGCN_KNN_LAUNCH_S(TMV, V4, 0);                                                                     \
  } while (0)
  switch (TM) {
    case 8: if (vec4) DGCN_KNN_LAUNCH(8, true); else DGCN_KNN_LAUNCH(8, false); break;
    case 4: if (vec4) DGCN_KNN_LAUNCH(4, true); else DGCN_KNN_LAUNCH(4, false); break;
    case 2: if (vec4) DGCN_KNN_LAUNCH(2, true); else DGCN_KNN_LAUNCH(2, false); break;
    default: if (vec4) DGCN_KNN_LAUNCH(1, true); else DGCN_KNN_LAUNCH(1, false); break;
  }
#undef DGCN_KNN_LAUNCH
#undef DGCN_KNN_LAUNCH_S
  return launch_status();
}
