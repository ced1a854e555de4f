// Device-side graph structure building for gfx950: COO -> CSR / CSC and induced sub-graph extraction.
//
// The reference hands every layer a COO edge_index (2, E) int64 and lets torch_scatter rediscover the segments with
// atomics on every call; its data pipeline re-partitions the graph on the HOST every epoch (scipy CSR slicing of the
// adjacency per cluster, a python dict lookup per edge for the edge ids: utils/data_util.py:43-61,
// examples/ogb/ogbn_proteins/dataset.py:87-151, examples/ogb/ogbn_products/main.py:120-124).  Once the aggregation
// runs at HBM speed that preparation is the epoch bottleneck, so it lives on the device:
//
//   dgcn_graph_csr_build      stable counting order of the edges by a row key (destination for the forward walk, source
//                             for the backward walk): histogram -> exclusive scan (rowptr) -> LSD radix sort of 32-bit
//                             keys limited to the significant bits of the row count (3 passes of 8 bits for 2.4 M rows
//                             instead of the 8 passes of an int64 sort) -> gather of the other endpoint.  Range check,
//                             maximum degree and "already sorted" come back in a small device status block: the host
//                             reads it once per graph instead of synchronising after min / max / all / bincount.
//   dgcn_graph_work_list      the hub-row work list of the aggregation kernels (rows longer than 2 * chunk edges cut into
//                             chunk-edge items with partial-result slots): its three totals are counted inside
//                             dgcn_graph_csr_build and travel in the same status block, so sizing the arrays costs no
//                             extra host read; the list itself is two scans and one fill kernel.
//   dgcn_graph_coalesce       sorted, duplicate-free (row, col) pairs of an edge list, optionally of both directions
//                             (PyG to_undirected = examples/ogb/ogbn_arxiv/main.py:72-75): one radix sort of compact
//                             row * 2^b + col keys (2 * ceil(log2 n) bits), first-of-run flags, scan, compaction.
//   dgcn_subgraph_extract     nodes with parts == cluster (ascending), edges with both endpoints inside, relabelled, in
//                             the original edge order, plus the kept edge ids (to slice edge_attr): flag -> scan ->
//                             compact, counts returned on the device.
//
// Scan and radix sort are rocPRIM device primitives (header templates compiled into this library, launched on the
// caller's stream with caller-provided temporary storage); the flag / histogram / gather / compaction kernels are here.
// Integer work, HBM-bound: ~E * (8 + 8) bytes in, 4 * E * (passes * 4 + 3) bytes of sort traffic, E * 12 out.

#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "dgcn_common.h"

namespace dgcn {
namespace {

constexpr int kGbThreads = 256;

inline int gb_grid(int64_t n) {
  int64_t b = (n + kGbThreads - 1) / kGbThreads;
  if (b > 65536) b = 65536;
  if (b < 1) b = 1;
  return static_cast<int>(b);
}

inline size_t align_up(size_t v) { return (v + 255) & ~static_cast<size_t>(255); }

__global__ __launch_bounds__(kGbThreads) void csr_prep_kernel(const int64_t* __restrict__ key,
                                                              const int64_t* __restrict__ other, int64_t n_edges,
                                                              int32_t n_rows, int32_t n_other,
                                                              int32_t* __restrict__ key32, int32_t* __restrict__ val,
                                                              int32_t* __restrict__ counts,
                                                              int32_t* __restrict__ status) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  bool bad = false, unsorted = false;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < n_edges; e += stride) {
    const int64_t k = key[e], o = other[e];
    const bool ok = k >= 0 && k < n_rows && o >= 0 && o < n_other;
    bad = bad || !ok;
    const int32_t kk = ok ? static_cast<int32_t>(k) : 0;
    key32[e] = kk;
    val[e] = static_cast<int32_t>(e);
    if (ok) atomicAdd(&counts[kk], 1);
    if (e > 0 && key[e - 1] > k) unsorted = true;
  }
  if (bad) atomicOr(&status[0], 1);       // rare
  if (unsorted) atomicOr(&status[2], 1);
}

__global__ __launch_bounds__(kGbThreads) void csr_gather_kernel(const int64_t* __restrict__ other,
                                                                const int32_t* __restrict__ eperm, int64_t n_edges,
                                                                int32_t* __restrict__ col) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; p < n_edges; p += stride) {
    col[p] = static_cast<int32_t>(other[eperm[p]]);
  }
}

// status[1] = max degree; status[3..5] = items / partial slots / split rows of the hub work list for `chunk`
// (integer atomics: order-independent)
__global__ __launch_bounds__(kGbThreads) void max_degree_kernel(const int32_t* __restrict__ counts, int32_t n_rows,
                                                                int32_t chunk, int32_t* __restrict__ status) {
  int m = 0, items = 0, slots = 0, splits = 0;
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_rows; i += stride) {
    const int d = counts[i];
    m = max(m, d);
    if (chunk > 0 && d > 2 * chunk) {
      const int nc = (d + chunk - 1) / chunk;
      items += nc; slots += nc; splits += 1;
    } else {
      items += 1;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    m = max(m, __shfl_xor(m, off));
    items += __shfl_xor(items, off);
    slots += __shfl_xor(slots, off);
    splits += __shfl_xor(splits, off);
  }
  if (lane_id() == 0) {
    atomicMax(&status[1], m);
    if (chunk > 0) {
      atomicAdd(&status[3], items);
      if (slots) atomicAdd(&status[4], slots);
      if (splits) atomicAdd(&status[5], splits);
    }
  }
}

// ---- hub work list ---------------------------------------------------------------------------------------------------
// per row: a[r] = items of the row (1, or ceil(deg / chunk) when deg > 2 chunk), b[r] = partial slots (0 or a[r]),
// c[r] = 1 for a split row; slot [n_rows] = 0 so that the exclusive scans end with the totals
__global__ __launch_bounds__(kGbThreads) void work_count_kernel(const int32_t* __restrict__ rowptr, int32_t n_rows,
                                                                int32_t chunk, int32_t* __restrict__ a,
                                                                int32_t* __restrict__ b, int32_t* __restrict__ c) {
  const int stride = gridDim.x * blockDim.x;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r <= n_rows; r += stride) {
    int na = 0, nb = 0, nc = 0;
    if (r < n_rows) {
      const int d = rowptr[r + 1] - rowptr[r];
      const bool split = d > 2 * chunk;
      na = split ? (d + chunk - 1) / chunk : 1;
      nb = split ? na : 0;
      nc = split ? 1 : 0;
    }
    a[r] = na; b[r] = nb; c[r] = nc;
  }
}

__global__ __launch_bounds__(kGbThreads) void work_fill_kernel(const int32_t* __restrict__ rowptr, int32_t n_rows,
                                                               int32_t chunk, const int32_t* __restrict__ first,
                                                               const int32_t* __restrict__ slot0,
                                                               const int32_t* __restrict__ sidx,
                                                               int32_t* __restrict__ work_row,
                                                               int32_t* __restrict__ work_beg,
                                                               int32_t* __restrict__ work_end,
                                                               int32_t* __restrict__ work_slot,
                                                               int32_t* __restrict__ split_item) {
  const int stride = gridDim.x * blockDim.x;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += stride) {
    const int b0 = rowptr[r], e0 = rowptr[r + 1];
    const int f = first[r];
    const int n = first[r + 1] - f;
    if (n == 1) {
      work_row[f] = r; work_beg[f] = b0; work_end[f] = e0; work_slot[f] = -1;
    } else {
      split_item[sidx[r]] = f;
      const int s0 = slot0[r];
      for (int k = 0; k < n; ++k) {
        const int bb = b0 + k * chunk;
        work_row[f + k] = r;
        work_beg[f + k] = bb;
        work_end[f + k] = min(bb + chunk, e0);
        work_slot[f + k] = s0 + k;
      }
    }
  }
}

struct CsrWs {
  int32_t* counts;   // [n_rows + 1]
  int32_t* key32;    // [E]
  int32_t* val;      // [E]
  int32_t* keyout;   // [E]
  void* temp;
  size_t temp_bytes;
  size_t total;
};

inline int key_bits(int32_t n_rows) {
  int bits = 1;
  while (bits < 31 && (static_cast<int64_t>(1) << bits) < n_rows) ++bits;
  return bits;
}

inline CsrWs csr_layout(void* base, int64_t n_edges, int32_t n_rows) {
  CsrWs w;
  size_t sort_bytes = 0, scan_bytes = 0;
  const size_t ne = static_cast<size_t>(n_edges > 0 ? n_edges : 1);
  (void)rocprim::radix_sort_pairs(nullptr, sort_bytes, static_cast<int32_t*>(nullptr), static_cast<int32_t*>(nullptr),
                                  static_cast<int32_t*>(nullptr), static_cast<int32_t*>(nullptr), ne, 0,
                                  key_bits(n_rows), static_cast<hipStream_t>(nullptr));
  (void)rocprim::exclusive_scan(nullptr, scan_bytes, static_cast<int32_t*>(nullptr), static_cast<int32_t*>(nullptr), 0,
                                static_cast<size_t>(n_rows) + 1, rocprim::plus<int32_t>(),
                                static_cast<hipStream_t>(nullptr));
  w.temp_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
  char* p = static_cast<char*>(base);
  size_t off = 0;
  w.counts = reinterpret_cast<int32_t*>(p + off); off += align_up((static_cast<size_t>(n_rows) + 1) * 4);
  w.key32 = reinterpret_cast<int32_t*>(p + off); off += align_up(ne * 4);
  w.val = reinterpret_cast<int32_t*>(p + off); off += align_up(ne * 4);
  w.keyout = reinterpret_cast<int32_t*>(p + off); off += align_up(ne * 4);
  w.temp = p + off; off += align_up(w.temp_bytes);
  w.total = off;
  return w;
}

// ---- induced sub-graph -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kGbThreads) void node_flag_kernel(const int64_t* __restrict__ parts, int32_t n_nodes,
                                                               int64_t cluster, int32_t* __restrict__ flag) {
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i <= n_nodes; i += stride) {
    flag[i] = (i < n_nodes && parts[i] == cluster) ? 1 : 0;      // flag[n_nodes] = 0: the scan's last slot = the count
  }
}

// an endpoint outside [0, n_nodes) is never kept and raises the error flag (counts[2]) instead of reading out of bounds
__global__ __launch_bounds__(kGbThreads) void edge_flag_kernel(const int64_t* __restrict__ src,
                                                               const int64_t* __restrict__ dst, int64_t n_edges,
                                                               int32_t n_nodes, const int32_t* __restrict__ nflag,
                                                               int32_t* __restrict__ eflag,
                                                               int32_t* __restrict__ bad_flag) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  bool bad = false;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e <= n_edges; e += stride) {
    int keep = 0;
    if (e < n_edges) {
      const int64_t a = src[e], b = dst[e];
      const bool ok = a >= 0 && a < n_nodes && b >= 0 && b < n_nodes;
      bad = bad || !ok;
      keep = (ok && nflag[a] && nflag[b]) ? 1 : 0;
    }
    eflag[e] = keep;
  }
  if (bad) atomicOr(bad_flag, 1);
}

__global__ __launch_bounds__(kGbThreads) void node_compact_kernel(const int32_t* __restrict__ nflag,
                                                                  const int32_t* __restrict__ npos, int32_t n_nodes,
                                                                  int64_t* __restrict__ node_ids) {
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_nodes; i += stride) {
    if (nflag[i]) node_ids[npos[i]] = i;
  }
}

__global__ __launch_bounds__(kGbThreads) void edge_compact_kernel(const int64_t* __restrict__ src,
                                                                  const int64_t* __restrict__ dst, int64_t n_edges,
                                                                  const int32_t* __restrict__ eflag,
                                                                  const int32_t* __restrict__ epos,
                                                                  const int32_t* __restrict__ npos,
                                                                  int64_t* __restrict__ sub_src,
                                                                  int64_t* __restrict__ sub_dst,
                                                                  int64_t* __restrict__ eids) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < n_edges; e += stride) {
    if (eflag[e]) {
      const int32_t q = epos[e];
      sub_src[q] = npos[src[e]];
      sub_dst[q] = npos[dst[e]];
      eids[q] = e;
    }
  }
}

__global__ void write_counts_kernel(const int32_t* __restrict__ npos, int32_t n_nodes,
                                    const int32_t* __restrict__ epos, int64_t n_edges, const int32_t* __restrict__ bad,
                                    int64_t* __restrict__ counts) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    counts[0] = npos[n_nodes];
    counts[1] = epos[n_edges];
    counts[2] = *bad;
  }
}

// ---- coalesce / to_undirected ------------------------------------------------------------------------------------
// key = row << cb | col (cb = bits of n_nodes): position p < E is edge p, position p >= E is edge p - E reversed
__global__ __launch_bounds__(kGbThreads) void pair_key_kernel(const int64_t* __restrict__ src,
                                                              const int64_t* __restrict__ dst, int64_t n_edges,
                                                              int64_t n_keys, int32_t n_nodes, int cb,
                                                              uint64_t* __restrict__ keys, int32_t* __restrict__ bad_flag) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  bool bad = false;
  for (int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; p < n_keys; p += stride) {
    const bool rev = p >= n_edges;
    const int64_t e = rev ? p - n_edges : p;
    int64_t a = rev ? dst[e] : src[e], b = rev ? src[e] : dst[e];
    const bool ok = a >= 0 && a < n_nodes && b >= 0 && b < n_nodes;
    bad = bad || !ok;
    if (!ok) { a = 0; b = 0; }
    keys[p] = (static_cast<uint64_t>(a) << cb) | static_cast<uint64_t>(b);
  }
  if (bad) atomicOr(bad_flag, 1);
}

__global__ __launch_bounds__(kGbThreads) void first_of_run_kernel(const uint64_t* __restrict__ keys, int64_t n_keys,
                                                                  int32_t* __restrict__ flag) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; p <= n_keys; p += stride) {
    flag[p] = (p < n_keys && (p == 0 || keys[p] != keys[p - 1])) ? 1 : 0;
  }
}

__global__ __launch_bounds__(kGbThreads) void pair_compact_kernel(const uint64_t* __restrict__ keys, int64_t n_keys,
                                                                  const int32_t* __restrict__ flag,
                                                                  const int32_t* __restrict__ pos, int cb,
                                                                  int64_t out_stride, int64_t* __restrict__ out,
                                                                  const int32_t* __restrict__ bad_flag,
                                                                  int64_t* __restrict__ counts) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const uint64_t cmask = (static_cast<uint64_t>(1) << cb) - 1;
  for (int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; p < n_keys; p += stride) {
    if (flag[p]) {
      const int32_t q = pos[p];
      out[q] = static_cast<int64_t>(keys[p] >> cb);
      out[out_stride + q] = static_cast<int64_t>(keys[p] & cmask);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    counts[0] = pos[n_keys];
    counts[1] = *bad_flag;
  }
}

struct SubWs {
  void* bad;       // one int32: an endpoint was out of range
  int32_t* nflag;  // [N + 1]
  int32_t* npos;   // [N + 1]
  int32_t* eflag;  // [E + 1]
  int32_t* epos;   // [E + 1]
  void* temp;
  size_t temp_bytes;
  size_t total;
};

inline SubWs sub_layout(void* base, int64_t n_edges, int32_t n_nodes) {
  SubWs w;
  size_t a = 0, b = 0;
  (void)rocprim::exclusive_scan(nullptr, a, static_cast<int32_t*>(nullptr), static_cast<int32_t*>(nullptr), 0,
                                static_cast<size_t>(n_nodes) + 1, rocprim::plus<int32_t>(),
                                static_cast<hipStream_t>(nullptr));
  (void)rocprim::exclusive_scan(nullptr, b, static_cast<int32_t*>(nullptr), static_cast<int32_t*>(nullptr), 0,
                                static_cast<size_t>(n_edges) + 1, rocprim::plus<int32_t>(),
                                static_cast<hipStream_t>(nullptr));
  w.temp_bytes = a > b ? a : b;
  char* p = static_cast<char*>(base);
  size_t off = 0;
  w.bad = p + off; off += align_up(4);
  w.nflag = reinterpret_cast<int32_t*>(p + off); off += align_up((static_cast<size_t>(n_nodes) + 1) * 4);
  w.npos = reinterpret_cast<int32_t*>(p + off); off += align_up((static_cast<size_t>(n_nodes) + 1) * 4);
  w.eflag = reinterpret_cast<int32_t*>(p + off); off += align_up((static_cast<size_t>(n_edges) + 1) * 4);
  w.epos = reinterpret_cast<int32_t*>(p + off); off += align_up((static_cast<size_t>(n_edges) + 1) * 4);
  w.temp = p + off; off += align_up(w.temp_bytes);
  w.total = off;
  return w;
}

}  // namespace
}  // namespace dgcn

using namespace dgcn;

extern "C" size_t dgcn_graph_csr_workspace_bytes(int64_t n_edges, int32_t n_rows) {
  if (n_edges < 0 || n_rows < 0) return 0;
  return csr_layout(nullptr, n_edges, n_rows).total;
}

extern "C" int dgcn_graph_csr_build(const int64_t* key, const int64_t* other, int64_t n_edges, int32_t n_rows,
                                    int32_t n_other, int32_t hub_chunk, int32_t* rowptr, int32_t* col, int32_t* eperm,
                                    int32_t* erow, int32_t* status, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  if (!rowptr || !status || !workspace) return DGCN_E_NULL;
  if (n_edges < 0 || n_edges > 0x7fffffffLL || n_rows < 0 || n_other < 0) return DGCN_E_SHAPE;
  if (n_edges > 0 && (!key || !other || !col || !eperm)) return DGCN_E_NULL;
  const CsrWs w = csr_layout(workspace, n_edges, n_rows);
  if (workspace_bytes < w.total) return DGCN_E_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e = hipMemsetAsync(w.counts, 0, (static_cast<size_t>(n_rows) + 1) * 4, s);
  if (e != hipSuccess) return static_cast<int>(e);
  e = hipMemsetAsync(status, 0, 8 * sizeof(int32_t), s);
  if (e != hipSuccess) return static_cast<int>(e);
  if (n_edges > 0) {
    hipLaunchKernelGGL(csr_prep_kernel, dim3(gb_grid(n_edges)), dim3(kGbThreads), 0, s, key, other, n_edges, n_rows,
                       n_other, w.key32, w.val, w.counts, status);
  }
  size_t tb = w.temp_bytes;
  e = rocprim::exclusive_scan(w.temp, tb, w.counts, rowptr, 0, static_cast<size_t>(n_rows) + 1,
                              rocprim::plus<int32_t>(), s);
  if (e != hipSuccess) return static_cast<int>(e);
  if (n_rows > 0) {
    hipLaunchKernelGGL(max_degree_kernel, dim3(gb_grid(n_rows)), dim3(kGbThreads), 0, s, w.counts, n_rows, hub_chunk,
                       status);
  }
  if (n_edges > 0) {
    tb = w.temp_bytes;
    int32_t* kout = erow ? erow : w.keyout;
    e = rocprim::radix_sort_pairs(w.temp, tb, w.key32, kout, w.val, eperm, static_cast<size_t>(n_edges), 0,
                                  key_bits(n_rows), s);
    if (e != hipSuccess) return static_cast<int>(e);
    hipLaunchKernelGGL(csr_gather_kernel, dim3(gb_grid(n_edges)), dim3(kGbThreads), 0, s, other, eperm, n_edges, col);
  }
  return launch_status();
}

extern "C" size_t dgcn_subgraph_workspace_bytes(int64_t n_edges, int32_t n_nodes) {
  if (n_edges < 0 || n_nodes < 0) return 0;
  return sub_layout(nullptr, n_edges, n_nodes).total;
}

extern "C" int dgcn_subgraph_extract(const int64_t* src, const int64_t* dst, int64_t n_edges, const int64_t* parts,
                                     int32_t n_nodes, int64_t cluster, int64_t* node_ids, int64_t* sub_src,
                                     int64_t* sub_dst, int64_t* edge_ids, int64_t* counts, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  if (!parts || !node_ids || !counts || !workspace) return DGCN_E_NULL;
  if (n_edges < 0 || n_edges > 0x7fffffffLL || n_nodes < 0) return DGCN_E_SHAPE;
  if (n_edges > 0 && (!src || !dst || !sub_src || !sub_dst || !edge_ids)) return DGCN_E_NULL;
  const SubWs w = sub_layout(workspace, n_edges, n_nodes);
  if (workspace_bytes < w.total) return DGCN_E_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(node_flag_kernel, dim3(gb_grid(n_nodes + 1)), dim3(kGbThreads), 0, s, parts, n_nodes, cluster,
                     w.nflag);
  size_t tb = w.temp_bytes;
  hipError_t e = rocprim::exclusive_scan(w.temp, tb, w.nflag, w.npos, 0, static_cast<size_t>(n_nodes) + 1,
                                         rocprim::plus<int32_t>(), s);
  if (e != hipSuccess) return static_cast<int>(e);
  int32_t* bad_flag = static_cast<int32_t*>(w.bad);
  hipError_t me = hipMemsetAsync(bad_flag, 0, sizeof(int32_t), s);
  if (me != hipSuccess) return static_cast<int>(me);
  hipLaunchKernelGGL(edge_flag_kernel, dim3(gb_grid(n_edges + 1)), dim3(kGbThreads), 0, s, src, dst, n_edges, n_nodes,
                     w.nflag, w.eflag, bad_flag);
  tb = w.temp_bytes;
  e = rocprim::exclusive_scan(w.temp, tb, w.eflag, w.epos, 0, static_cast<size_t>(n_edges) + 1,
                              rocprim::plus<int32_t>(), s);
  if (e != hipSuccess) return static_cast<int>(e);
  hipLaunchKernelGGL(node_compact_kernel, dim3(gb_grid(n_nodes)), dim3(kGbThreads), 0, s, w.nflag, w.npos, n_nodes,
                     node_ids);
  if (n_edges > 0) {
    hipLaunchKernelGGL(edge_compact_kernel, dim3(gb_grid(n_edges)), dim3(kGbThreads), 0, s, src, dst, n_edges, w.eflag,
                       w.epos, w.npos, sub_src, sub_dst, edge_ids);
  }
  hipLaunchKernelGGL(write_counts_kernel, dim3(1), dim3(64), 0, s, w.npos, n_nodes, w.epos, n_edges, bad_flag, counts);
  return launch_status();
}

// ---- hub work list -----------------------------------------------------------------------------------------------
namespace dgcn {
namespace {
struct WorkWs {
  int32_t *a, *b, *c, *fa, *fb, *fc;
  void* temp;
  size_t temp_bytes, total;
};
inline WorkWs work_layout(void* base, int32_t n_rows) {
  WorkWs w;
  size_t tb = 0;
  (void)rocprim::exclusive_scan(nullptr, tb, static_cast<int32_t*>(nullptr), static_cast<int32_t*>(nullptr), 0,
                                static_cast<size_t>(n_rows) + 1, rocprim::plus<int32_t>(),
                                static_cast<hipStream_t>(nullptr));
  w.temp_bytes = tb;
  char* p = static_cast<char*>(base);
  size_t off = 0;
  const size_t each = align_up((static_cast<size_t>(n_rows) + 1) * 4);
  int32_t** slots[6] = {&w.a, &w.b, &w.c, &w.fa, &w.fb, &w.fc};
  for (auto sp : slots) { *sp = reinterpret_cast<int32_t*>(p + off); off += each; }
  w.temp = p + off; off += align_up(tb);
  w.total = off;
  return w;
}
}  // namespace
}  // namespace dgcn

extern "C" size_t dgcn_graph_work_list_workspace_bytes(int32_t n_rows) {
  if (n_rows < 0) return 0;
  return work_layout(nullptr, n_rows).total;
}

extern "C" int dgcn_graph_work_list(const int32_t* rowptr, int32_t n_rows, int32_t hub_chunk, int32_t* work_row,
                                    int32_t* work_beg, int32_t* work_end, int32_t* work_slot, int32_t* split_item,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  if (!rowptr || !work_row || !work_beg || !work_end || !work_slot || !split_item || !workspace) return DGCN_E_NULL;
  if (n_rows <= 0 || hub_chunk <= 0) return DGCN_E_SHAPE;
  const WorkWs w = work_layout(workspace, n_rows);
  if (workspace_bytes < w.total) return DGCN_E_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(work_count_kernel, dim3(gb_grid(n_rows + 1)), dim3(kGbThreads), 0, s, rowptr, n_rows, hub_chunk,
                     w.a, w.b, w.c);
  int32_t* in[3] = {w.a, w.b, w.c};
  int32_t* out[3] = {w.fa, w.fb, w.fc};
  for (int i = 0; i < 3; ++i) {
    size_t tb = w.temp_bytes;
    hipError_t e = rocprim::exclusive_scan(w.temp, tb, in[i], out[i], 0, static_cast<size_t>(n_rows) + 1,
                                           rocprim::plus<int32_t>(), s);
    if (e != hipSuccess) return static_cast<int>(e);
  }
  hipLaunchKernelGGL(work_fill_kernel, dim3(gb_grid(n_rows)), dim3(kGbThreads), 0, s, rowptr, n_rows, hub_chunk, w.fa,
                     w.fb, w.fc, work_row, work_beg, work_end, work_slot, split_item);
  return launch_status();
}

// ---- coalesce / to_undirected --------------------------------------------------------------------------------------
namespace dgcn {
namespace {
struct CoWs {
  uint64_t *keys, *sorted;
  int32_t *flag, *pos, *bad;
  void* temp;
  size_t temp_bytes, total;
};
inline CoWs co_layout(void* base, int64_t n_keys, int32_t n_nodes) {
  CoWs w;
  const size_t nk = static_cast<size_t>(n_keys > 0 ? n_keys : 1);
  const int cb = key_bits(n_nodes);
  size_t a = 0, b = 0;
  (void)rocprim::radix_sort_keys(nullptr, a, static_cast<uint64_t*>(nullptr), static_cast<uint64_t*>(nullptr), nk, 0,
                                 2 * cb, static_cast<hipStream_t>(nullptr));
  (void)rocprim::exclusive_scan(nullptr, b, static_cast<int32_t*>(nullptr), static_cast<int32_t*>(nullptr), 0, nk + 1,
                                rocprim::plus<int32_t>(), static_cast<hipStream_t>(nullptr));
  w.temp_bytes = a > b ? a : b;
  char* p = static_cast<char*>(base);
  size_t off = 0;
  w.bad = reinterpret_cast<int32_t*>(p + off); off += align_up(4);
  w.keys = reinterpret_cast<uint64_t*>(p + off); off += align_up(nk * 8);
  w.sorted = reinterpret_cast<uint64_t*>(p + off); off += align_up(nk * 8);
  w.flag = reinterpret_cast<int32_t*>(p + off); off += align_up((nk + 1) * 4);
  w.pos = reinterpret_cast<int32_t*>(p + off); off += align_up((nk + 1) * 4);
  w.temp = p + off; off += align_up(w.temp_bytes);
  w.total = off;
  return w;
}
}  // namespace
}  // namespace dgcn

extern "C" size_t dgcn_graph_coalesce_workspace_bytes(int64_t n_edges, int32_t n_nodes, int32_t both_directions) {
  if (n_edges < 0 || n_nodes < 0) return 0;
  return co_layout(nullptr, n_edges * (both_directions ? 2 : 1), n_nodes).total;
}

extern "C" int dgcn_graph_coalesce(const int64_t* src, const int64_t* dst, int64_t n_edges, int32_t n_nodes,
                                   int32_t both_directions, int64_t* out, int64_t out_stride, int64_t* counts,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  if (!out || !counts || !workspace) return DGCN_E_NULL;
  const int64_t n_keys = n_edges * (both_directions ? 2 : 1);
  if (n_edges < 0 || n_keys > 0x7fffffffLL || n_nodes <= 0 || out_stride < n_keys) return DGCN_E_SHAPE;
  if (n_edges > 0 && (!src || !dst)) return DGCN_E_NULL;
  const CoWs w = co_layout(workspace, n_keys, n_nodes);
  if (workspace_bytes < w.total) return DGCN_E_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int cb = key_bits(n_nodes);
  hipError_t e = hipMemsetAsync(w.bad, 0, sizeof(int32_t), s);
  if (e != hipSuccess) return static_cast<int>(e);
  if (n_keys > 0) {
    hipLaunchKernelGGL(pair_key_kernel, dim3(gb_grid(n_keys)), dim3(kGbThreads), 0, s, src, dst, n_edges, n_keys,
                       n_nodes, cb, w.keys, w.bad);
    size_t tb = w.temp_bytes;
    e = rocprim::radix_sort_keys(w.temp, tb, w.keys, w.sorted, static_cast<size_t>(n_keys), 0, 2 * cb, s);
    if (e != hipSuccess) return static_cast<int>(e);
  }
  hipLaunchKernelGGL(first_of_run_kernel, dim3(gb_grid(n_keys + 1)), dim3(kGbThreads), 0, s, w.sorted, n_keys, w.flag);
  size_t tb = w.temp_bytes;
  e = rocprim::exclusive_scan(w.temp, tb, w.flag, w.pos, 0, static_cast<size_t>(n_keys) + 1, rocprim::plus<int32_t>(), s);
  if (e != hipSuccess) return static_cast<int>(e);
  hipLaunchKernelGGL(pair_compact_kernel, dim3(gb_grid(n_keys > 0 ? n_keys : 1)), dim3(kGbThreads), 0, s, w.sorted,
                     n_keys, w.flag, w.pos, cb, out_stride, out, w.bad, counts);
  return launch_status();
}
