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

__global__ __launch_bounds__(kGbThreads) void max_degree_kernel(const int32_t* __restrict__ counts, int32_t n_rows,
                                                                int32_t* __restrict__ status) {
  int m = 0;
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_rows; i += stride) m = max(m, counts[i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off));
  if (lane_id() == 0) atomicMax(&status[1], m);
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

__global__ __launch_bounds__(kGbThreads) void edge_flag_kernel(const int64_t* __restrict__ src,
                                                               const int64_t* __restrict__ dst, int64_t n_edges,
                                                               const int32_t* __restrict__ nflag,
                                                               int32_t* __restrict__ eflag) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e <= n_edges; e += stride) {
    eflag[e] = (e < n_edges && nflag[src[e]] && nflag[dst[e]]) ? 1 : 0;
  }
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
                                    const int32_t* __restrict__ epos, int64_t n_edges, int64_t* __restrict__ counts) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    counts[0] = npos[n_nodes];
    counts[1] = epos[n_edges];
  }
}

struct SubWs {
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
                                    int32_t n_other, int32_t* rowptr, int32_t* col, int32_t* eperm, int32_t* erow,
                                    int32_t* status, void* workspace, size_t workspace_bytes, void* stream) {
  if (!rowptr || !status || !workspace) return DGCN_E_NULL;
  if (n_edges < 0 || n_edges > 0x7fffffffLL || n_rows < 0 || n_other < 0) return DGCN_E_SHAPE;
  if (n_edges > 0 && (!key || !other || !col || !eperm)) return DGCN_E_NULL;
  const CsrWs w = csr_layout(workspace, n_edges, n_rows);
  if (workspace_bytes < w.total) return DGCN_E_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e = hipMemsetAsync(w.counts, 0, (static_cast<size_t>(n_rows) + 1) * 4, s);
  if (e != hipSuccess) return static_cast<int>(e);
  e = hipMemsetAsync(status, 0, 4 * sizeof(int32_t), s);
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
    hipLaunchKernelGGL(max_degree_kernel, dim3(gb_grid(n_rows)), dim3(kGbThreads), 0, s, w.counts, n_rows, status);
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
  hipLaunchKernelGGL(edge_flag_kernel, dim3(gb_grid(n_edges + 1)), dim3(kGbThreads), 0, s, src, dst, n_edges, w.nflag,
                     w.eflag);
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
  hipLaunchKernelGGL(write_counts_kernel, dim3(1), dim3(64), 0, s, w.npos, n_nodes, w.epos, n_edges, counts);
  return launch_status();
}
