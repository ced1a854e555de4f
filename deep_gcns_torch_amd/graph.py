"""Device-resident graph structure consumed by libdgcn (struct dgcn_graph, include/dgcn.h).

The reference hands every layer the same COO ``edge_index`` (2,E) int64 tensor
(examples/ogb/ogbn_arxiv/main.py:72-75 builds it once; gcn_lib/sparse/torch_vertex.py:68
passes it to ``propagate``) and lets torch_scatter rediscover the segments with atomics on
every call.  Here the segment structure is built ONCE per distinct ``edge_index``:

  * CSR keyed by destination (``edge_index[1]``) -> forward walk, one wave per row;
  * CSC keyed by source      (``edge_index[0]``) -> backward walk, deterministic, no atomics;
  * both are STABLE sorts, so edges of a row keep their original order (first-max semantics
    of scatter_max, duplicate edges counted twice, self-loops kept);
  * rows longer than ``2*HUB_CHUNK`` edges are split into ``HUB_CHUNK``-edge work items whose
    partial results are merged by a second tiny kernel (degree skew: SURVEY.md §7 hard-part 4).

Index arrays are int32 (E < 2^31).  Built with device-side torch sort/scan (plumbing, run
once); the per-layer hot path never touches COO again.
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import Optional

import torch

from . import _lib

HUB_CHUNK = 256                  # rows longer than 2 * chunk are cut into chunk-edge work items (big graphs)
HUB_CHUNK_SMALL = 48             # (64 until the end of round 6; ogbn-proteins cluster, per coupling function: encoder walk
                                 # 108.7 -> 102.0 us, max backward 83.3 -> 78.3, the two merges 15.3 -> 19.8; 32: 98.6 / 80.5 /
                                 # 29.9; 128: 136 / 100 / 11.4 -- the tail of the persistent walk against the merge's pieces)
                                 # graphs of up to SMALL_GRAPH_EDGES edges: their few thousand rows have to fill the chip's
SMALL_GRAPH_EDGES = 8 << 20      # 8192 wave slots AND balance them, which 64-edge items do (an ogbn-proteins cluster:
                                 # 13 k rows, degrees 1 .. 3000) and 256-edge ones do not


def default_hub_chunk(n_edges: int) -> int:
    return HUB_CHUNK_SMALL if n_edges <= SMALL_GRAPH_EDGES else HUB_CHUNK


def _work_list(rowptr: torch.Tensor, chunk: int):
    """Split rows with more than 2*chunk edges.  Returns (n_work, n_slots, row, beg, end, slot, split_first)
    as int32 device tensors, or None when no row needs splitting."""
    deg = rowptr[1:] - rowptr[:-1]
    if deg.numel() == 0 or int(deg.max()) <= 2 * chunk:
        return None
    n_rows = deg.numel()
    nchunk = torch.where(deg > 2 * chunk, (deg + chunk - 1) // chunk, torch.ones_like(deg)).long()
    rows = torch.arange(n_rows, device=rowptr.device)
    work_row = torch.repeat_interleave(rows, nchunk)
    first = torch.cumsum(nchunk, 0) - nchunk                      # first work item of each row
    k = torch.arange(work_row.numel(), device=rowptr.device) - first[work_row]
    split = nchunk[work_row] > 1
    beg = rowptr[work_row].long() + k * chunk
    end = torch.where(split, torch.minimum(beg + chunk, rowptr[work_row + 1].long()),
                      rowptr[work_row + 1].long())
    slot = torch.where(split, torch.cumsum(split.long(), 0) - 1, torch.full_like(k, -1))
    n_slots = int(split.sum())
    split_first = torch.nonzero(split & (k == 0)).flatten()      # first work item of every split row
    i32 = lambda t: t.to(torch.int32).contiguous()
    return work_row.numel(), n_slots, i32(work_row), i32(beg), i32(end), i32(slot), i32(split_first)


class Graph:
    """CSR-by-destination + CSC-by-source of one edge list, plus the ctypes view of it."""

    def __init__(self, src: torch.Tensor, dst: torch.Tensor, n_src: int, n_dst: int,
                 need_transpose: bool = True, hub_chunk: int = None):
        dev = src.device  # structure building is index plumbing and also runs on CPU tensors (host-logic tests)
        if hub_chunk is None:
            hub_chunk = default_hub_chunk(src.numel())
        if src.dim() != 1 or src.shape != dst.shape:
            raise ValueError("src/dst must be 1-D tensors of equal length")
        E = src.numel()
        if E >= 2 ** 31 or max(n_src, n_dst) >= 2 ** 31:
            raise ValueError("graph too large for int32 indices")
        self.device = dev
        self.n_src, self.n_dst, self.n_edges = int(n_src), int(n_dst), int(E)
        src = src.long().contiguous()
        dst = dst.long().contiguous()
        self.t_rowptr = self.t_col = self.t_eperm = None
        self.t_work = None
        self.out_deg = None
        if dev.type == "cuda":
            self._build_on_device(src, dst, need_transpose, hub_chunk)
        else:
            self._build_with_torch(src, dst, need_transpose, hub_chunk)
        self._c = self._make_struct()

    # ------------------------------------------------------------------
    def _build_on_device(self, src, dst, need_transpose, hub_chunk):
        """libdgcn's graph builder (csrc/graph_build.hip): histogram + scan + 32-bit LSD radix sort per orientation,
        validation / maximum degree / sortedness returned in a device status block that is read ONCE for both
        orientations (the torch composition below synchronises after min, max, is-sorted and every bincount)."""
        lib = _lib.load()
        dev, E = self.device, self.n_edges
        i32 = dict(device=dev, dtype=torch.int32)
        stream = _lib.current_stream_handle(dev)

        def build(key, other, n_rows, n_other, want_erow):
            rowptr = torch.empty(n_rows + 1, **i32)
            col, eperm = torch.empty(E, **i32), torch.empty(E, **i32)
            erow = torch.empty(E, **i32) if want_erow else None
            status = torch.empty(8, **i32)
            ws_bytes = lib.dgcn_graph_csr_workspace_bytes(E, n_rows)
            ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
            with _lib.device_ctx(dev):
                rc = lib.dgcn_graph_csr_build(key.data_ptr(), other.data_ptr(), E, n_rows, n_other, hub_chunk,
                                              rowptr.data_ptr(), _lib.ptr(col), _lib.ptr(eperm), _lib.ptr(erow),
                                              status.data_ptr(), ws.data_ptr(), ws_bytes, stream)
            _lib.check(rc, "dgcn_graph_csr_build")
            return rowptr, col, eperm, erow, status

        rowptr, col, eperm, erow, st = build(dst, src, self.n_dst, self.n_src, True)
        parts = [st]
        if need_transpose:
            t_rowptr, t_col, t_eperm, _, t_st = build(src, dst, self.n_src, self.n_dst, False)
            parts.append(t_st)
        flags = torch.stack(parts).tolist()                      # the one host read of this graph
        if any(f[0] for f in flags):
            raise ValueError("edge_index out of range")
        self.rowptr, self.col, self._erow = rowptr, col, erow
        self.eperm = eperm if flags[0][2] else None               # already destination-sorted: identity permutation
        self.deg = (rowptr[1:] - rowptr[:-1]).to(torch.float32)   # in-degree, float like PyG degree()
        self.work = self._work_list_on_device(rowptr, self.n_dst, hub_chunk, flags[0])
        if need_transpose:
            self.t_rowptr, self.t_col, self.t_eperm = t_rowptr, t_col, t_eperm
            self.out_deg = (t_rowptr[1:] - t_rowptr[:-1]).to(torch.float32)
            self.t_work = self._work_list_on_device(t_rowptr, self.n_src, hub_chunk, flags[1])

    def _work_list_on_device(self, rowptr, n_rows, hub_chunk, status):
        """Hub work list from csrc/graph_build.hip; its sizes arrived with the status block (no further host read)."""
        if status[1] <= 2 * hub_chunk or n_rows == 0:
            return None
        lib = _lib.load()
        dev = self.device
        n_work, n_slots, n_split = status[3], status[4], status[5]
        i32 = dict(device=dev, dtype=torch.int32)
        row, beg, end, slot = (torch.empty(n_work, **i32) for _ in range(4))
        split = torch.empty(n_split, **i32)
        ws_bytes = lib.dgcn_graph_work_list_workspace_bytes(n_rows)
        ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
        with _lib.device_ctx(dev):
            rc = lib.dgcn_graph_work_list(rowptr.data_ptr(), n_rows, hub_chunk, row.data_ptr(), beg.data_ptr(),
                                          end.data_ptr(), slot.data_ptr(), split.data_ptr(), ws.data_ptr(), ws_bytes,
                                          _lib.current_stream_handle(dev))
        _lib.check(rc, "dgcn_graph_work_list")
        return n_work, n_slots, row, beg, end, slot, split

    def _build_with_torch(self, src, dst, need_transpose, hub_chunk):
        """The same structure from torch ops (CPU tensors: host-logic and gloo tests)."""
        dev, E, n_src, n_dst = self.device, self.n_edges, self.n_src, self.n_dst
        if E:
            lo = int(torch.minimum(src.min(), dst.min()))
            if lo < 0 or int(src.max()) >= n_src or int(dst.max()) >= n_dst:
                raise ValueError("edge_index out of range")

        # --- CSR by destination (stable) ---
        if E == 0 or bool((dst[1:] >= dst[:-1]).all()):
            perm = None
            col = src
        else:
            perm = torch.sort(dst, stable=True).indices
            col = src[perm]
        counts = torch.bincount(dst, minlength=n_dst) if E else torch.zeros(n_dst, dtype=torch.long, device=dev)
        rowptr = torch.zeros(n_dst + 1, dtype=torch.long, device=dev)
        torch.cumsum(counts, 0, out=rowptr[1:])
        self.rowptr = rowptr.to(torch.int32)
        self.col = col.to(torch.int32).contiguous()
        self.eperm = None if perm is None else perm.to(torch.int32).contiguous()
        self.deg = counts.to(torch.float32)                      # in-degree, float like PyG degree()
        self.work = _work_list(self.rowptr, hub_chunk)

        # --- CSC by source (stable) ---
        if need_transpose:
            tperm = torch.sort(src, stable=True).indices if E else torch.zeros(0, dtype=torch.long, device=dev)
            tcounts = torch.bincount(src, minlength=n_src) if E else torch.zeros(n_src, dtype=torch.long, device=dev)
            t_rowptr = torch.zeros(n_src + 1, dtype=torch.long, device=dev)
            torch.cumsum(tcounts, 0, out=t_rowptr[1:])
            self.t_rowptr = t_rowptr.to(torch.int32)
            self.t_col = dst[tperm].to(torch.int32).contiguous()
            self.t_eperm = tperm.to(torch.int32).contiguous()
            self.out_deg = tcounts.to(torch.float32)
            self.t_work = _work_list(self.t_rowptr, hub_chunk)

    @classmethod
    def from_edge_index(cls, edge_index: torch.Tensor, num_nodes: int, **kw) -> "Graph":
        if edge_index.dim() != 2 or edge_index.size(0) != 2:
            raise ValueError("edge_index must have shape (2, E)")
        return cls(edge_index[0], edge_index[1], num_nodes, num_nodes, **kw)

    def _make_struct(self) -> _lib.DgcnGraph:
        g = _lib.DgcnGraph()
        g.n_dst, g.n_src, g.n_edges = self.n_dst, self.n_src, self.n_edges
        p = _lib.ptr
        g.rowptr, g.col, g.eperm = p(self.rowptr), p(self.col), p(self.eperm)
        g.t_rowptr, g.t_col, g.t_eperm = p(self.t_rowptr), p(self.t_col), p(self.t_eperm)
        if self.work is not None:
            g.n_work, g.n_slots = self.work[0], self.work[1]
            g.work_row, g.work_beg, g.work_end, g.work_slot = (p(t) for t in self.work[2:6])
            g.n_split, g.split_item = self.work[6].numel(), p(self.work[6])
        if self.t_work is not None:
            g.t_n_work, g.t_n_slots = self.t_work[0], self.t_work[1]
            g.t_work_row, g.t_work_beg, g.t_work_end, g.t_work_slot = (p(t) for t in self.t_work[2:6])
            g.t_n_split, g.t_split_item = self.t_work[6].numel(), p(self.t_work[6])
        return g

    @property
    def erow(self) -> torch.Tensor:
        """Destination row of every CSR position (int32 [E]) = the sorted ``edge_index[1]``; built on first use
        (the fused edge-GEMM kernel cuts its work items by edge count and reads the row of each edge from here)."""
        t = getattr(self, "_erow", None)
        if t is None:
            counts = (self.rowptr[1:] - self.rowptr[:-1]).long()
            rows = torch.arange(self.n_dst, device=self.device, dtype=torch.int32)
            t = self._erow = torch.repeat_interleave(rows, counts).contiguous()
        return t

    @property
    def t_cpos(self) -> torch.Tensor:
        """CSR position of every CSC position (int32 [E]): the index the source-keyed walk uses into per-edge arrays kept
        in destination order (the arg-max bit masks of the max backward).  Built on first use."""
        t = getattr(self, "_t_cpos", None)
        if t is None:
            if self.eperm is None:                       # destination-sorted input: CSR position == original edge id
                t = self.t_eperm
            else:
                inv = torch.empty(self.n_edges, device=self.device, dtype=torch.int32)
                inv[self.eperm.long()] = torch.arange(self.n_edges, device=self.device, dtype=torch.int32)
                t = inv[self.t_eperm.long()].contiguous()
            self._t_cpos = t
        return t

    @property
    def c_struct(self):
        return C.byref(self._c)

    def nbytes(self) -> int:
        tot = 0
        for t in (self.rowptr, self.col, self.eperm, self.t_rowptr, self.t_col, self.t_eperm):
            if t is not None:
                tot += t.numel() * t.element_size()
        return tot


# ----------------------------------------------------------------------------------------
# cache: one Graph per live edge_index tensor object (same object reused by every layer and
# epoch in the reference's training loops).  Keyed by the tensor OBJECT through a weak
# reference, never by data_ptr: a recycled allocation must not resurrect a stale structure.
# ----------------------------------------------------------------------------------------
_cache: dict = {}      # id(tensor) -> (weakref to tensor, key, Graph)
_by_storage: dict = {}  # (data_ptr, shape, stride, dtype, num_nodes) -> id of the LIVE tensor that owns the entry


def _alias_key(t, num_nodes):
    return (t.data_ptr(), tuple(t.shape), tuple(t.stride()), t.dtype, int(num_nodes))


def graph_of(edge_index, num_nodes: Optional[int] = None) -> Graph:
    """Return the cached Graph for ``edge_index`` (a (2,E) tensor) or pass a Graph through.

    Hits: the same tensor object at the same version, or an ALIAS of a cached tensor that is still alive
    (same storage pointer, shape, strides and shared version counter) -- e.g. the ``edge_index.detach()`` that
    re-entrant ``torch.utils.checkpoint`` hands to the recomputed layer.  While the cached tensor is alive its
    storage cannot have been recycled, so an alias necessarily holds the same edges."""
    if isinstance(edge_index, Graph):
        return edge_index
    if num_nodes is None:
        raise ValueError("num_nodes is required to build a Graph from edge_index")
    ident = id(edge_index)
    key = (edge_index._version, int(num_nodes), edge_index.data_ptr(), tuple(edge_index.shape))
    hit = _cache.get(ident)
    if hit is not None and hit[0]() is edge_index and hit[1] == key:
        return hit[2]
    akey = _alias_key(edge_index, num_nodes)
    owner = _by_storage.get(akey)
    if owner is not None:
        ohit = _cache.get(owner)
        if ohit is not None:
            live = ohit[0]()
            if live is not None and live._version == edge_index._version and ohit[1][0] == live._version \
                    and live.data_ptr() == edge_index.data_ptr():
                return ohit[2]
    g = Graph.from_edge_index(edge_index, int(num_nodes))

    def _evict(_r, ident=ident, akey=akey):
        _cache.pop(ident, None)
        if _by_storage.get(akey) == ident:
            _by_storage.pop(akey, None)

    _cache[ident] = (weakref.ref(edge_index, _evict), key, g)
    _by_storage[akey] = ident
    return g


_WARM = set()


def warm_up(device=None) -> bool:
    """Load the graph-build code object (``csrc/graph_build.hip``: histogram, rocPRIM scan / radix sort, gather, hub work
    list) and take the first small blocks of the allocator by building one 64-edge graph: the first ``Graph`` of a
    process otherwise pays ~100 ms of lazy module loading on top of the 25 ms the build of a 126 M-edge graph takes
    (bench.py ``graph_build_cold_ms``, VERDICT r5 weak #5c).  Called by ``install()`` when a GPU is present; no-op on a
    CPU-only host, idempotent per device."""
    if not torch.cuda.is_available():
        return False
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev in _WARM:
        return True
    ei = torch.stack([torch.arange(64, device=dev) % 16, torch.arange(64, device=dev) % 13])
    Graph.from_edge_index(ei, 16)
    torch.cuda.synchronize(dev)
    _WARM.add(dev)
    return True


def clear_cache() -> None:
    _cache.clear()
    _by_storage.clear()
    _scatter_cache.clear()


# ----------------------------------------------------------------------------------------
# "scatter" graphs: aggregate the rows of an ALREADY materialised (E, C) tensor by a destination index
# (utils/pyg_util.scatter_, GenMessagePassing.aggregate).  Source of edge e is row e, so only the destination index
# defines the structure; it is cached per live index tensor exactly like edge_index above (the reference's
# sem_seg_sparse / part_sem_seg models call scatter_ with the same `edge_index[1]` object in every layer and epoch).
# ----------------------------------------------------------------------------------------
_scatter_cache: dict = {}   # (id(owner), geometry) -> (weakref to owner, Graph); owner = the index tensor or its view base


def scatter_graph_of(index: torch.Tensor, n_dst: int) -> Graph:
    """The usual call is ``scatter_(..., edge_index[1], ...)``: a fresh VIEW object per call whose base
    (``edge_index``) stays alive in the caller.  Entries are keyed by the owner object + the view's geometry and
    version and die with the owner, so a recycled allocation never resurrects a stale structure."""
    owner = index._base if index._base is not None else index
    key = (id(owner), index.storage_offset(), tuple(index.shape), tuple(index.stride()), index._version, int(n_dst))
    hit = _scatter_cache.get(key)
    if hit is not None and hit[0]() is owner:
        return hit[1]
    E = index.numel()
    ids = torch.arange(E, device=index.device, dtype=torch.long)
    g = Graph(ids, index.reshape(-1), n_src=E, n_dst=int(n_dst))

    def _evict(_r, key=key):
        _scatter_cache.pop(key, None)

    _scatter_cache[key] = (weakref.ref(owner, _evict), g)
    return g
