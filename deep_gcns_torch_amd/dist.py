"""Node-partitioned sparse aggregation across the GPUs of one node (one process per GPU, RCCL/xGMI).

The reference has no distributed path; this is the MI355X-native scaling of the
sparse hot path (§8e).  Destination rows are independent, so nodes are split into `world`
contiguous ranges balanced by in-edge count; rank r owns x[lo_r:hi_r], the output rows of its
range and the CSR rows of its destinations, whose column ids keep pointing at GLOBAL sources:

  forward   x_full = all_gather(x_local)           one collective: the halo feature exchange
            out_r  = gen_aggregate(x_full, G_r)     purely local HIP kernel on a rectangular graph
  backward  partial grad_x_full from the local edges (deterministic CSC walk over local edges)
            grad_x_local = reduce_scatter(sum)     one collective

Shards are padded to the largest range so both collectives are the single-buffer tensor forms
(`all_gather_into_tensor` / `reduce_scatter_tensor`); column ids are remapped once to the padded
layout, so no compaction copy is needed on either side.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist

from .graph import Graph


def balanced_bounds(in_degree: torch.Tensor, world: int) -> List[int]:
    """Contiguous node ranges with (nearly) equal in-edge counts: bounds[r] .. bounds[r+1]."""
    n = in_degree.numel()
    csum = torch.cumsum(in_degree.to(torch.int64), 0)
    total = int(csum[-1]) if n else 0
    bounds = [0]
    for r in range(1, world):
        target = (total * r) // world
        cut = int(torch.searchsorted(csum, torch.tensor([target], device=csum.device, dtype=csum.dtype),
                                     right=False)) if n else 0
        bounds.append(min(max(cut, bounds[-1]), n))
    bounds.append(n)
    return bounds


class PartitionedGraph:
    """Rank-local view of a destination-partitioned graph."""

    def __init__(self, graph: Graph, bounds: List[int], rank: int, world: int, max_rows: int, n_local_edges: int):
        self.graph = graph            # n_dst = local rows, n_src = world * max_rows (padded global layout)
        self.bounds = bounds
        self.rank, self.world = rank, world
        self.lo, self.hi = bounds[rank], bounds[rank + 1]
        self.max_rows = max_rows
        self.n_local_edges = n_local_edges

    @property
    def n_local(self) -> int:
        return self.hi - self.lo

    @classmethod
    def from_edge_index(cls, edge_index: torch.Tensor, num_nodes: int, rank: int, world: int,
                        bounds: Optional[List[int]] = None, need_transpose: bool = True) -> "PartitionedGraph":
        src, dst = edge_index[0], edge_index[1]
        if bounds is None:
            bounds = balanced_bounds(torch.bincount(dst, minlength=num_nodes), world)
        assert len(bounds) == world + 1 and bounds[0] == 0 and bounds[-1] == num_nodes
        max_rows = max(bounds[r + 1] - bounds[r] for r in range(world))
        max_rows = (max_rows + 3) // 4 * 4
        lo, hi = bounds[rank], bounds[rank + 1]
        mine = (dst >= lo) & (dst < hi)
        lsrc, ldst = src[mine], dst[mine] - lo
        b = torch.tensor(bounds, device=src.device, dtype=src.dtype)
        owner = torch.bucketize(lsrc, b[1:], right=True)          # rank owning each source node
        padded_src = owner * max_rows + (lsrc - b[owner])
        g = Graph(padded_src, ldst, n_src=world * max_rows, n_dst=hi - lo, need_transpose=need_transpose)
        return cls(g, list(bounds), rank, world, max_rows, int(mine.sum()))


def _supports_tensor_collectives(group) -> bool:
    return dist.get_backend(group) != "gloo"   # gloo (CPU test harness) lacks reduce_scatter


class _AllGatherRows(torch.autograd.Function):
    """(n_local, C) -> (world * max_rows, C) padded global layout; backward = reduce-scatter(sum)."""

    @staticmethod
    def forward(ctx, x_local, max_rows: int, group):
        world = dist.get_world_size(group)
        n_local, C = x_local.shape
        ctx.n_local, ctx.max_rows, ctx.group = n_local, max_rows, group
        send = x_local
        if n_local != max_rows:
            send = x_local.new_zeros(max_rows, C)
            send[:n_local] = x_local
        send = send.contiguous()
        full = x_local.new_empty(world * max_rows, C)
        if _supports_tensor_collectives(group):
            dist.all_gather_into_tensor(full, send, group=group)
        else:
            dist.all_gather(list(full.view(world, max_rows, C).unbind(0)), send, group=group)
        return full

    @staticmethod
    def backward(ctx, g_full):
        group = ctx.group
        world = dist.get_world_size(group)
        rank = dist.get_rank(group)
        g_full = g_full.contiguous()
        C = g_full.size(1)
        if _supports_tensor_collectives(group):
            out = g_full.new_empty(ctx.max_rows, C)
            dist.reduce_scatter_tensor(out, g_full, op=dist.ReduceOp.SUM, group=group)
        else:
            tmp = g_full.clone()
            dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=group)
            out = tmp.view(world, ctx.max_rows, C)[rank]
        return out[:ctx.n_local].contiguous(), None, None


def all_gather_rows(x_local: torch.Tensor, part: PartitionedGraph, group=None) -> torch.Tensor:
    return _AllGatherRows.apply(x_local, part.max_rows, group)


def partitioned_gen_aggregate(x_local: torch.Tensor, part: PartitionedGraph, aggr: str = "softmax", group=None,
                              local_aggregate=None, **kw) -> torch.Tensor:
    """Aggregation of this rank's destination rows; ``x_local`` = this rank's feature rows.
    ``local_aggregate(x_full, graph, aggr=..., **kw)`` defaults to the HIP op; the CPU/gloo tests
    inject the oracle there to exercise partitioning + collectives without a GPU."""
    if local_aggregate is None:
        from . import ops
        local_aggregate = ops.gen_aggregate
    x_full = all_gather_rows(x_local, part, group)
    return local_aggregate(x_full, part.graph, aggr=aggr, **kw)
