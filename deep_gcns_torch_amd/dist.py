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

Second scheme, same caller contract (node-partitioned rows in, node-partitioned rows out): the
**channel-transposed** exchange (`TransposedGraph`, `transposed_gen_aggregate`).  Every aggregator of the
path works per channel, so instead of bringing all N source rows x C channels to every rank
(all-gather: each rank receives (W-1)/W * N*C floats per direction) the layer is transposed:

  forward   all_to_all: rank r receives channels [r*C/W, (r+1)*C/W) of ALL rows   (N*C/W floats in)
            every rank aggregates ALL edges for its own channel block (whole graph replicated:
            indices are 12 B/edge against 4C B/edge of features)
            all_to_all back to row ownership                                        (N*C/W floats out)
  backward  the same two exchanges in the opposite direction, no reduction needed (each rank holds the
            complete gradient of its channel block)

Per direction that is 2*N*C/W floats per rank instead of (W-1)*N*C/W: W-1 = 7 x less at 8 ranks
for graphs without locality, where the all-gather scheme is bound by the per-link xGMI rate.  It also
load-balances any degree distribution exactly.  It needs edge-feature-free aggregation and C % (4*W) == 0;
`aggregate(...)` picks the scheme.

2-D form (``node_groups`` = Wn > 1, W = Wn * Wc): rank (a, b) aggregates channel block b (C/Wc channels) for the
destination rows of node group a only.  Gathered rows stay >= 128 bytes when C/W would fall below 32 channels
(C = 128 on 8 ranks: 4 channel groups x 2 node groups), at the price of a Wn-fold replicated input exchange and a
sum over node groups in the backward exchange (both still all-to-all, no reduction collective).

Halo scheme (``HaloGraph``, ``halo_gen_aggregate``): the destination partition of the first scheme, but only the
remote rows a partition actually references travel (index lists exchanged once at build time, uneven all-to-all per
layer; gradients of halo rows return the same way and are summed into their owners' rows by a CSR add kernel).
This is the scheme for graphs with locality; on a uniform random graph it degenerates to the all-gather volume.

``build_partition(..., scheme=...)`` + ``aggregate(x_local, part, ...)`` are the scheme-agnostic entry points.
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
        part = cls(g, list(bounds), rank, world, max_rows, int(mine.sum()))
        part.edge_mask = mine         # rows of a global (E, C) edge_attr that belong to this rank: edge_attr[part.edge_mask]
        return part


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


def _channel_chunks(C: int, nchunk: int):
    """Contiguous channel blocks (multiples of 4 channels so rows stay 16-byte aligned)."""
    nchunk = max(1, min(nchunk, C // 4 if C >= 4 else 1))
    base = (C // nchunk) // 4 * 4
    if base == 0:
        return [(0, C)]
    cuts, lo = [], 0
    for i in range(nchunk):
        hi = C if i == nchunk - 1 else lo + base
        cuts.append((lo, hi))
        lo = hi
    return cuts


class _PipelinedPartitionedAggregate(torch.autograd.Function):
    """Channel-pipelined form of ``local_aggregate(all_gather(x_local), graph)``.

    Every aggregator works per channel, so the layer splits exactly into independent channel blocks.  All
    block all-gathers are enqueued up front on the collective's stream; the compute stream waits only for
    block i before aggregating it, so the exchange of block i+1 overlaps the kernel of block i.  The backward
    mirrors it: the reduce-scatter of block i is left in flight while block i+1's gradient kernel runs, and
    everything is awaited once at the end.  (With the stock "wait after every collective" form the two never
    overlap.)"""

    @staticmethod
    def forward(ctx, x_local, part, group, local_aggregate, aggr, kw, nchunk):
        world = dist.get_world_size(group)
        n_local, C = x_local.shape
        mr = part.max_rows
        cuts = _channel_chunks(C, nchunk)
        tensor_coll = _supports_tensor_collectives(group)
        fulls, works = [], []
        for lo, hi in cuts:
            send = x_local.new_zeros(mr, hi - lo)
            send[:n_local] = x_local[:, lo:hi]
            full = x_local.new_empty(world * mr, hi - lo)
            if tensor_coll:
                w = dist.all_gather_into_tensor(full, send, group=group, async_op=True)
            else:
                w = dist.all_gather(list(full.view(world, mr, hi - lo).unbind(0)), send, group=group, async_op=True)
            fulls.append(full)
            works.append(w)
        outs, leaves = [], []
        for full, w in zip(fulls, works):
            w.wait()                                     # stream-level wait for THIS block only
            with torch.enable_grad():
                leaf = full.detach().requires_grad_(x_local.requires_grad)
                out = local_aggregate(leaf, part.graph, aggr=aggr, **kw)
            leaves.append(leaf)
            outs.append(out)
        ctx.part, ctx.group, ctx.cuts, ctx.tensor_coll = part, group, cuts, tensor_coll
        ctx.n_local = n_local
        ctx.leaves, ctx.outs = leaves, outs              # nested graphs of the per-block aggregations
        return torch.cat([o.detach() for o in outs], dim=1)

    @staticmethod
    def backward(ctx, g):
        part, group = ctx.part, ctx.group
        world = dist.get_world_size(group)
        rank = dist.get_rank(group)
        mr = part.max_rows
        pieces, works, tmps = [], [], []
        for (lo, hi), leaf, out in zip(ctx.cuts, ctx.leaves, ctx.outs):
            if out.requires_grad:
                gfull, = torch.autograd.grad(out, leaf, g[:, lo:hi].contiguous())
                gfull = gfull.contiguous()
            else:                                         # a rank without local edges contributes nothing
                gfull = torch.zeros_like(leaf)
            if ctx.tensor_coll:
                dst = gfull.new_empty(mr, hi - lo)
                works.append(dist.reduce_scatter_tensor(dst, gfull, op=dist.ReduceOp.SUM, group=group, async_op=True))
                pieces.append(dst)
            else:
                tmp = gfull.clone()
                works.append(dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=group, async_op=True))
                pieces.append(tmp.view(world, mr, hi - lo)[rank])
            tmps.append(gfull)                            # keep alive until the collective has consumed it
        for w in works:
            w.wait()
        ctx.leaves = ctx.outs = None
        grad = torch.cat([p[:ctx.n_local] for p in pieces], dim=1)
        return grad, None, None, None, None, None, None


def _pipelinable(kw: dict) -> bool:
    """The channel-pipelined Function differentiates w.r.t. the gathered feature blocks ONLY (its backward calls
    ``autograd.grad(out, leaf)``), and hands every block the keyword arguments unchanged.  Anything else that
    needs a gradient (learnable ``t`` / ``p``: 1-element Parameters whose per-rank gradients the caller must
    SUM-all-reduce like any replicated parameter) or that is per-channel data (``edge_attr`` (E, C), a fused
    ``edge_encoder``) takes the plain all-gather -> aggregate composition, where autograd sees everything."""
    if kw.get("learn_t") or kw.get("learn_p"):
        return False
    if kw.get("edge_attr") is not None or kw.get("edge_encoder") is not None:
        return False
    return not any(isinstance(v, torch.Tensor) and v.requires_grad for v in kw.values())


def partitioned_gen_aggregate(x_local: torch.Tensor, part: PartitionedGraph, aggr: str = "softmax", group=None,
                              local_aggregate=None, pipeline_chunks: Optional[int] = None, **kw) -> torch.Tensor:
    """Aggregation of this rank's destination rows; ``x_local`` = this rank's feature rows.

    ``pipeline_chunks`` > 1 splits the channels into that many blocks and overlaps the exchange of one block
    with the aggregation of the previous one (see ``_PipelinedPartitionedAggregate``); 1 gives the plain
    all-gather -> aggregate composition.  ``local_aggregate(x_full, graph, aggr=..., **kw)`` defaults to the HIP
    op; the CPU/gloo tests inject the oracle there to exercise partitioning + collectives without a GPU."""
    if local_aggregate is None:
        from . import ops
        local_aggregate = ops.gen_aggregate
    C = x_local.size(1)
    if pipeline_chunks is None:
        # measured on one MI355X (products shape, C=128): 2 blocks cost +7 % kernel time, 4 blocks +28 % (narrower
        # row gathers); the more ranks, the more the exchange dominates and the more overlap is worth
        pipeline_chunks = 2 if dist.get_world_size(group) <= 2 else 4
    if (pipeline_chunks > 1 and dist.get_world_size(group) > 1 and C >= 8 and C % 4 == 0
            and _pipelinable(kw)):
        return _PipelinedPartitionedAggregate.apply(x_local, part, group, local_aggregate, aggr, kw, pipeline_chunks)
    x_full = all_gather_rows(x_local, part, group)
    return local_aggregate(x_full, part.graph, aggr=aggr, **kw)


# ------------------------------------------------------------------------------------------------
# channel-transposed scheme (optionally 2-D: node groups x channel groups)
# ------------------------------------------------------------------------------------------------
def equal_row_bounds(num_nodes: int, world: int) -> List[int]:
    return [num_nodes * r // world for r in range(world + 1)]


def default_node_groups(channels: int, world: int) -> int:
    """Fewest node groups that keep a rank's channel block at >= 32 channels: a gathered row is then at least one
    128-byte line (MI355X fetches whole lines: 16-channel rows cost the same HBM/MALL traffic as 32-channel ones)."""
    wn = 1
    while (world // wn) > 1 and channels // (world // wn) < 32 and world % (wn * 2) == 0:
        wn *= 2
    return wn


class TransposedGraph:
    """Rank (a, b) = (node group, channel group), rank = a * Wc + b, W = Wn * Wc.

    It aggregates, for channel block b (C / Wc channels), the edges whose DESTINATION row belongs to node group a
    (the rows owned by ranks a*Wc .. a*Wc+Wc-1), reading source rows of the whole graph.  Ids are in the padded
    row layout ``owner * max_rows + local index`` that the equal-split all-to-all produces; destinations are
    re-based to the group.  Wn = 1 is the pure channel transpose: every rank walks ALL edges for C / W channels."""

    def __init__(self, graph: Graph, bounds: List[int], rank: int, world: int, max_rows: int, node_groups: int):
        self.graph = graph            # n_src = world * max_rows, n_dst = channel_groups * max_rows
        self.bounds = bounds
        self.rank, self.world = rank, world
        self.lo, self.hi = bounds[rank], bounds[rank + 1]
        self.max_rows = max_rows
        self.node_groups = node_groups
        self.channel_groups = world // node_groups
        self.node_group = rank // self.channel_groups
        self.channel_group = rank % self.channel_groups
        in_group = [mr if r // self.channel_groups == self.node_group else 0
                    for r, mr in enumerate([max_rows] * world)]
        self.group_splits = in_group  # rows exchanged with each rank in the group-local all-to-all

    @property
    def n_local(self) -> int:
        return self.hi - self.lo

    @property
    def n_edges(self) -> int:
        return self.graph.n_edges

    @classmethod
    def from_edge_index(cls, edge_index: torch.Tensor, num_nodes: int, rank: int, world: int,
                        bounds: Optional[List[int]] = None, need_transpose: bool = True,
                        node_groups: int = 1) -> "TransposedGraph":
        assert world % node_groups == 0
        wc = world // node_groups
        if bounds is None:
            if node_groups == 1:      # every rank walks all edges: only the rows need balancing
                bounds = equal_row_bounds(num_nodes, world)
            else:                     # node groups split the edges by destination: balance in-edges
                bounds = balanced_bounds(torch.bincount(edge_index[1], minlength=num_nodes), world)
        assert len(bounds) == world + 1 and bounds[0] == 0 and bounds[-1] == num_nodes
        max_rows = max(max(bounds[r + 1] - bounds[r] for r in range(world)), 1)
        b = torch.tensor(bounds, device=edge_index.device, dtype=edge_index.dtype)

        def padded(ids):
            owner = torch.bucketize(ids, b[1:], right=True)
            return owner * max_rows + (ids - b[owner])

        src, dst = padded(edge_index[0]), padded(edge_index[1])
        if node_groups > 1:
            a = rank // wc
            mine = (dst >= a * wc * max_rows) & (dst < (a + 1) * wc * max_rows)
            src, dst = src[mine], dst[mine] - a * wc * max_rows
        g = Graph(src, dst, n_src=world * max_rows, n_dst=wc * max_rows, need_transpose=need_transpose)
        return cls(g, list(bounds), rank, world, max_rows, node_groups)


def _replicate_rows_to_blocks(x_local: torch.Tensor, tg: TransposedGraph, lo: int, hi: int, group, async_op: bool):
    """Node-owned rows (n_local, C) -> sub-block [lo, hi) of this rank's channel block for ALL rows,
    (world*max_rows, hi-lo).  Every node group gets its own copy (Wn-fold replication of the send)."""
    world, mr, n_local, wc, wn = tg.world, tg.max_rows, tg.n_local, tg.channel_groups, tg.node_groups
    Cw = x_local.size(1) // wc
    cs = hi - lo
    send = x_local.new_zeros(wn, wc, mr, cs) if n_local != mr else x_local.new_empty(wn, wc, mr, cs)
    send[:, :, :n_local] = x_local.view(n_local, wc, Cw)[:, :, lo:hi].transpose(0, 1).unsqueeze(0)
    recv = x_local.new_empty(world * mr, cs)
    work = dist.all_to_all_single(recv, send.view(world * mr, cs), group=group, async_op=async_op)
    return recv, work


def _sum_blocks_into_rows(dst_local: torch.Tensor, y: torch.Tensor, tg: TransposedGraph, lo: int, hi: int, group):
    """Adjoint of ``_replicate_rows_to_blocks``: y (world*max_rows, cs) = this rank's values for all rows; every row
    owner receives the Wn x Wc pieces of its rows, sums over node groups and places the channel blocks."""
    recv = torch.empty_like(y)
    work = dist.all_to_all_single(recv, y.contiguous(), group=group, async_op=True)

    def finish():
        work.wait()
        world, mr, n_local, wc, wn = tg.world, tg.max_rows, tg.n_local, tg.channel_groups, tg.node_groups
        Cw = dst_local.size(1) // wc
        pieces = recv.view(wn, wc, mr, hi - lo)
        summed = pieces[0] if wn == 1 else pieces.sum(0)
        dst_local.view(n_local, wc, Cw)[:, :, lo:hi] = summed[:, :n_local].transpose(0, 1)
    return finish


def _group_blocks_to_rows(dst_local: torch.Tensor, y: torch.Tensor, tg: TransposedGraph, lo: int, hi: int, group):
    """y (Wc*max_rows, cs) = rows of this rank's node group, its channel sub-block -> each row owner in the group
    receives the Wc channel blocks of its rows (all-to-all inside the node group; other ranks exchange nothing)."""
    recv = torch.empty_like(y)
    sp = tg.group_splits
    work = dist.all_to_all_single(recv, y.contiguous(), output_split_sizes=sp, input_split_sizes=sp, group=group,
                                  async_op=True)

    def finish():
        work.wait()
        mr, n_local, wc = tg.max_rows, tg.n_local, tg.channel_groups
        Cw = dst_local.size(1) // wc
        dst_local.view(n_local, wc, Cw)[:, :, lo:hi] = recv.view(wc, mr, hi - lo)[:, :n_local].transpose(0, 1)
    return finish


def _group_rows_to_blocks(g_local: torch.Tensor, tg: TransposedGraph, lo: int, hi: int, group, async_op: bool):
    """Adjoint of ``_group_blocks_to_rows``: node-owned rows (n_local, C) -> (Wc*max_rows, cs) rows of the node
    group for this rank's channel sub-block."""
    mr, n_local, wc = tg.max_rows, tg.n_local, tg.channel_groups
    Cw = g_local.size(1) // wc
    cs = hi - lo
    send = g_local.new_zeros(wc, mr, cs) if n_local != mr else g_local.new_empty(wc, mr, cs)
    send[:, :n_local] = g_local.view(n_local, wc, Cw)[:, :, lo:hi].transpose(0, 1)
    recv = g_local.new_empty(wc * mr, cs)
    sp = tg.group_splits
    work = dist.all_to_all_single(recv, send.view(wc * mr, cs), output_split_sizes=sp, input_split_sizes=sp,
                                  group=group, async_op=async_op)
    return recv, work


class _TransposedAggregate(torch.autograd.Function):
    """all_to_all (rows -> channel block) -> local aggregation -> all_to_all back, pipelined over sub-blocks of the
    rank's channel block so the exchanges overlap the kernels (see module docstring)."""

    @staticmethod
    def forward(ctx, x_local, tg, group, local_aggregate, aggr, kw, nchunk):
        n_local, C = x_local.shape
        x_local = x_local.contiguous()
        cuts = _channel_chunks(C // tg.channel_groups, nchunk)
        ins = [_replicate_rows_to_blocks(x_local, tg, lo, hi, group, True) for lo, hi in cuts]
        out_local = x_local.new_empty(n_local, C)
        leaves, outs, finishers = [], [], []
        for (lo, hi), (blk, w) in zip(cuts, ins):
            w.wait()
            with torch.enable_grad():
                leaf = blk.requires_grad_(x_local.requires_grad)
                out = local_aggregate(leaf, tg.graph, aggr=aggr, **kw)
            leaves.append(leaf)
            outs.append(out)
            finishers.append(_group_blocks_to_rows(out_local, out.detach(), tg, lo, hi, group))
        for fin in finishers:
            fin()
        ctx.tg, ctx.group, ctx.cuts = tg, group, cuts
        ctx.leaves, ctx.outs = leaves, outs
        return out_local

    @staticmethod
    def backward(ctx, g):
        tg, group, cuts = ctx.tg, ctx.group, ctx.cuts
        g = g.contiguous()
        ins = [_group_rows_to_blocks(g, tg, lo, hi, group, True) for lo, hi in cuts]
        grad_local = g.new_empty(g.shape)
        finishers = []
        for (lo, hi), (gblk, w), leaf, out in zip(cuts, ins, ctx.leaves, ctx.outs):
            w.wait()
            if out.requires_grad:
                gfull, = torch.autograd.grad(out, leaf, gblk)
            else:                                         # no local edges
                gfull = torch.zeros_like(leaf)
            finishers.append(_sum_blocks_into_rows(grad_local, gfull, tg, lo, hi, group))
        for fin in finishers:
            fin()
        ctx.leaves = ctx.outs = None
        return grad_local, None, None, None, None, None, None


def transposed_supported(C: int, world: int, edge_attr=None, node_groups: int = 1) -> bool:
    return edge_attr is None and world % node_groups == 0 and C % (4 * (world // node_groups)) == 0


def transposed_gen_aggregate(x_local: torch.Tensor, tg: TransposedGraph, aggr: str = "softmax", group=None,
                             local_aggregate=None, pipeline_chunks: Optional[int] = None, **kw) -> torch.Tensor:
    """Aggregation of this rank's rows through the channel-transposed exchange; ``x_local`` = this rank's
    feature rows ``x[tg.lo:tg.hi]`` (n_local, C) with C % (4*channel_groups) == 0, no edge features, t/p not
    learnable (their gradients would need a cross-rank sum: use the all-gather scheme for those)."""
    if local_aggregate is None:
        from . import ops
        local_aggregate = ops.gen_aggregate
    C = x_local.size(1)
    if not transposed_supported(C, tg.world, kw.get("edge_attr"), tg.node_groups):
        raise ValueError("channel-transposed scheme needs no edge features and C % (4*channel_groups) == 0 "
                         f"(C={C}, world={tg.world}, node_groups={tg.node_groups})")
    if kw.get("learn_t") or kw.get("learn_p"):
        raise ValueError("learnable t/p: use the all-gather scheme (partitioned_gen_aggregate)")
    if x_local.size(0) != tg.n_local:
        raise ValueError(f"x_local has {x_local.size(0)} rows, this rank owns {tg.n_local}")
    if pipeline_chunks is None:
        # sub-blocks narrower than 32 channels (one 128-byte line per gathered row) waste gather bandwidth
        pipeline_chunks = max(1, min(4, (C // tg.channel_groups) // 32))
    return _TransposedAggregate.apply(x_local, tg, group, local_aggregate, aggr, kw, pipeline_chunks)


# ------------------------------------------------------------------------------------------------
# halo scheme: exchange only the rows a partition actually references
# ------------------------------------------------------------------------------------------------
class HaloGraph:
    """Destination-partitioned graph whose remote sources are the HALO only: at build time every rank tells each
    owner which of the owner's rows its edges reference (one all-to-all of index lists); per layer the owners send
    exactly those rows (``all_to_all_single`` with uneven splits) and the local kernel runs on
    ``[own rows | halo rows]``.  On a graph with locality (METIS-like orderings of real graphs) the halo is a small
    fraction of N; on a uniform random graph it degenerates to the all-gather volume plus a gather copy."""

    def __init__(self, graph: Graph, bounds, rank, world, send_idx, send_counts, recv_counts, back_graph, n_local_edges):
        self.graph = graph                  # n_src = n_local + n_halo, n_dst = n_local
        self.bounds = bounds
        self.rank, self.world = rank, world
        self.lo, self.hi = bounds[rank], bounds[rank + 1]
        self.send_idx = send_idx            # my local row ids requested by the others, grouped by requester
        self.send_counts = send_counts      # python ints, per requester
        self.recv_counts = recv_counts      # python ints, per owner
        self.back_graph = back_graph        # Graph(position in send_idx -> my local row): the backward's scatter-add
        self.n_local_edges = n_local_edges

    @property
    def n_local(self) -> int:
        return self.hi - self.lo

    @property
    def n_halo(self) -> int:
        return sum(self.recv_counts)

    @staticmethod
    def local_graph(edge_index: torch.Tensor, num_nodes: int, rank: int, world: int, bounds: Optional[List[int]] = None,
                    need_transpose: bool = True):
        """``(Graph over [own rows | halo rows], n_halo)`` of one rank WITHOUT the index exchange (no collective): what
        the rank's kernels run on.  ``bench.py --emulate-ranks`` times it for every rank on one GPU."""
        src, dst = edge_index[0], edge_index[1]
        if bounds is None:
            bounds = balanced_bounds(torch.bincount(dst, minlength=num_nodes), world)
        lo, hi = bounds[rank], bounds[rank + 1]
        mine = (dst >= lo) & (dst < hi)
        lsrc, ldst = src[mine], dst[mine] - lo
        is_local = (lsrc >= lo) & (lsrc < hi)
        uniq, inverse = torch.unique(lsrc[~is_local], return_inverse=True)
        col = torch.empty_like(lsrc)
        col[is_local] = lsrc[is_local] - lo
        col[~is_local] = (hi - lo) + inverse
        n_halo = int(uniq.numel())
        return Graph(col, ldst, n_src=hi - lo + n_halo, n_dst=hi - lo, need_transpose=need_transpose), n_halo

    @classmethod
    def from_edge_index(cls, edge_index: torch.Tensor, num_nodes: int, rank: int, world: int,
                        bounds: Optional[List[int]] = None, group=None, need_transpose: bool = True) -> "HaloGraph":
        src, dst = edge_index[0], edge_index[1]
        if bounds is None:
            bounds = balanced_bounds(torch.bincount(dst, minlength=num_nodes), world)
        assert len(bounds) == world + 1 and bounds[0] == 0 and bounds[-1] == num_nodes
        lo, hi = bounds[rank], bounds[rank + 1]
        n_local = hi - lo
        dev = edge_index.device
        b = torch.tensor(bounds, device=dev, dtype=src.dtype)
        mine = (dst >= lo) & (dst < hi)
        lsrc, ldst = src[mine], dst[mine] - lo
        is_local = (lsrc >= lo) & (lsrc < hi)
        uniq, inverse = torch.unique(lsrc[~is_local], return_inverse=True)     # ascending: grouped by owner
        owner = torch.bucketize(uniq, b[1:], right=True)
        recv_counts_t = torch.bincount(owner, minlength=world).to(torch.int64)
        send_counts_t = torch.empty_like(recv_counts_t)
        dist.all_to_all_single(send_counts_t, recv_counts_t, group=group)        # how many of MY rows each rank wants
        recv_counts = [int(v) for v in recv_counts_t.tolist()]
        send_counts = [int(v) for v in send_counts_t.tolist()]
        want = (uniq - b[owner]).to(torch.int64)                                 # owner-local ids of the rows I need
        send_idx = torch.empty(sum(send_counts), device=dev, dtype=torch.int64)
        dist.all_to_all_single(send_idx, want, output_split_sizes=send_counts, input_split_sizes=recv_counts, group=group)
        col = torch.empty_like(lsrc)
        col[is_local] = lsrc[is_local] - lo
        col[~is_local] = n_local + inverse
        g = Graph(col, ldst, n_src=n_local + int(uniq.numel()), n_dst=n_local, need_transpose=need_transpose)
        back = None
        if need_transpose and send_idx.numel() > 0:
            ids = torch.arange(send_idx.numel(), device=dev, dtype=send_idx.dtype)
            back = Graph(ids, send_idx, n_src=send_idx.numel(), n_dst=n_local, need_transpose=False)
        return cls(g, list(bounds), rank, world, send_idx, send_counts, recv_counts, back, int(mine.sum()))


class _HaloExchange(torch.autograd.Function):
    """x_local (n_local, C) -> [x_local | halo rows] (n_local + n_halo, C); backward: the halo rows' gradients travel
    back to their owners and are summed into the owners' rows in a fixed (CSR) order."""

    @staticmethod
    def forward(ctx, x_local, hg, group, local_add):
        n_local, C = x_local.shape
        ext = x_local.new_empty(n_local + hg.n_halo, C)
        ext[:n_local] = x_local
        send = x_local.index_select(0, hg.send_idx) if hg.send_idx.numel() else x_local.new_empty(0, C)
        dist.all_to_all_single(ext[n_local:], send, output_split_sizes=hg.recv_counts, input_split_sizes=hg.send_counts,
                               group=group)
        ctx.hg, ctx.group, ctx.local_add = hg, group, local_add
        return ext

    @staticmethod
    def backward(ctx, g_ext):
        hg = ctx.hg
        n_local = hg.n_local
        C = g_ext.size(1)
        g_ext = g_ext.contiguous()
        back = g_ext.new_empty(hg.send_idx.numel(), C)
        dist.all_to_all_single(back, g_ext[n_local:].contiguous(), output_split_sizes=hg.send_counts,
                               input_split_sizes=hg.recv_counts, group=ctx.group)
        grad = g_ext[:n_local]
        if back.size(0):
            grad = grad + ctx.local_add(back, hg)       # deterministic segment sum over the requesters
        return grad.contiguous(), None, None, None


def _halo_back_add_hip(back_rows, hg):
    from . import ops
    return ops.gen_aggregate(back_rows, hg.back_graph, aggr="add", relu_eps=False)


def _halo_back_add_torch(back_rows, hg):
    out = back_rows.new_zeros(hg.n_local, back_rows.size(1))
    return out.index_add_(0, hg.send_idx, back_rows)


def halo_gen_aggregate(x_local: torch.Tensor, hg: HaloGraph, aggr: str = "softmax", group=None,
                       local_aggregate=None, **kw) -> torch.Tensor:
    """Aggregation of this rank's destination rows with a halo-only exchange (see ``HaloGraph``)."""
    hip = local_aggregate is None
    if hip:
        from . import ops
        local_aggregate = ops.gen_aggregate
    x_ext = _HaloExchange.apply(x_local, hg, group, _halo_back_add_hip if hip else _halo_back_add_torch)
    return local_aggregate(x_ext, hg.graph, aggr=aggr, **kw)


# ------------------------------------------------------------------------------------------------
# local-first scheme ("split the local CSR into {sources owned locally} U {halo sources}; launch the local
# part while the all-gather is in flight")
# ------------------------------------------------------------------------------------------------
class SplitGraph:
    """Destination partition whose edges are cut by the OWNER OF THE SOURCE: ``local`` (sources among this rank's own
    rows: n_src = n_local) and ``remote`` (sources on other ranks, ids in the padded all-gather layout).  The softmax
    aggregation of a row over the union merges exactly from the two partial results and their log-sum-exps, so the
    local kernel runs while the remote rows travel, and in the backward the remote gradient's reduce-scatter is in
    flight while the local gradient kernel runs."""

    def __init__(self, local: Graph, remote: Graph, bounds, rank, world, max_rows, n_local_edges):
        self.local, self.remote = local, remote
        self.graph = remote               # (phase_times / bench read .graph)
        self.bounds = bounds
        self.rank, self.world = rank, world
        self.lo, self.hi = bounds[rank], bounds[rank + 1]
        self.max_rows = max_rows
        self.n_local_edges = n_local_edges

    @property
    def n_local(self) -> int:
        return self.hi - self.lo

    @classmethod
    def from_edge_index(cls, edge_index, num_nodes, rank, world, bounds=None, need_transpose=True) -> "SplitGraph":
        src, dst = edge_index[0], edge_index[1]
        if bounds is None:
            bounds = balanced_bounds(torch.bincount(dst, minlength=num_nodes), world)
        assert len(bounds) == world + 1 and bounds[0] == 0 and bounds[-1] == num_nodes
        max_rows = max(bounds[r + 1] - bounds[r] for r in range(world))
        max_rows = (max_rows + 3) // 4 * 4
        lo, hi = bounds[rank], bounds[rank + 1]
        mine = (dst >= lo) & (dst < hi)
        lsrc, ldst = src[mine], dst[mine] - lo
        own = (lsrc >= lo) & (lsrc < hi)
        g_loc = Graph(lsrc[own] - lo, ldst[own], n_src=hi - lo, n_dst=hi - lo, need_transpose=need_transpose)
        b = torch.tensor(bounds, device=src.device, dtype=src.dtype)
        rs = lsrc[~own]
        owner = torch.bucketize(rs, b[1:], right=True)
        g_rem = Graph(owner * max_rows + (rs - b[owner]), ldst[~own], n_src=world * max_rows, n_dst=hi - lo,
                      need_transpose=need_transpose)
        return cls(g_loc, g_rem, list(bounds), rank, world, max_rows, int(mine.sum()))


def _hip_state_fns():
    from . import ops
    return ops.softmax_state_forward, ops.softmax_state_backward


def merge_softmax_states(oa, la, has_a, ob, lb, has_b):
    """(out, L) of the union of two disjoint edge sets from their partial (out, L); ``has_*``: (n, 1) bool, the row has
    edges in that set (L is 0, not -inf, for a row without edges)."""
    both = has_a & has_b
    wa = torch.where(both, torch.sigmoid(la - lb), has_a.to(oa.dtype).expand_as(oa))
    wb = torch.where(both, 1.0 - wa, has_b.to(oa.dtype).expand_as(oa))
    out = wa * oa + wb * ob
    L = torch.where(both, torch.logaddexp(la, lb), torch.where(has_a, la, lb))
    return out, L


class _SplitSoftmaxAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_local, sg, group, t, state_fwd, state_bwd, msg_kw):
        world = dist.get_world_size(group)
        n_local, C = x_local.shape
        mr = sg.max_rows
        send = x_local.new_zeros(mr, C)
        send[:n_local] = x_local.detach()
        full = x_local.new_empty(world * mr, C)
        tensor_coll = _supports_tensor_collectives(group)
        if tensor_coll:
            work = dist.all_gather_into_tensor(full, send, group=group, async_op=True)
        else:
            work = dist.all_gather(list(full.view(world, mr, C).unbind(0)), send, group=group, async_op=True)
        xl = x_local.detach().contiguous()
        oa, la = state_fwd(xl, sg.local, t, **msg_kw)    # runs while the remote rows are in flight
        work.wait()
        ob, lb = state_fwd(full, sg.remote, t, **msg_kw)
        if state_fwd is _hip_state_fns()[0]:
            from . import ops
            out, L = ops.softmax_state_merge(oa, la, sg.local, ob, lb, sg.remote)      # one launch, in place in (oa, la)
        else:
            has_a = (sg.local.deg > 0).unsqueeze(1)
            has_b = (sg.remote.deg > 0).unsqueeze(1)
            out, L = merge_softmax_states(oa, la, has_a, ob, lb, has_b)
        ctx.sg, ctx.group, ctx.t, ctx.state_bwd, ctx.tensor_coll = sg, group, t, state_bwd, tensor_coll
        ctx.msg_kw = msg_kw
        ctx.save_for_backward(xl, full, L)
        return out

    @staticmethod
    def backward(ctx, g):
        xl, full, L = ctx.saved_tensors
        sg, group = ctx.sg, ctx.group
        world = dist.get_world_size(group)
        rank = dist.get_rank(group)
        mr, C = sg.max_rows, xl.size(1)
        g = g.contiguous()
        extra = dict(ctx.msg_kw)
        if ctx.state_bwd is _hip_state_fns()[1]:
            from . import ops
            prep = ops.softmax_state_prepare(g, L)          # one node-wise prologue for both launches (single-gather form)
            if prep is not None:
                extra["prep"] = prep
        g_full = ctx.state_bwd(full, sg.remote, g, L, ctx.t, **extra).contiguous()   # gradient of the remote rows first ...
        if ctx.tensor_coll:
            back = g_full.new_empty(mr, C)
            work = dist.reduce_scatter_tensor(back, g_full, op=dist.ReduceOp.SUM, group=group, async_op=True)
        else:
            tmp = g_full.clone()
            work = dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=group, async_op=True)
            back = None
        g_loc = ctx.state_bwd(xl, sg.local, g, L, ctx.t, **extra)               # ... its reduce-scatter flies during this
        work.wait()
        if back is None:
            back = tmp.view(world, mr, C)[rank]
        return g_loc + back[:xl.size(0)], None, None, None, None, None, None


class _AllGatherRowsAsync(torch.autograd.Function):
    """``_AllGatherRows`` whose forward only STARTS the collective: the caller waits on ``holder["work"]`` before the first
    use of the returned rows (the local-source aggregation runs in between).  Backward = reduce-scatter(sum), as there."""

    @staticmethod
    def forward(ctx, x_local, max_rows: int, group, holder):
        world = dist.get_world_size(group)
        n_local, C = x_local.shape
        ctx.n_local, ctx.max_rows, ctx.group = n_local, max_rows, group
        send = x_local.new_zeros(max_rows, C)
        send[:n_local] = x_local
        full = x_local.new_empty(world * max_rows, C)
        if _supports_tensor_collectives(group):
            holder["work"] = dist.all_gather_into_tensor(full, send, group=group, async_op=True)
        else:
            holder["work"] = dist.all_gather(list(full.view(world, max_rows, C).unbind(0)), send, group=group, async_op=True)
        holder["send"] = send                      # kept alive until the wait
        return full

    @staticmethod
    def backward(ctx, g_full):
        return _AllGatherRows.backward(ctx, g_full) + (None,)


class _TouchGrad(torch.autograd.Function):
    """``out`` unchanged, but ``other`` receives a zero gradient: keeps a collective's backward in the graph of a rank whose
    rows have no remote-source edge (every rank must enter the reduce-scatter)."""

    @staticmethod
    def forward(ctx, out, other):
        ctx.shape, ctx.dtype, ctx.device = other.shape, other.dtype, other.device
        return out.view_as(out)

    @staticmethod
    def backward(ctx, g):
        return g, torch.zeros(ctx.shape, dtype=ctx.dtype, device=ctx.device)


def _split_composed_aggregate(x_local, sg, aggr, group, local_aggregate, kw):
    """add / mean / max over a SplitGraph: the two partial aggregations are this library's kernels (each with its own
    autograd function), the (n, C) merge is elementwise torch whose backward autograd derives:
      add:   a + b                      mean: (add_a + add_b) / max(deg, 1)
      max:   the larger of the two partial maxima (a row without edges in one set takes the other's)
    The remote rows travel while the local-source part runs.  (Power-mean is NOT composed from partial OUTPUTS: the
    reference clamps the mean of m^p before the root, torch_message.py:70-74, so a partial output whose mean fell below
    the clamp -- every message of the part at eps -- no longer determines sum m^p; measured 4 % error on such rows.  It
    merges from the pre-clamp means instead: _SplitPowerAggregate.)"""
    holder = {}
    full = _AllGatherRowsAsync.apply(x_local, sg.max_rows, group, holder)
    base = aggr
    part_aggr = "add" if base == "mean" else base
    a = local_aggregate(x_local, sg.local, aggr=part_aggr, **kw)
    holder.pop("work").wait()
    holder.clear()
    b = local_aggregate(full, sg.remote, aggr=part_aggr, **kw)
    if sg.remote.n_edges == 0 or not b.requires_grad:
        a = _TouchGrad.apply(a, full)
    da, db = sg.local.deg.unsqueeze(1).to(a.dtype), sg.remote.deg.unsqueeze(1).to(a.dtype)
    if base == "add":
        return a + b
    if base == "mean":
        return (a + b) / (da + db).clamp_min(1.0)
    if base == "max":
        both = (da > 0) & (db > 0)
        return torch.where(both, torch.maximum(a, b), torch.where(da > 0, a, b))
    raise NotImplementedError(aggr)


_SPLIT_COMPOSED = ("add", "mean", "max")
_SPLIT_KWARGS = {"t", "eps", "relu_eps", "learn_t", "learn_p", "p", "edge_attr", "edge_encoder", "dim_size", "add_root"}


def _hip_power_state_fns():
    from . import ops
    return ops.power_state_forward, ops.power_state_backward


class _SplitPowerAggregate(torch.autograd.Function):
    """Power-mean over a SplitGraph with a fixed p: the two parts' PRE-CLAMP means (what the forward kernel saves anyway)
    merge exactly with the degrees, q = (q_a deg_a + q_b deg_b) / deg, out = clamp(q)^(1/p) as on one GPU; the backward
    forms the per-destination coefficient from the merged mean and hands it to the edge walk of each part (the remote
    part first: its reduce-scatter flies during the local one)."""

    @staticmethod
    def forward(ctx, x_local, sg, group, p, state_fwd, state_bwd, msg_kw):
        world = dist.get_world_size(group)
        n_local, C = x_local.shape
        mr = sg.max_rows
        send = x_local.new_zeros(mr, C)
        send[:n_local] = x_local.detach()
        full = x_local.new_empty(world * mr, C)
        tensor_coll = _supports_tensor_collectives(group)
        if tensor_coll:
            work = dist.all_gather_into_tensor(full, send, group=group, async_op=True)
        else:
            work = dist.all_gather(list(full.view(world, mr, C).unbind(0)), send, group=group, async_op=True)
        xl = x_local.detach().contiguous()
        _, qa = state_fwd(xl, sg.local, p, **msg_kw)     # runs while the remote rows are in flight
        work.wait()
        _, qb = state_fwd(full, sg.remote, p, **msg_kw)
        da, db = sg.local.deg.unsqueeze(1).to(qa.dtype), sg.remote.deg.unsqueeze(1).to(qa.dtype)
        deg = (da + db).clamp_min(1.0)
        q = (qa * da + qb * db) / deg
        r = q.clamp(1e-7, 10.0)                                              # torch_message.py:69-74 (ops.POW_LO / POW_HI)
        out = r.pow(1.0 / p)
        ctx.sg, ctx.group, ctx.p, ctx.state_bwd, ctx.tensor_coll, ctx.msg_kw = sg, group, p, state_bwd, tensor_coll, msg_kw
        ctx.save_for_backward(xl, full, q, deg)
        return out

    @staticmethod
    def backward(ctx, g):
        xl, full, q, deg = ctx.saved_tensors
        sg, group, p = ctx.sg, ctx.group, ctx.p
        world = dist.get_world_size(group)
        rank = dist.get_rank(group)
        mr, C = sg.max_rows, xl.size(1)
        r = q.clamp(1e-7, 10.0)
        inr = ((q >= 1e-7) & (q <= 10.0)).to(g.dtype)
        coef = (g * r.pow(1.0 / p - 1.0) * inr / deg).contiguous()
        g_full = ctx.state_bwd(full, sg.remote, coef, q, p, **ctx.msg_kw).contiguous()
        if ctx.tensor_coll:
            back = g_full.new_empty(mr, C)
            work = dist.reduce_scatter_tensor(back, g_full, op=dist.ReduceOp.SUM, group=group, async_op=True)
        else:
            tmp = g_full.clone()
            work = dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=group, async_op=True)
            back = None
        g_loc = ctx.state_bwd(xl, sg.local, coef, q, p, **ctx.msg_kw)
        work.wait()
        if back is None:
            back = tmp.view(world, mr, C)[rank]
        return g_loc + back[:xl.size(0)], None, None, None, None, None, None


def split_supported(aggr: str, kw: dict) -> bool:
    """Node features only; softmax / softmax_sg with a fixed temperature (exact merge of the partial states from their
    log-sum-exps, one HIP launch), power-mean with a fixed p (exact merge of the pre-clamp means), add / mean / max (partial
    aggregations merged elementwise).  ``eps`` /
    ``relu_eps`` are passed on to the kernels; anything that changes the result and is not handled (``add_root``, a ``dim_size`` other than the
    partition's rows, an unknown keyword) makes the caller fall back to the all-gather scheme instead of being dropped."""
    if any(k not in _SPLIT_KWARGS for k in kw):
        return False
    if kw.get("edge_attr") is not None or kw.get("edge_encoder") is not None or kw.get("add_root"):
        return False
    if aggr in _SPLIT_COMPOSED:               # partial aggregations merged elementwise
        return True
    if aggr == "power":                       # fixed p: merged from the pre-clamp means (_SplitPowerAggregate)
        return not kw.get("learn_p") and not isinstance(kw.get("p", 1.0), torch.Tensor)
    return (aggr in ("softmax", "softmax_sg") and not kw.get("learn_t") and not isinstance(kw.get("t", 1.0), torch.Tensor)
            and not kw.get("learn_p"))


def split_gen_aggregate(x_local: torch.Tensor, sg: SplitGraph, aggr: str = "softmax", group=None, state_fns=None,
                        local_aggregate=None, **kw) -> torch.Tensor:
    """Local-first aggregation of this rank's destination rows (``SplitGraph``).  softmax / softmax_sg with a fixed
    temperature (BASELINE config 4); ``state_fns = (forward, backward)`` defaults to the HIP entry points
    (``ops.softmax_state_forward`` / ``_backward``), the gloo tests inject torch restatements."""
    if not split_supported(aggr, kw):
        raise NotImplementedError("the local-first scheme covers softmax / softmax_sg / power with fixed t / p and add / "
                                  "mean / max on node features; use the allgather scheme otherwise")
    if kw.get("dim_size") not in (None, sg.n_local):
        raise ValueError(f"dim_size = {kw['dim_size']} but this rank's partition has {sg.n_local} destination rows")
    if aggr in _SPLIT_COMPOSED:
        if local_aggregate is None:
            from . import ops
            local_aggregate = ops.gen_aggregate
        ckw = {k: v for k, v in kw.items() if k in ("eps", "relu_eps")}
        return _split_composed_aggregate(x_local, sg, aggr, group, local_aggregate, ckw)
    msg_kw = {k: kw[k] for k in ("eps", "relu_eps") if k in kw}
    if aggr == "power":
        fwd, bwd = state_fns or _hip_power_state_fns()
        return _SplitPowerAggregate.apply(x_local, sg, group, float(kw.get("p", 1.0)), fwd, bwd, msg_kw)
    fwd, bwd = state_fns or _hip_state_fns()
    return _SplitSoftmaxAggregate.apply(x_local, sg, group, float(kw.get("t", 1.0)), fwd, bwd, msg_kw)


def aggregate(x_local: torch.Tensor, part, aggr: str = "softmax", group=None, **kw) -> torch.Tensor:
    """Scheme-agnostic entry: ``part`` is a PartitionedGraph (all-gather scheme) or a TransposedGraph."""
    if isinstance(part, SplitGraph):
        kw.pop("pipeline_chunks", None)
        return split_gen_aggregate(x_local, part, aggr=aggr, group=group, **kw)
    if isinstance(part, TransposedGraph):
        return transposed_gen_aggregate(x_local, part, aggr=aggr, group=group, **kw)
    if isinstance(part, HaloGraph):
        kw.pop("pipeline_chunks", None)
        return halo_gen_aggregate(x_local, part, aggr=aggr, group=group, **kw)
    return partitioned_gen_aggregate(x_local, part, aggr=aggr, group=group, **kw)


def exchange_bytes(edge_index: torch.Tensor, num_nodes: int, channels: int, rank: int, world: int,
                   node_groups: Optional[int] = None, bounds: Optional[List[int]] = None) -> dict:
    """Bytes THIS rank receives per direction (forward; the backward sends the same amount back) under each scheme, and how
    its edges split by the owner of the source -- computed from the edge list alone, no collective:
      allgather / split: (W - 1) padded row blocks of C channels;   halo: the distinct remote source rows of its edges;
      transposed: the rows of the other ranks in its channel block + the group-local exchange of the output."""
    src, dst = edge_index[0], edge_index[1]
    if bounds is None:
        bounds = balanced_bounds(torch.bincount(dst, minlength=num_nodes), world)
    max_rows = (max(bounds[r + 1] - bounds[r] for r in range(world)) + 3) // 4 * 4
    lo, hi = bounds[rank], bounds[rank + 1]
    mine = (dst >= lo) & (dst < hi)
    lsrc = src[mine]
    own = (lsrc >= lo) & (lsrc < hi)
    n_edges, n_own = int(mine.sum()), int(own.sum())
    n_halo = int(torch.unique(lsrc[~own]).numel())
    out = dict(edges=n_edges, local_source_edges=n_own, remote_source_edges=n_edges - n_own, halo_rows=n_halo,
               rows=hi - lo, allgather=(world - 1) * max_rows * channels * 4, halo=n_halo * channels * 4)
    out["split"] = out["allgather"]
    if node_groups is None:
        node_groups = default_node_groups(channels, world)
    if world > 1 and transposed_supported(channels, world, None, node_groups):
        wc = world // node_groups
        cw = channels // wc
        eq = equal_row_bounds(num_nodes, world)
        mr = (max(eq[r + 1] - eq[r] for r in range(world)) + 3) // 4 * 4
        out["transposed"] = (world - 1) * mr * cw * 4 + (wc - 1) * mr * cw * 4
    return out


def choose_scheme(costs: dict, aggr: Optional[str] = None, kw: Optional[dict] = None) -> str:
    """The scheme ``build_partition("auto")`` takes, from ``exchange_bytes`` (of the rank with the largest halo when the
    caller reduced it): the halo exchange when the partition references few remote rows (at most half the bytes of the
    best dense scheme: its all-to-all is irregular and pays a gather copy on the owner); otherwise the channel-transposed
    scheme where it applies (W / 2 times fewer bytes than the all-gather); otherwise the all-gather volume -- as the
    local-first ``split`` when the aggregator has an associative partial state and at least a quarter of the edges have
    a local source (that share of the kernel time hides the exchange), plain ``allgather`` if not."""
    dense = min(costs.get("transposed", costs["allgather"]), costs["allgather"])
    if costs["halo"] * 2 <= dense:
        return "halo"
    if "transposed" in costs:
        return "transposed"
    if aggr is not None and split_supported(aggr, kw or {}) and costs["local_source_edges"] * 4 >= costs["edges"]:
        return "split"
    return "allgather"


def build_partition(edge_index: torch.Tensor, num_nodes: int, channels: int, rank: int, world: int,
                    scheme: str = "auto", edge_attr=None, need_transpose: bool = True,
                    node_groups: Optional[int] = None, aggr: Optional[str] = None, group=None):
    """``scheme``: "transposed", "allgather", "halo", "split" or "auto" = ``choose_scheme`` on the bytes each scheme moves
    for THIS graph (``exchange_bytes``; the ranks agree on the largest halo with one all-reduce when a process group is
    up): a locality-ordered graph takes the halo exchange, a graph whose partitions reference every row the
    channel-transposed scheme (where the channel count allows it) or the all-gather / local-first pair.
    ``node_groups`` (transposed only): None = ``default_node_groups(channels, world)``."""
    if node_groups is None:
        node_groups = default_node_groups(channels, world)
    if scheme == "auto":
        if world <= 1:
            scheme = "allgather"
        else:
            costs = exchange_bytes(edge_index, num_nodes, channels, rank, world, node_groups)
            if edge_attr is not None:
                costs.pop("transposed", None)
            if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) == world:
                t = torch.tensor([costs["halo"], -costs["local_source_edges"] * 4 + costs["edges"]],
                                 device=edge_index.device, dtype=torch.int64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)      # every rank takes the same decision
                costs["halo"] = int(t[0])
                if int(t[1]) > 0:                                          # some rank has < 1/4 local-source edges
                    costs["local_source_edges"] = 0
            scheme = choose_scheme(costs, aggr, {} if edge_attr is None else {"edge_attr": edge_attr})
    if scheme == "transposed":
        return TransposedGraph.from_edge_index(edge_index, num_nodes, rank, world, need_transpose=need_transpose,
                                               node_groups=node_groups)
    if scheme == "allgather":
        return PartitionedGraph.from_edge_index(edge_index, num_nodes, rank, world, need_transpose=need_transpose)
    if scheme == "halo":
        return HaloGraph.from_edge_index(edge_index, num_nodes, rank, world, need_transpose=need_transpose)
    if scheme == "split":
        return SplitGraph.from_edge_index(edge_index, num_nodes, rank, world, need_transpose=need_transpose)
    raise ValueError(f"unknown scheme {scheme!r}")


def phase_times(x_local: torch.Tensor, g_local: torch.Tensor, part, aggr: str = "softmax", reps: int = 5, group=None,
                local_aggregate=None, **kw) -> dict:
    """Where one forward+backward of ``aggregate`` spends its time on this job (max over ranks, ms):
    ``step`` = exchange + kernels as they run together (pipelined / overlapped where the scheme does that);
    ``kernels`` = the rank-local aggregation kernels alone, on a feature tensor of the shape the exchange delivers;
    ``exchange_and_wait`` = step - kernels (what the collectives, packing and cross-rank waiting add on top).
    For the all-gather scheme the bare collectives (all-gather forward, reduce-scatter backward) are timed as well.
    Meant for reading a scaling run: kernels shrink with 1/W, the exchange does not."""
    import time as _time
    from . import ops
    dev = x_local.device

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        dist.barrier(group)
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)

    def timed(fn):
        fn()
        sync()
        t0 = _time.perf_counter()
        for _ in range(reps):
            fn()
        sync()
        tt = torch.tensor([(_time.perf_counter() - t0) / reps * 1e3], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=group)
        return float(tt.item())

    xl = x_local.detach().clone().requires_grad_(True)

    akw = dict(kw) if local_aggregate is None else dict(kw, local_aggregate=local_aggregate)
    local = ops.gen_aggregate if local_aggregate is None else local_aggregate

    def step():
        torch.autograd.grad(aggregate(xl, part, aggr=aggr, group=group, **akw), xl, g_local)

    C = x_local.size(1)
    c_loc = C // part.channel_groups if isinstance(part, TransposedGraph) else C
    g = part.graph
    xin = torch.randn(g.n_src, c_loc, device=dev, requires_grad=True)
    gout = torch.randn(g.n_dst, c_loc, device=dev)
    kk = {k: v for k, v in kw.items() if k != "pipeline_chunks"}

    def kernels():
        torch.autograd.grad(local(xin, g, aggr=aggr, **kk), xin, gout)

    out = {"step": timed(step), "kernels": timed(kernels)}
    out["exchange_and_wait"] = out["step"] - out["kernels"]
    if isinstance(part, PartitionedGraph):
        def gather():
            xf = all_gather_rows(xl, part, group)
            torch.autograd.grad(xf, xl, torch.ones_like(xf))
        out["allgather_plus_reduce_scatter_alone"] = timed(gather)
    out["local_edges"] = int(g.n_edges)
    out["local_channels"] = int(c_loc)
    return out



# ------------------------------------------------------------------------------------------------
# Model-level node partition (BASELINE config 4: DeeperGCN on ogbn-products, node-partitioned across the GPUs of a node)
# ------------------------------------------------------------------------------------------------
# ``with dist.partitioned(part): out = model(x_local, edge_index_placeholder)`` runs an UNCHANGED gcn_lib.sparse model on
# this rank's rows: while the context is active
#   * GenMessagePassing.propagate (gcn_lib/sparse/torch_vertex.py:68) aggregates through ``aggregate(x_local, part)`` --
#     the exchange scheme of ``part`` -- instead of building a graph from the edge_index argument;
#   * BatchNorm1d (norm_layer('batch'), gcn_lib/sparse/torch_nn.py:23-34) takes its batch statistics over ALL ranks'
#     rows: the per-rank partial sums (2 x C doubles: sum, sum of squares; in the backward sum g', sum g' xhat) are
#     all-reduced, forward and backward -- values, gradients and running statistics equal the
#     single-process ones;
#   * everything else of the layer (Linear, LayerNorm, ReLU, dropout, residual) is row-local.
# Parameters are replicated: after backward, ``allreduce_gradients(model)`` sums their gradients (one flat bucket).
import threading

# The context is looked up by code that autograd may run on ANOTHER thread than the one that entered it: the
# recomputation of a reentrant torch.utils.checkpoint (blocks.res_plus_layer, or the model file's own
# checkpoint(self.gcns[layer], ...)) runs on the device's autograd worker thread.  One process drives one GPU, so the
# lookup is: this thread's innermost context, else the innermost context entered (and not yet left) by any thread of
# the process.  Code that may run after the ``with`` block was left (loss.backward() outside it) captures the context
# object and re-enters it: blocks.res_plus_layer does (``reentered``).
# ONE partition per process at a time: a thread that enters a DIFFERENT context object while another thread holds one
# is refused (a thread without a context of its own would otherwise inherit whichever was entered last); the same
# object may be entered from several threads (``reentered`` on the autograd thread) and nested on one thread.
_ACTIVE = threading.local()     # .ctx = this thread's innermost context, .stack = the ones it shadows
_ACTIVE_PROCESS = []            # (context, entering thread id) of every live entry, innermost last
_ACTIVE_LOCK = threading.Lock()


class BatchSync:
    """All-reduce of BatchNorm partial sums across the ranks of ``group``; ``total_rows`` = rows of the whole graph."""

    def __init__(self, total_rows: int, group=None):
        self.total_rows, self.group = int(total_rows), group

    def reduce(self, partials: torch.Tensor) -> torch.Tensor:
        """(P, 2, C) fp32 per-workgroup partial sums of this rank -> (2, 2, C) fp32 = the global sums as a (hi, lo)
        pair (the finalize kernels add partials in float64, so the pair carries ~48 bits)."""
        tot = partials.double().sum(0)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=self.group)
        hi = tot.float()
        lo = (tot - hi.double()).float()
        return torch.stack([hi, lo]).contiguous()

    def reduce64(self, t: torch.Tensor) -> torch.Tensor:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t


class partitioned:
    """Context: run gcn_lib.sparse modules on the rows of one partition (see the section comment)."""

    def __init__(self, part, group=None, local_aggregate=None, **aggregate_kw):
        self.part, self.group, self.local_aggregate, self.kw = part, group, local_aggregate, aggregate_kw
        self.sync = BatchSync(part.bounds[-1], group)

    def __enter__(self):
        me = threading.get_ident()
        with _ACTIVE_LOCK:
            for other, tid in _ACTIVE_PROCESS:
                if other is not self and tid != me:
                    raise RuntimeError("dist.partitioned: another thread of this process is inside a different partition "
                                       "context (one process drives one GPU and one partition at a time)")
            _ACTIVE_PROCESS.append((self, me))
        if not hasattr(_ACTIVE, "stack"):
            _ACTIVE.stack = []                              # per THREAD (the context object may be shared by threads)
        _ACTIVE.stack.append(getattr(_ACTIVE, "ctx", None))
        _ACTIVE.ctx = self
        return self

    def __exit__(self, *exc):
        me = threading.get_ident()
        _ACTIVE.ctx = _ACTIVE.stack.pop()
        with _ACTIVE_LOCK:
            for k in range(len(_ACTIVE_PROCESS) - 1, -1, -1):       # the innermost entry of THIS context by THIS thread
                if _ACTIVE_PROCESS[k][0] is self and _ACTIVE_PROCESS[k][1] == me:
                    del _ACTIVE_PROCESS[k]
                    break
        return False


class reentered:
    """``with reentered(ctx):`` makes a captured ``partitioned`` context (or None: no-op) the active one on the calling
    thread -- for code that runs after, or on another thread than, the ``with partitioned(...)`` block that was active
    when it was set up (the recomputation of a checkpointed layer inside the backward pass)."""

    def __init__(self, ctx):
        self.ctx = ctx

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()
        return self.ctx

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


def active_partition():
    ctx = getattr(_ACTIVE, "ctx", None)
    if ctx is None and _ACTIVE_PROCESS:
        with _ACTIVE_LOCK:
            ctx = _ACTIVE_PROCESS[-1][0] if _ACTIVE_PROCESS else None
    return ctx


def partition_aggregate(ctx: "partitioned", x_local: torch.Tensor, aggr: str, **kw) -> torch.Tensor:
    extra = dict(ctx.kw)
    if ctx.local_aggregate is not None:
        extra["local_aggregate"] = ctx.local_aggregate
    return aggregate(x_local, ctx.part, aggr=aggr, group=ctx.group, **extra, **kw)


class _SyncBatchNormTorch(torch.autograd.Function):
    """BatchNorm1d (training) over the rows of ALL ranks in plain torch ops: the CPU / gloo form of what
    node_ops._BatchNormRows does on the device with the same ``BatchSync`` (and the reference the GPU tests compare it
    with).  y = [relu](gamma * xhat + beta) [* factors]."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, num_batches, momentum, eps, relu, factors, sync):
        x64 = x.double()
        s = sync.reduce64(torch.stack([x64.sum(0), (x64 * x64).sum(0)]))
        n = float(sync.total_rows)
        mean = s[0] / n
        var = (s[1] / n - mean * mean).clamp_min(0.0)
        istd = torch.rsqrt(var + eps)
        if running_mean is not None:
            with torch.no_grad():
                running_mean.mul_(1 - momentum).add_(momentum * mean.to(running_mean.dtype))
                running_var.mul_(1 - momentum).add_(momentum * (var * (n / max(n - 1.0, 1.0))).to(running_var.dtype))
                if num_batches is not None:
                    num_batches.add_(1)
        xh = (x64 - mean) * istd
        g = weight.double() if weight is not None else torch.ones_like(mean)
        b = bias.double() if bias is not None else torch.zeros_like(mean)
        pre = xh * g + b
        y = torch.relu(pre) if relu else pre
        if factors is not None:
            y = y * factors.double()
        ctx.save_for_backward(xh, g, istd, pre, factors)
        ctx.cfg = (relu, sync, n, weight is not None, bias is not None)
        return y.to(x.dtype)

    @staticmethod
    def backward(ctx, gy):
        xh, g, istd, pre, factors = ctx.saved_tensors
        relu, sync, n, has_w, has_b = ctx.cfg
        gp = gy.double()
        if factors is not None:
            gp = gp * factors.double()
        if relu:
            gp = gp * (pre > 0)
        s = sync.reduce64(torch.stack([gp.sum(0), (gp * xh).sum(0)]))
        dx = g * istd * (gp - s[0] / n - xh * (s[1] / n))
        # dgamma / dbeta: the GLOBAL sums (identical on every rank); allreduce_gradients averages replicated-parameter
        # gradients by summing per-rank parts, so hand back this rank's share
        gw = (gp * xh).sum(0) if has_w else None
        gb = gp.sum(0) if has_b else None
        return (dx.to(gy.dtype), None if gw is None else gw.to(gy.dtype), None if gb is None else gb.to(gy.dtype),
                None, None, None, None, None, None, None, None)


def allreduce_gradients(module: torch.nn.Module, group=None) -> None:
    """Sum the gradients of the (replicated) parameters over the ranks -- every rank differentiated the loss terms of
    its own rows.  One flat bucket, one collective."""
    buckets = {}
    for p in module.parameters():
        if p.grad is not None:
            buckets.setdefault((p.grad.dtype, p.grad.device), []).append(p)
    for ps in buckets.values():                      # one flat bucket per dtype (normally exactly one)
        flat = torch.cat([p.grad.reshape(-1) for p in ps])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        off = 0
        for p in ps:
            n = p.grad.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n
