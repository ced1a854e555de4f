"""CPU, world_size=2 over gloo: the destination-partitioned path (partition bounds, padded-layout
column remap, all-gather forward, reduce-scatter backward).  The local aggregation is injected
from the ORACLE here (tests may do that; the product default is the HIP op)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from deep_gcns_torch_amd import synth
from deep_gcns_torch_amd.dist import (HaloGraph, PartitionedGraph, TransposedGraph, balanced_bounds, halo_gen_aggregate,
                                      partitioned_gen_aggregate,
                                      transposed_gen_aggregate, transposed_supported)


def _pack(obj):
    """Tensors cross the queue BY VALUE (numpy), not through torch's file-descriptor sharing: a rank that exits before
    the parent unpickles its result would otherwise take the shared storage's socket with it (flaky FileNotFoundError)."""
    if isinstance(obj, torch.Tensor):
        return ("__tensor__", obj.detach().cpu().numpy())
    if isinstance(obj, tuple):
        return tuple(_pack(o) for o in obj)
    return obj


def _unpack(obj):
    if isinstance(obj, tuple):
        if len(obj) == 2 and isinstance(obj[0], str) and obj[0] == "__tensor__":
            return torch.from_numpy(obj[1])
        return tuple(_unpack(o) for o in obj)
    return obj



def _retry_rendezvous(times=3):
    """Multi-process tests rendezvous on a freshly picked local port; on a busy host that can lose a race (port
    taken between probing and binding, slow spawn).  Re-run with a new port before reporting a failure."""
    import functools

    def deco(fn):
        @functools.wraps(fn)
        def wrapper(*a, **kw):
            last = None
            for _ in range(times):
                try:
                    return fn(*a, **kw)
                except Exception as exc:      # noqa: BLE001 -- re-raised below if it persists
                    last = exc
                    for child in mp.active_children():      # ranks of the failed attempt must not linger
                        child.terminate()
                        child.join(timeout=10)
            raise last
        return wrapper
    return deco


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_local(x_full, graph, aggr="softmax", **kw):
    from oracle import sparse_ref
    deg = (graph.rowptr[1:] - graph.rowptr[:-1]).long()
    dst = torch.repeat_interleave(torch.arange(graph.n_dst), deg)
    ei = torch.stack([graph.col.long(), dst])
    if kw.get("edge_attr") is not None and graph.eperm is not None:      # CSR position -> original local edge
        kw = dict(kw, edge_attr=kw["edge_attr"].index_select(0, graph.eperm.long()))
    return sparse_ref.gen_propagate(x_full, ei, aggr=aggr, dim_size=graph.n_dst, **kw)


def _worker(rank, world, port, aggr, kw, q, chunks=1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        n, C = 257, 16
        ei = synth.tricky_graph()
        g = torch.Generator().manual_seed(5)
        x = torch.randn(n, C, generator=g, dtype=torch.float64)
        probe = torch.randn(n, C, generator=g, dtype=torch.float64)
        part = PartitionedGraph.from_edge_index(ei, n, rank, world)
        xl = x[part.lo:part.hi].clone().requires_grad_(True)
        out = partitioned_gen_aggregate(xl, part, aggr=aggr, local_aggregate=_oracle_local, pipeline_chunks=chunks, **kw)
        (out * probe[part.lo:part.hi]).sum().backward()
        q.put(_pack((rank, part.bounds, out.detach(), xl.grad.detach(), part.n_local_edges)))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("aggr,kw,chunks", [("softmax", dict(t=0.7), 1), ("power", dict(p=2.0), 1), ("mean", {}, 1),
                                            ("softmax", dict(t=0.7), 4), ("max", {}, 3)])
@_retry_rendezvous()
def test_partitioned_aggregate_world2_matches_single_process(aggr, kw, chunks):
    from oracle import sparse_ref
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, aggr, kw, q, chunks)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([_unpack(q.get(timeout=120)) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n, C = 257, 16
    ei = synth.tricky_graph()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, C, generator=g, dtype=torch.float64).requires_grad_(True)
    probe = torch.randn(n, C, generator=g, dtype=torch.float64)
    ref = sparse_ref.gen_propagate(x, ei, aggr=aggr, **kw)
    (ref * probe).sum().backward()
    out = torch.cat([r[2] for r in res])
    grad = torch.cat([r[3] for r in res])
    assert res[0][1] == res[1][1] and res[0][1][0] == 0 and res[0][1][-1] == n
    assert sum(r[4] for r in res) == ei.size(1)
    torch.testing.assert_close(out, ref.detach(), rtol=1e-10, atol=1e-12)
    torch.testing.assert_close(grad, x.grad, rtol=1e-10, atol=1e-12)


def _worker_params(rank, world, port, case, q, chunks):
    """learnable t / edge features through partitioned_gen_aggregate with pipeline_chunks > 1 (the
    channel-pipelined Function used to drop their gradients; such calls now take the plain composition)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        n, C = 257, 16
        ei = synth.tricky_graph()
        g = torch.Generator().manual_seed(5)
        x = torch.randn(n, C, generator=g, dtype=torch.float64)
        probe = torch.randn(n, C, generator=g, dtype=torch.float64)
        ea = torch.randn(ei.size(1), C, generator=g, dtype=torch.float64)
        part = PartitionedGraph.from_edge_index(ei, n, rank, world)
        xl = x[part.lo:part.hi].clone().requires_grad_(case != "t_only")
        t = torch.tensor([0.7], dtype=torch.float64, requires_grad=True)
        kw = dict(t=t, learn_t=True)
        ea_l = None
        if case == "edge_attr":
            ea_l = ea[part.edge_mask].clone().requires_grad_(True)
            kw["edge_attr"] = ea_l
        out = partitioned_gen_aggregate(xl, part, aggr="softmax", local_aggregate=_oracle_local,
                                        pipeline_chunks=chunks, **kw)
        (out * probe[part.lo:part.hi]).sum().backward()
        gt = t.grad.clone()
        dist.all_reduce(gt)                     # replicated parameter: per-rank gradients are summed by the caller
        q.put(_pack((rank, out.detach(), None if xl.grad is None else xl.grad.detach(), gt,
               None if ea_l is None else (part.edge_mask.nonzero().flatten(), ea_l.grad.detach()))))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["learn_t", "t_only", "edge_attr"])
@_retry_rendezvous()
def test_partitioned_aggregate_keeps_parameter_and_edge_gradients(case):
    from oracle import sparse_ref
    world, chunks = 2, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_params, args=(r, world, port, case, q, chunks)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([_unpack(q.get(timeout=120)) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n, C = 257, 16
    ei = synth.tricky_graph()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, C, generator=g, dtype=torch.float64).requires_grad_(True)
    probe = torch.randn(n, C, generator=g, dtype=torch.float64)
    ea = torch.randn(ei.size(1), C, generator=g, dtype=torch.float64).requires_grad_(True)
    t = torch.tensor([0.7], dtype=torch.float64, requires_grad=True)
    ref = sparse_ref.gen_propagate(x, ei, ea if case == "edge_attr" else None, aggr="softmax", t=t, learn_t=True)
    (ref * probe).sum().backward()
    torch.testing.assert_close(torch.cat([r[1] for r in res]), ref.detach(), rtol=1e-10, atol=1e-12)
    torch.testing.assert_close(res[0][3], t.grad, rtol=1e-9, atol=1e-12)
    if case != "t_only":
        torch.testing.assert_close(torch.cat([r[2] for r in res]), x.grad, rtol=1e-10, atol=1e-12)
    if case == "edge_attr":
        gea = torch.zeros_like(ea)
        for r in res:
            gea[r[4][0]] = r[4][1]
        torch.testing.assert_close(gea, ea.grad, rtol=1e-10, atol=1e-12)


def test_balanced_bounds_and_padded_remap():
    ei = synth.tricky_graph()
    deg = torch.bincount(ei[1], minlength=257)
    for world in (1, 2, 4, 8):
        b = balanced_bounds(deg, world)
        assert len(b) == world + 1 and b[0] == 0 and b[-1] == 257 and all(b[i] <= b[i + 1] for i in range(world))
        parts = [PartitionedGraph.from_edge_index(ei, 257, r, world, bounds=b, need_transpose=False) for r in range(world)]
        assert sum(p.n_local_edges for p in parts) == ei.size(1)
        mr = parts[0].max_rows
        assert mr % 4 == 0 and all(p.max_rows == mr for p in parts)
        bt = torch.tensor(b)
        for p in parts:
            # un-remap the padded column ids and compare with the original edge multiset of this range
            col = p.graph.col.long()
            owner, off = col // mr, col % mr
            src = bt[owner] + off
            assert bool((off < (bt[owner + 1] - bt[owner])).all())
            d = torch.repeat_interleave(torch.arange(p.graph.n_dst), (p.graph.rowptr[1:] - p.graph.rowptr[:-1]).long()) + p.lo
            mine = (ei[1] >= p.lo) & (ei[1] < p.hi)
            want = torch.sort(ei[0][mine] * 1000 + ei[1][mine]).values
            got = torch.sort(src * 1000 + d).values
            assert torch.equal(want, got)
    # a uniform graph is split into near-equal edge shares
    u = synth.undirected_random_graph(5000, 40000, seed=1)
    b = balanced_bounds(torch.bincount(u[1], minlength=5000), 8)
    shares = [int(((u[1] >= b[r]) & (u[1] < b[r + 1])).sum()) for r in range(8)]
    assert max(shares) < 1.05 * (u.size(1) / 8)


def _worker_transposed(rank, world, port, aggr, kw, q, chunks, node_groups=1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        n, C = 257, 16
        ei = synth.tricky_graph()
        g = torch.Generator().manual_seed(5)
        x = torch.randn(n, C, generator=g, dtype=torch.float64)
        probe = torch.randn(n, C, generator=g, dtype=torch.float64)
        tg = TransposedGraph.from_edge_index(ei, n, rank, world, node_groups=node_groups)
        if node_groups == 1:
            assert tg.n_edges == ei.size(1)                   # every rank holds the whole graph
        xl = x[tg.lo:tg.hi].clone().requires_grad_(True)
        out = transposed_gen_aggregate(xl, tg, aggr=aggr, local_aggregate=_oracle_local, pipeline_chunks=chunks, **kw)
        (out * probe[tg.lo:tg.hi]).sum().backward()
        q.put(_pack((rank, tg.bounds, out.detach(), xl.grad.detach(), tg.max_rows, tg.n_edges)))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("aggr,kw,chunks", [("softmax", dict(t=0.7), 1), ("softmax", dict(t=0.7), 2),
                                            ("power", dict(p=2.0), 2), ("max", {}, 1), ("mean", {}, 2)])
@_retry_rendezvous()
def test_channel_transposed_aggregate_world2_matches_single_process(aggr, kw, chunks):
    """all_to_all (rows -> channel block) -> aggregation of ALL edges on 8 of the 16 channels -> all_to_all back;
    uneven row ranges (128 / 129 rows) exercise the padded layout."""
    from oracle import sparse_ref
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_transposed, args=(r, world, port, aggr, kw, q, chunks)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([_unpack(q.get(timeout=120)) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n, C = 257, 16
    ei = synth.tricky_graph()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, C, generator=g, dtype=torch.float64).requires_grad_(True)
    probe = torch.randn(n, C, generator=g, dtype=torch.float64)
    ref = sparse_ref.gen_propagate(x, ei, aggr=aggr, **kw)
    (ref * probe).sum().backward()
    assert res[0][1] == [0, 128, 257] and res[0][4] == 129
    torch.testing.assert_close(torch.cat([r[2] for r in res]), ref.detach(), rtol=1e-10, atol=1e-12)
    torch.testing.assert_close(torch.cat([r[3] for r in res]), x.grad, rtol=1e-10, atol=1e-12)


def test_transposed_graph_padded_ids_round_trip():
    ei = synth.tricky_graph()
    n = 257
    for world in (1, 2, 4, 8):
        tgs = [TransposedGraph.from_edge_index(ei, n, r, world, need_transpose=False) for r in range(world)]
        mr = tgs[0].max_rows
        bt = torch.tensor(tgs[0].bounds)
        for tg in tgs:
            assert tg.graph.n_dst == world * mr and tg.graph.n_src == world * mr and tg.n_edges == ei.size(1)
            col = tg.graph.col.long()
            drow = torch.repeat_interleave(torch.arange(tg.graph.n_dst), (tg.graph.rowptr[1:] - tg.graph.rowptr[:-1]).long())
            src = bt[col // mr] + col % mr
            dst = bt[drow // mr] + drow % mr
            assert bool((col % mr < (bt[col // mr + 1] - bt[col // mr])).all())
            assert torch.equal(torch.sort(src * 1000 + dst).values, torch.sort(ei[0] * 1000 + ei[1]).values)
    assert transposed_supported(128, 8) and not transposed_supported(100, 8)
    assert not transposed_supported(128, 8, edge_attr=torch.zeros(1))


@pytest.mark.parametrize("world,node_groups,aggr,kw,chunks", [(4, 2, "softmax", dict(t=0.7), 1), (4, 2, "max", {}, 2),
                                                             (2, 2, "power", dict(p=2.0), 1), (4, 4, "mean", {}, 1)])
@_retry_rendezvous()
def test_two_dimensional_transposed_aggregate_matches_single_process(world, node_groups, aggr, kw, chunks):
    """node groups x channel groups: replicated input all-to-all, group-local output all-to-all (uneven splits),
    and in the backward the sum over node groups.  (4,4) and (2,2) degenerate to pure node partitioning."""
    from oracle import sparse_ref
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_transposed, args=(r, world, port, aggr, kw, q, chunks, node_groups))
             for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([_unpack(q.get(timeout=180)) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n, C = 257, 16
    ei = synth.tricky_graph()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, C, generator=g, dtype=torch.float64).requires_grad_(True)
    probe = torch.randn(n, C, generator=g, dtype=torch.float64)
    ref = sparse_ref.gen_propagate(x, ei, aggr=aggr, **kw)
    (ref * probe).sum().backward()
    wc = world // node_groups
    # each node group holds every edge exactly once (per channel group)
    assert sum(r[5] for r in res) == wc * ei.size(1)
    torch.testing.assert_close(torch.cat([r[2] for r in res]), ref.detach(), rtol=1e-10, atol=1e-12)
    torch.testing.assert_close(torch.cat([r[3] for r in res]), x.grad, rtol=1e-10, atol=1e-12)


def _banded_graph(n=300, seed=3):
    """Mostly-local edges (|src - dst| small) plus a few long ones: a partition's halo is a small part of the graph."""
    g = torch.Generator().manual_seed(seed)
    dst = torch.randint(0, n, (4000,), generator=g)
    off = torch.randint(-6, 7, (4000,), generator=g)
    src = (dst + off).clamp(0, n - 1)
    far = torch.randint(0, n, (2, 60), generator=g)
    return torch.cat([torch.stack([src, dst]), far], dim=1)


def _worker_halo(rank, world, port, aggr, kw, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        n, C = 300, 8
        ei = _banded_graph(n)
        g = torch.Generator().manual_seed(5)
        x = torch.randn(n, C, generator=g, dtype=torch.float64)
        probe = torch.randn(n, C, generator=g, dtype=torch.float64)
        hg = HaloGraph.from_edge_index(ei, n, rank, world)
        xl = x[hg.lo:hg.hi].clone().requires_grad_(True)
        out = halo_gen_aggregate(xl, hg, aggr=aggr, local_aggregate=_oracle_local, **kw)
        (out * probe[hg.lo:hg.hi]).sum().backward()
        q.put(_pack((rank, hg.bounds, out.detach(), xl.grad.detach(), hg.n_halo, hg.n_local, sum(hg.send_counts))))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world,aggr,kw", [(2, "softmax", dict(t=0.7)), (3, "max", {}), (4, "power", dict(p=2.0))])
@_retry_rendezvous()
def test_halo_exchange_matches_single_process(world, aggr, kw):
    """Only referenced remote rows travel (uneven all-to-all), gradients of halo rows return to their owners."""
    from oracle import sparse_ref
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_halo, args=(r, world, port, aggr, kw, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([_unpack(q.get(timeout=180)) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n, C = 300, 8
    ei = _banded_graph(n)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, C, generator=g, dtype=torch.float64).requires_grad_(True)
    probe = torch.randn(n, C, generator=g, dtype=torch.float64)
    ref = sparse_ref.gen_propagate(x, ei, aggr=aggr, **kw)
    (ref * probe).sum().backward()
    torch.testing.assert_close(torch.cat([r[2] for r in res]), ref.detach(), rtol=1e-10, atol=1e-12)
    torch.testing.assert_close(torch.cat([r[3] for r in res]), x.grad, rtol=1e-10, atol=1e-12)
    # the halo is a fraction of the remote rows, and what is sent equals what is received overall
    assert all(r[4] < 0.6 * (n - r[5]) for r in res)
    assert sum(r[4] for r in res) == sum(r[6] for r in res)


# ---- the local-first scheme (dist.SplitGraph) ---------------------------------------------------------
def _state_fwd_torch(x, graph, t):
    """(out, L) of the softmax aggregation over ``graph`` in plain torch (what ops.softmax_state_forward returns)."""
    deg = (graph.rowptr[1:] - graph.rowptr[:-1]).long()
    dst = torch.repeat_interleave(torch.arange(graph.n_dst), deg)
    src = graph.col.long()
    C = x.size(1)
    m = torch.relu(x[src]) + 1e-7
    s = t * m
    mx = torch.full((graph.n_dst, C), float("-inf"), dtype=x.dtype).scatter_reduce(0, dst.unsqueeze(1).expand_as(s), s,
                                                                                    "amax", include_self=True)
    mx0 = torch.where(torch.isinf(mx), torch.zeros_like(mx), mx)
    e = torch.exp(s - mx0[dst])
    den = torch.zeros(graph.n_dst, C, dtype=x.dtype).index_add_(0, dst, e)
    num = torch.zeros(graph.n_dst, C, dtype=x.dtype).index_add_(0, dst, e * m)
    has = (deg > 0).unsqueeze(1)
    out = torch.where(has, num / den.clamp_min(1e-300), torch.zeros_like(num))
    L = torch.where(has, mx0 + torch.log(den.clamp_min(1e-300)), torch.zeros_like(num))
    return out, L


def _state_bwd_torch(x, graph, g, L, t):
    deg = (graph.rowptr[1:] - graph.rowptr[:-1]).long()
    dst = torch.repeat_interleave(torch.arange(graph.n_dst), deg)
    src = graph.col.long()
    z = x[src]
    m = torch.relu(z) + 1e-7
    w = torch.exp(t * m - L[dst])
    dz = w * g[dst] * (z > 0).to(x.dtype)
    return torch.zeros(graph.n_src, x.size(1), dtype=x.dtype).index_add_(0, src, dz)


def _split_worker(rank, world, port, t, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from deep_gcns_torch_amd.dist import SplitGraph, build_partition, aggregate
        torch.set_num_threads(2)
        n, C = 257, 16
        ei = synth.tricky_graph()
        g = torch.Generator().manual_seed(5)
        x = torch.randn(n, C, generator=g, dtype=torch.float64)
        probe = torch.randn(n, C, generator=g, dtype=torch.float64)
        part = build_partition(ei, n, C, rank, world, scheme="split")
        assert isinstance(part, SplitGraph) and part.local.n_edges + part.remote.n_edges == part.n_local_edges
        xl = x[part.lo:part.hi].clone().requires_grad_(True)
        out = aggregate(xl, part, aggr="softmax_sg", t=t, state_fns=(_state_fwd_torch, _state_bwd_torch))
        (out * probe[part.lo:part.hi]).sum().backward()
        q.put(_pack((rank, part.bounds, out.detach(), xl.grad.detach(), part.local.n_edges)))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world,t", [(2, 0.7), (3, 0.1)])
@_retry_rendezvous()
def test_local_first_split_scheme_matches_single_process(world, t):
    """dist.SplitGraph: the edges of a destination partition cut by the owner of the source, the local part aggregated
    while the all-gather is in flight, the two partial softmax states merged; backward: the remote gradient's
    reduce-scatter in flight during the local gradient kernel.  Same outputs and gradients as the oracle on the whole
    graph (the tricky graph: a hub row whose in-edges come from every rank, rows without local or without remote edges,
    isolated rows)."""
    from oracle import sparse_ref
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_split_worker, args=(r, world, port, t, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([_unpack(q.get(timeout=120)) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n, C = 257, 16
    ei = synth.tricky_graph()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, C, generator=g, dtype=torch.float64).requires_grad_(True)
    probe = torch.randn(n, C, generator=g, dtype=torch.float64)
    ref = sparse_ref.gen_propagate(x, ei, aggr="softmax_sg", t=t)
    (ref * probe).sum().backward()
    bounds = res[0][1]
    assert sum(r[4] for r in res) > 0                                     # some edges are local-source
    for rank, b, out, gx, _ in res:
        lo, hi = bounds[rank], bounds[rank + 1]
        torch.testing.assert_close(out, ref[lo:hi].detach(), rtol=1e-10, atol=1e-12)
        torch.testing.assert_close(gx, x.grad[lo:hi], rtol=1e-10, atol=1e-12)


# ---- local-first scheme for the aggregators with an associative partial state (round 6) --------------------------------
def _split_composed_worker(rank, world, port, aggr, kw, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from deep_gcns_torch_amd.dist import SplitGraph, build_partition, aggregate, split_supported, exchange_bytes, choose_scheme
        torch.set_num_threads(2)
        n, C = 257, 16
        ei = synth.tricky_graph()
        g = torch.Generator().manual_seed(6)
        x = torch.randn(n, C, generator=g, dtype=torch.float64)
        probe = torch.randn(n, C, generator=g, dtype=torch.float64)
        part = build_partition(ei, n, C, rank, world, scheme="split")
        assert isinstance(part, SplitGraph) and split_supported(aggr, kw)
        assert not split_supported(aggr, dict(kw, add_root=True)) and not split_supported(aggr, dict(kw, bogus=1))
        assert split_supported("power", {"p": 2.0}) and not split_supported("power", {"p": 2.0, "learn_p": True})
        xl = x[part.lo:part.hi].clone().requires_grad_(True)
        pk = dict(kw)
        p_param = None
        if pk.get("learn_p"):
            p_param = torch.tensor([pk["p"]], dtype=torch.float64, requires_grad=True)
            pk["p"] = p_param
        out = aggregate(xl, part, aggr=aggr, local_aggregate=_oracle_local, **pk)
        # (a rank whose rows have no in-edge at all -- the tricky graph has one -- returns constants: keep the loss
        # differentiable there, every rank must enter the backward's collective)
        ((out * probe[part.lo:part.hi]).sum() + 0.0 * xl.sum()).backward()
        gp = (p_param.grad.detach() if p_param is not None and p_param.grad is not None
              else torch.zeros(1, dtype=torch.float64))
        costs = exchange_bytes(ei, n, C, rank, world)
        auto = build_partition(ei, n, C, rank, world, scheme="auto", aggr=aggr)     # every rank must take the same scheme
        q.put(_pack((rank, part.bounds, out.detach(), xl.grad.detach(), gp, costs["halo_rows"], type(auto).__name__,
                     choose_scheme(costs, aggr))))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("aggr,kw", [("max", {}), ("add", {}), ("mean", {}), ("max", {"eps": 1e-3}),
                                     ("mean", {"eps": 1e-3})])
@_retry_rendezvous()
def test_local_first_split_scheme_for_max_add_mean(aggr, kw):
    """The partial aggregations over the local-source and the remote-source edges merged elementwise (dist.
    _split_composed_aggregate): outputs, input gradients and the gradient of a learnable p equal the oracle's on the whole
    graph; keywords the scheme does not handle are refused, not dropped; ``build_partition("auto")`` takes the same
    scheme on every rank."""
    from oracle import sparse_ref
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_split_composed_worker, args=(r, world, port, aggr, kw, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([_unpack(q.get(timeout=120)) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n, C = 257, 16
    ei = synth.tricky_graph()
    g = torch.Generator().manual_seed(6)
    x = torch.randn(n, C, generator=g, dtype=torch.float64).requires_grad_(True)
    probe = torch.randn(n, C, generator=g, dtype=torch.float64)
    rk = dict(kw)
    p_ref = None
    if rk.get("learn_p"):
        p_ref = torch.tensor([rk["p"]], dtype=torch.float64, requires_grad=True)
        rk["p"] = p_ref
    rk.pop("learn_p", None)
    ref = sparse_ref.gen_propagate(x, ei, aggr=aggr, **rk)
    (ref * probe).sum().backward()
    bounds = res[0][1]
    for rank, b, out, gx, gp, halo, auto_cls, chosen in res:
        lo, hi = bounds[rank], bounds[rank + 1]
        torch.testing.assert_close(out, ref[lo:hi].detach(), rtol=1e-9, atol=1e-11)
        torch.testing.assert_close(gx, x.grad[lo:hi], rtol=1e-9, atol=1e-11)
    if p_ref is not None:
        torch.testing.assert_close(sum(r[4] for r in res), p_ref.grad, rtol=1e-8, atol=1e-10)
    assert len({r[6] for r in res}) == 1, "the ranks disagree on the scheme build_partition('auto') took"


def _power_state_fwd_torch(x, graph, p, eps=1e-7):
    """(out, q) of the power-mean aggregation over ``graph`` in plain torch (what ops.power_state_forward returns):
    gcn_lib/sparse/torch_message.py:66-80 with q = the pre-clamp mean."""
    deg = (graph.rowptr[1:] - graph.rowptr[:-1]).long()
    dst = torch.repeat_interleave(torch.arange(graph.n_dst), deg)
    m = (torch.relu(x[graph.col.long()]) + eps).clamp(1e-7, 10.0)
    q = torch.zeros(graph.n_dst, x.size(1), dtype=x.dtype).index_add_(0, dst, m.pow(p)) / deg.clamp_min(1).unsqueeze(1)
    return q.clamp(1e-7, 10.0).pow(1.0 / p), q


def _power_state_bwd_torch(x, graph, coef, q, p, eps=1e-7):
    deg = (graph.rowptr[1:] - graph.rowptr[:-1]).long()
    dst = torch.repeat_interleave(torch.arange(graph.n_dst), deg)
    src = graph.col.long()
    z = x[src]
    m = torch.relu(z) + eps
    inside = ((m >= 1e-7) & (m <= 10.0)).to(x.dtype)
    dz = coef[dst] * m.clamp(1e-7, 10.0).pow(p - 1.0) * inside * (z > 0).to(x.dtype)
    return torch.zeros(graph.n_src, x.size(1), dtype=x.dtype).index_add_(0, src, dz)


def _split_power_worker(rank, world, port, p, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from deep_gcns_torch_amd.dist import build_partition, aggregate
        torch.set_num_threads(2)
        n, C = 257, 16
        ei = synth.tricky_graph()
        g = torch.Generator().manual_seed(7)
        x = torch.randn(n, C, generator=g, dtype=torch.float64)
        probe = torch.randn(n, C, generator=g, dtype=torch.float64)
        part = build_partition(ei, n, C, rank, world, scheme="split")
        xl = x[part.lo:part.hi].clone().requires_grad_(True)
        out = aggregate(xl, part, aggr="power", p=p, state_fns=(_power_state_fwd_torch, _power_state_bwd_torch))
        (out * probe[part.lo:part.hi]).sum().backward()
        q.put(_pack((rank, part.bounds, out.detach(), xl.grad.detach())))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world,p", [(2, 2.0), (3, 1.0), (3, 3.0)])
@_retry_rendezvous()
def test_local_first_split_scheme_for_power_mean(world, p):
    """Power-mean over dist.SplitGraph merges from the two parts' pre-clamp means and degrees (rows whose partial mean sits
    under the reference's clamp -- every message of a part at eps -- included: the tricky graph has them): outputs and
    input gradients equal the oracle's on the whole graph."""
    from oracle import sparse_ref
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_split_power_worker, args=(r, world, port, p, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = sorted([_unpack(q.get(timeout=120)) for _ in range(world)], key=lambda r: r[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    n, C = 257, 16
    ei = synth.tricky_graph()
    g = torch.Generator().manual_seed(7)
    x = torch.randn(n, C, generator=g, dtype=torch.float64).requires_grad_(True)
    probe = torch.randn(n, C, generator=g, dtype=torch.float64)
    ref = sparse_ref.gen_propagate(x, ei, aggr="power", p=p)
    (ref * probe).sum().backward()
    bounds = res[0][1]
    for rank, b, out, gx in res:
        lo, hi = bounds[rank], bounds[rank + 1]
        torch.testing.assert_close(out, ref[lo:hi].detach(), rtol=1e-9, atol=1e-12)
        torch.testing.assert_close(gx, x.grad[lo:hi], rtol=1e-9, atol=1e-12)
