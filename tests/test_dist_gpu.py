"""GPU: the RCCL code path of the destination-partitioned aggregation with a single rank (the box has one
GPU): all_gather_into_tensor / reduce_scatter_tensor on the "nccl" backend, padded layout, HIP local kernel.
Multi-rank behaviour is covered on CPU by tests/test_dist_gloo.py."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_partitioned_aggregate_on_rccl_single_rank():
    import torch.distributed as dist
    from deep_gcns_torch_amd import ops, synth
    from deep_gcns_torch_amd.dist import PartitionedGraph, partitioned_gen_aggregate
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        created = True
    try:
        ei = synth.tricky_graph().to(dev)
        g = torch.Generator().manual_seed(2)
        x = torch.randn(257, 64, generator=g).to(dev)
        probe = torch.randn(257, 64, generator=g).to(dev)
        part = PartitionedGraph.from_edge_index(ei, 257, 0, 1)
        assert part.max_rows % 4 == 0 and part.max_rows >= 257          # padded layout is exercised
        for aggr, kw in (("softmax_sg", dict(t=0.1)), ("max", {}), ("power", dict(p=2.0))):
            xa = x.clone().requires_grad_(True)
            out = partitioned_gen_aggregate(xa, part, aggr=aggr, **kw)
            (out * probe).sum().backward()
            xb = x.clone().requires_grad_(True)
            ref = ops.gen_aggregate(xb, ei, aggr=aggr, **kw)
            (ref * probe).sum().backward()
            torch.testing.assert_close(out, ref, rtol=1e-6, atol=1e-7)
            torch.testing.assert_close(xa.grad, xb.grad, rtol=1e-5, atol=1e-6)
            # channel-pipelined form (async block collectives on RCCL, HIP kernel per 16-channel block)
            from deep_gcns_torch_amd.dist import _PipelinedPartitionedAggregate
            xc = x.clone().requires_grad_(True)
            outp = _PipelinedPartitionedAggregate.apply(xc, part, None, ops.gen_aggregate, aggr, kw, 4)
            (outp * probe).sum().backward()
            torch.testing.assert_close(outp, ref, rtol=1e-6, atol=1e-7)
            torch.testing.assert_close(xc.grad, xb.grad, rtol=1e-5, atol=1e-6)
            # channel-transposed scheme: all_to_all_single on RCCL, whole graph in the padded layout, 2 sub-blocks
            from deep_gcns_torch_amd.dist import TransposedGraph, transposed_gen_aggregate
            tg = TransposedGraph.from_edge_index(ei, 257, 0, 1)
            xd = x.clone().requires_grad_(True)
            outt = transposed_gen_aggregate(xd, tg, aggr=aggr, pipeline_chunks=2, **kw)
            (outt * probe).sum().backward()
            torch.testing.assert_close(outt, ref, rtol=1e-6, atol=1e-7)
            torch.testing.assert_close(xd.grad, xb.grad, rtol=1e-5, atol=1e-6)
            # halo scheme: with one rank the halo is empty (zero-size all_to_all on RCCL), the local kernel sees its rows
            from deep_gcns_torch_amd.dist import HaloGraph, halo_gen_aggregate
            hg = HaloGraph.from_edge_index(ei, 257, 0, 1)
            assert hg.n_halo == 0 and hg.graph.n_src == 257
            xe = x.clone().requires_grad_(True)
            outh = halo_gen_aggregate(xe, hg, aggr=aggr, **kw)
            (outh * probe).sum().backward()
            torch.testing.assert_close(outh, ref, rtol=1e-6, atol=1e-7)
            torch.testing.assert_close(xe.grad, xb.grad, rtol=1e-5, atol=1e-6)
        # the per-phase timing helper bench.py prints for multi-rank runs (exchange vs local kernels), every scheme
        from deep_gcns_torch_amd.dist import phase_times
        for p_ in (part, tg, hg):
            ph = phase_times(x, probe, p_, aggr="softmax_sg", t=0.1, reps=2)
            assert ph["step"] > 0 and ph["kernels"] > 0 and ph["local_edges"] == ei.size(1)
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("world,rank,node_groups", [(4, 2, 1), (4, 1, 2), (8, 5, 2), (8, 7, 1)])
def test_rank_local_rectangular_graphs_of_a_multi_rank_job(world, rank, node_groups):
    """The local graphs a rank of a W > 1 job hands to the HIP kernels (built without any collective): the
    destination-partitioned one (n_dst = local rows, n_src = W * max_rows padded) and the transposed one (whole graph or
    one node group, padded ids).  Forward and backward against the CPU oracle on the same rectangular graph."""
    from deep_gcns_torch_amd import ops, synth
    from deep_gcns_torch_amd.dist import PartitionedGraph, TransposedGraph
    from oracle import sparse_ref
    dev = torch.device("cuda:0")
    n, C = 3001, 32
    ei = synth.powerlaw_graph(n, 20_000, seed=21, exponent=2.2)
    parts = [PartitionedGraph.from_edge_index(ei.to(dev), n, rank, world),
             TransposedGraph.from_edge_index(ei.to(dev), n, rank, world, node_groups=node_groups)]
    for part in parts:
        g = part.graph
        gen = torch.Generator().manual_seed(world * 10 + rank)
        x = torch.randn(g.n_src, C, generator=gen)
        probe = torch.randn(g.n_dst, C, generator=gen)
        deg = (g.rowptr[1:] - g.rowptr[:-1]).long().cpu()
        eic = torch.stack([g.col.long().cpu(), torch.repeat_interleave(torch.arange(g.n_dst), deg)])
        for aggr, kw in (("softmax_sg", dict(t=0.5)), ("max", {}), ("power", dict(p=2.0))):
            xr = x.double().requires_grad_(True)
            ref = sparse_ref.gen_propagate(xr, eic, aggr=aggr, dim_size=g.n_dst, **kw)
            (ref * probe.double()).sum().backward()
            xd = x.to(dev).requires_grad_(True)
            out = ops.gen_aggregate(xd, g, aggr=aggr, **kw)
            (out * probe.to(dev)).sum().backward()
            torch.testing.assert_close(out.detach().cpu().double(), ref.detach(), rtol=1e-4, atol=1e-6)
            gs = max(1.0, float(xr.grad.abs().max()))
            torch.testing.assert_close(xd.grad.cpu().double(), xr.grad, rtol=1e-4, atol=2e-6 * gs)


@pytest.mark.parametrize("world,rank", [(2, 1), (4, 2), (8, 5)])
def test_local_first_split_states_on_the_hip_kernels(world, rank):
    """dist.SplitGraph on the device without a collective: the rank's local-source and remote-source aggregations
    (ops.softmax_state_forward), merged, equal the one-launch aggregation of the all-gather scheme's rectangular graph for
    the same rank; the two gradient launches (ops.softmax_state_backward with the MERGED log-sum-exp) add up to its
    gradient.  (The exchange itself: tests/test_dist_gloo.py::test_local_first_split_scheme_matches_single_process.)"""
    from deep_gcns_torch_amd import ops, synth
    from deep_gcns_torch_amd.dist import PartitionedGraph, SplitGraph, merge_softmax_states
    dev = torch.device("cuda:0")
    n, C, t = 3001, 64, 0.3
    ei = synth.powerlaw_graph(n, 20_000, seed=21, exponent=2.2).to(dev)
    part = PartitionedGraph.from_edge_index(ei, n, rank, world)
    sg = SplitGraph.from_edge_index(ei, n, rank, world)
    assert sg.bounds == part.bounds and sg.local.n_edges + sg.remote.n_edges == part.graph.n_edges
    assert sg.local.n_edges > 0 and sg.remote.n_edges > 0
    gen = torch.Generator(device=dev).manual_seed(3)
    x_full = torch.randn(world * part.max_rows, C, device=dev, generator=gen)
    probe = torch.randn(part.n_local, C, device=dev, generator=gen)
    lo_p = rank * part.max_rows
    xf = x_full.clone().requires_grad_(True)
    ref = ops.gen_aggregate(xf, part.graph, aggr="softmax_sg", t=t)
    (ref * probe).sum().backward()
    x_loc = x_full[lo_p:lo_p + part.n_local].contiguous()
    oa, la = ops.softmax_state_forward(x_loc, sg.local, t)
    ob, lb = ops.softmax_state_forward(x_full, sg.remote, t)
    out, L = merge_softmax_states(oa, la, (sg.local.deg > 0).unsqueeze(1), ob, lb, (sg.remote.deg > 0).unsqueeze(1))
    torch.testing.assert_close(out, ref.detach(), rtol=1e-5, atol=1e-6)
    # the merge as ONE HIP launch (dgcn_softmax_state_merge_f32, in place in the local state) against the torch formula,
    # rows with edges on one side only included (the power-law graph has them on every rank)
    one_sided = int(((sg.local.deg > 0) ^ (sg.remote.deg > 0)).sum())
    assert one_sided > 0
    out_k, L_k = ops.softmax_state_merge(oa.clone(), la.clone(), sg.local, ob, lb, sg.remote)
    torch.testing.assert_close(out_k, out, rtol=2e-6, atol=1e-6)
    torch.testing.assert_close(L_k, L, rtol=2e-6, atol=1e-6)
    gs = max(1.0, float(xf.grad.abs().max()))
    for prep in (None, ops.softmax_state_prepare(probe, L)):        # two-gather form, single-gather form
        g_rem = ops.softmax_state_backward(x_full, sg.remote, probe, L, t, prep=prep)
        g_loc = ops.softmax_state_backward(x_loc, sg.local, probe, L, t, prep=prep)
        total = g_rem.clone()
        total[lo_p:lo_p + part.n_local] += g_loc
        torch.testing.assert_close(total, xf.grad, rtol=1e-4, atol=2e-6 * gs)


@pytest.mark.parametrize("world,rank,p", [(2, 1, 2.0), (4, 2, 1.0), (8, 5, 3.0)])
def test_local_first_power_states_on_the_hip_kernels(world, rank, p):
    """Power-mean over dist.SplitGraph on the device without a collective: the two parts' pre-clamp means
    (ops.power_state_forward) merged with the degrees equal the one-launch aggregation of the all-gather scheme's
    rectangular graph for the same rank, and the two gradient launches (ops.power_state_backward with the coefficient
    formed from the MERGED mean) add up to its gradient."""
    from deep_gcns_torch_amd import ops, synth
    from deep_gcns_torch_amd.dist import PartitionedGraph, SplitGraph
    dev = torch.device("cuda:0")
    n, C = 3001, 64
    ei = synth.powerlaw_graph(n, 20_000, seed=21, exponent=2.2).to(dev)
    part = PartitionedGraph.from_edge_index(ei, n, rank, world)
    sg = SplitGraph.from_edge_index(ei, n, rank, world)
    gen = torch.Generator(device=dev).manual_seed(4)
    x_full = torch.randn(world * part.max_rows, C, device=dev, generator=gen)
    probe = torch.randn(part.n_local, C, device=dev, generator=gen)
    lo_p = rank * part.max_rows
    xf = x_full.clone().requires_grad_(True)
    ref = ops.gen_aggregate(xf, part.graph, aggr="power", p=p)
    (ref * probe).sum().backward()
    x_loc = x_full[lo_p:lo_p + part.n_local].contiguous()
    _, qa = ops.power_state_forward(x_loc, sg.local, p)
    _, qb = ops.power_state_forward(x_full, sg.remote, p)
    da, db = sg.local.deg.unsqueeze(1), sg.remote.deg.unsqueeze(1)
    deg = (da + db).clamp_min(1.0)
    q = (qa * da + qb * db) / deg
    r = q.clamp(ops.POW_LO, ops.POW_HI)
    torch.testing.assert_close(r.pow(1.0 / p), ref.detach(), rtol=2e-5, atol=1e-6)
    coef = probe * r.pow(1.0 / p - 1.0) * ((q >= ops.POW_LO) & (q <= ops.POW_HI)).float() / deg
    total = ops.power_state_backward(x_full, sg.remote, coef, q, p)
    total[lo_p:lo_p + part.n_local] += ops.power_state_backward(x_loc, sg.local, coef, q, p)
    gs = max(1.0, float(xf.grad.abs().max()))
    torch.testing.assert_close(total, xf.grad, rtol=1e-4, atol=2e-6 * gs)


def _hip_local(x_full, graph, aggr="softmax", **kw):
    """local_aggregate for a gloo (CPU tensor) job whose per-rank compute runs on the GPU's HIP kernels."""
    from deep_gcns_torch_amd import ops
    from deep_gcns_torch_amd.graph import Graph
    dev = torch.device("cuda:0")
    deg = (graph.rowptr[1:] - graph.rowptr[:-1]).long()
    dst = torch.repeat_interleave(torch.arange(graph.n_dst), deg)
    gd = Graph(graph.col.long().to(dev), dst.to(dev), n_src=graph.n_src, n_dst=graph.n_dst)
    return ops.gen_aggregate(x_full.float().to(dev), gd, aggr=aggr, **kw).cpu().to(x_full.dtype)


def _worker_hybrid(rank, world, port, q):
    import torch.distributed as dist
    from deep_gcns_torch_amd import dist as ddist, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, C = 2000, 32
        ei = synth.powerlaw_graph(n, 15_000, seed=31, exponent=2.2)
        g = torch.Generator().manual_seed(6)
        x = torch.randn(n, C, generator=g)
        probe = torch.randn(n, C, generator=g)
        res = {}
        for name, build in (("allgather", lambda: ddist.PartitionedGraph.from_edge_index(ei, n, rank, world)),
                            ("transposed2d", lambda: ddist.TransposedGraph.from_edge_index(ei, n, rank, world, node_groups=2)),
                            ("halo", lambda: ddist.HaloGraph.from_edge_index(ei, n, rank, world))):
            part = build()
            xl = x[part.lo:part.hi].clone().requires_grad_(True)
            out = ddist.aggregate(xl, part, aggr="softmax_sg", t=0.5, local_aggregate=_hip_local)
            (out * probe[part.lo:part.hi]).sum().backward()
            res[name] = (part.lo, out.detach().numpy().copy(), xl.grad.detach().numpy().copy())   # plain arrays: no fd passing
        q.put((rank, res))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_four_rank_job_with_hip_local_kernels_matches_single_gpu():
    """End to end with W = 4: four processes exchange over gloo (CPU tensors) while every rank's local aggregation runs
    on the GPU's HIP kernels (all ranks share cuda:0) -- all-gather (channel-pipelined), 2-D transposed and halo schemes
    against the single-process HIP result."""
    import socket
    import torch.multiprocessing as mp
    from deep_gcns_torch_amd import ops, synth
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_hybrid, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        got = sorted([q.get(timeout=600) for _ in range(world)], key=lambda r: r[0])
    finally:
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.terminate()
    dev = torch.device("cuda:0")
    n, C = 2000, 32
    ei = synth.powerlaw_graph(n, 15_000, seed=31, exponent=2.2).to(dev)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(n, C, generator=g).to(dev).requires_grad_(True)
    probe = torch.randn(n, C, generator=g).to(dev)
    ref = ops.gen_aggregate(x, ei, aggr="softmax_sg", t=0.5)
    (ref * probe).sum().backward()
    for name in ("allgather", "transposed2d", "halo"):
        order = sorted(got, key=lambda r: r[1][name][0])
        out = torch.cat([torch.from_numpy(r[1][name][1]) for r in order])
        grad = torch.cat([torch.from_numpy(r[1][name][2]) for r in order])
        torch.testing.assert_close(out, ref.detach().cpu(), rtol=1e-5, atol=1e-6, msg=lambda m: f"{name}: {m}")
        gs = float(x.grad.abs().max())
        torch.testing.assert_close(grad, x.grad.cpu(), rtol=1e-4, atol=1e-5 * gs, msg=lambda m: f"{name}: {m}")


def test_dense_layer_is_hip_graph_capturable():
    """No hidden host synchronisation or allocation outside torch's allocator: a whole ResDynBlock2d
    forward+backward (kNN, MFMA GEMM, edge kernels, BatchNorm kernels) captures into a HIP graph and replays."""
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()
    from gcn_lib.dense import ResDynBlock2d
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    blk = ResDynBlock2d(32, 8, 2, "edge", "relu", "batch", True).to(dev).train()
    x = torch.randn(2, 32, 512, 1, device=dev, requires_grad=True)
    go = torch.randn(2, 32, 512, 1, device=dev)
    params = [x] + list(blk.parameters())
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):                                               # warm-up outside capture
            torch.autograd.grad(blk(x), params, go)
    torch.cuda.current_stream().wait_stream(s)
    eager = [g.clone() for g in torch.autograd.grad(blk(x), params, go)]
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        static = torch.autograd.grad(blk(x), params, go)
    graph.replay()
    torch.cuda.synchronize()
    for a, b in zip(static, eager):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)
