"""CPU, world_size 2 over gloo: BASELINE config 4 as a MODEL on a node partition -- the restated
DeeperGCN (GENConv + norm_layer('batch') 'res+' stack, examples/ogb/ogbn_products/model.py shape) run UNCHANGED inside
``dist.partitioned(part)``: every rank holds its rows, the aggregation goes through the partition's exchange, BatchNorm
takes its statistics over all ranks' rows (2 x C all-reduce forward and backward), parameter gradients are summed.
Must equal the single-process run: log-probabilities, every parameter gradient, BatchNorm running statistics.
The local aggregation is the oracle (tests may inject it; the product default is the HIP op)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import arch_restated
from deep_gcns_torch_amd import synth
from test_dist_gloo import _free_port, _oracle_local, _pack, _retry_rendezvous, _unpack

N, CIN, HID, NCLS, LAYERS = 300, 12, 16, 5, 4


def _inputs():
    g = torch.Generator().manual_seed(7)
    ei = synth.tricky_graph(n=N, e=5000, hub_deg=2100, seed=3)
    x = torch.randn(N, CIN, generator=g, dtype=torch.float64)
    y = torch.randint(0, NCLS, (N,), generator=g)
    return ei, x, y


def _model(norm, mlp_layers, fused):
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()
    torch.manual_seed(11)
    m = arch_restated.DeeperGCN(num_layers=LAYERS, in_channels=CIN, hidden=HID, num_tasks=NCLS, aggr="softmax_sg", t=0.5,
                                norm=norm, mlp_layers=mlp_layers, fused_layers=fused)
    return m.double().train()


def _oracle_propagate(self, edge_index, size=None, x=None, edge_attr=None, add_root=False, edge_encoder=None):
    from oracle import sparse_ref
    m = sparse_ref.gen_propagate(x, edge_index, edge_attr, aggr=self.aggr, t=getattr(self, "t", 1.0))
    return x + m if add_root else m


def _worker(rank, world, port, scheme, norm, mlp_layers, fused, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from deep_gcns_torch_amd import dist as ddist
        ei, x, y = _inputs()
        m = _model(norm, mlp_layers, fused)
        part = ddist.build_partition(ei, N, HID, rank, world, scheme=scheme)
        xl, yl = x[part.lo:part.hi], y[part.lo:part.hi]
        with ddist.partitioned(part, local_aggregate=_oracle_local):
            out = m(xl, ei)
            loss = torch.nn.functional.nll_loss(out, yl, reduction="sum") / N
            loss.backward()
        ddist.allreduce_gradients(m)
        grads = {k: p.grad for k, p in m.named_parameters()}
        bufs = {k: b for k, b in m.named_buffers() if "running" in k}
        q.put((rank, part.bounds, _pack(out.detach()), {k: _pack(v) for k, v in grads.items()},
               {k: _pack(v) for k, v in bufs.items()}))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("scheme,norm,mlp_layers,fused", [("allgather", "batch", 1, False), ("halo", "batch", 2, False),
                                                          ("allgather", "batch", 2, True), ("allgather", "layer", 1, False)])
@_retry_rendezvous()
def test_partitioned_deepergcn_equals_single_process(scheme, norm, mlp_layers, fused):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, scheme, norm, mlp_layers, fused, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process, same model, plain (unpartitioned) oracle aggregation
    ei, x, y = _inputs()
    m = _model(norm, mlp_layers, False)
    from gcn_lib.sparse import torch_message
    saved = torch_message.GenMessagePassing.propagate
    torch_message.GenMessagePassing.propagate = _oracle_propagate
    try:
        ref = m(x, ei)
        torch.nn.functional.nll_loss(ref, y, reduction="sum").div(N).backward()
    finally:
        torch_message.GenMessagePassing.propagate = saved
    out = torch.cat([_unpack(r[2]) for r in res])
    torch.testing.assert_close(out, ref.detach(), rtol=1e-9, atol=1e-10)
    for k, p in m.named_parameters():
        for r in res:                                    # identical on every rank after the all-reduce
            torch.testing.assert_close(_unpack(r[3][k]), p.grad, rtol=1e-8, atol=1e-10, msg=k)
    for k, b in m.named_buffers():
        if "running" in k:
            for r in res:
                torch.testing.assert_close(_unpack(r[4][k]), b, rtol=1e-9, atol=1e-11, msg=k)
