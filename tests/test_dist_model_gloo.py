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


def _model(norm, mlp_layers, fused, layers=LAYERS, checkpoint="reference"):
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()
    torch.manual_seed(11)
    m = arch_restated.DeeperGCN(num_layers=layers, in_channels=CIN, hidden=HID, num_tasks=NCLS, aggr="softmax_sg", t=0.5,
                                norm=norm, mlp_layers=mlp_layers, fused_layers=fused, checkpoint=checkpoint)
    return m.double().train()


def _oracle_propagate(self, edge_index, size=None, x=None, edge_attr=None, add_root=False, edge_encoder=None):
    from oracle import sparse_ref
    m = sparse_ref.gen_propagate(x, edge_index, edge_attr, aggr=self.aggr, t=getattr(self, "t", 1.0))
    return x + m if add_root else m


def _worker(rank, world, port, scheme, norm, mlp_layers, fused, q, layers=LAYERS, checkpoint="reference", bwd="inside"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from deep_gcns_torch_amd import dist as ddist
        ei, x, y = _inputs()
        m = _model(norm, mlp_layers, fused, layers, checkpoint)
        assert m.checkpoint_grad == (layers > 7 and checkpoint != "never")
        part = ddist.build_partition(ei, N, HID, rank, world, scheme=scheme)
        xl, yl = x[part.lo:part.hi], y[part.lo:part.hi]
        with ddist.partitioned(part, local_aggregate=_oracle_local):
            out = m(xl, ei)
            loss = torch.nn.functional.nll_loss(out, yl, reduction="sum") / N
            if bwd == "inside":
                loss.backward()
            elif bwd == "thread":
                # what the device's autograd worker does to the recomputation of a reentrant checkpoint: it runs on a
                # thread that never entered the context
                import threading
                err = []

                def run():
                    try:
                        loss.backward()
                    except BaseException as exc:   # noqa: BLE001 -- re-raised on the main thread
                        err.append(exc)
                th = threading.Thread(target=run)
                th.start()
                th.join()
                if err:
                    raise err[0]
        if bwd == "outside":                 # loss.backward() after the with block: the layers captured the context
            assert ddist.active_partition() is None
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
    _compare_with_single_process(res, norm, mlp_layers)


def _compare_with_single_process(res, norm, mlp_layers, layers=LAYERS, checkpoint="never"):
    # single process, same model, plain (unpartitioned) oracle aggregation (with the reference's checkpointing when the
    # workers used it: a BatchNorm inside a recomputed MLP updates its running statistics twice per step, as in the
    # reference)
    ei, x, y = _inputs()
    m = _model(norm, mlp_layers, False, layers, checkpoint)
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


@pytest.mark.parametrize("fused,checkpoint,bwd,mlp_layers", [
    (True, "reference", "thread", 2),          # res_plus_layer, aggregation results kept, BatchNorm inside the recomputed MLP
    (True, "reference_full", "outside", 2),    # everything recomputed, backward after the context was left
    (True, "reference", "outside", 1),
    (False, "reference", "thread", 2),         # the model file's own checkpoint(self.gcns[layer], h2, edge_index)
])
@_retry_rendezvous()
def test_partitioned_checkpointed_stack_recomputes_inside_the_partition(fused, checkpoint, bwd, mlp_layers):
    """10 layers: the reference's gradient checkpointing is on (num_layers > 7).  The recomputation runs inside the
    backward pass -- on another thread than the one that entered ``dist.partitioned``, or after the block was left --
    and must go through the partition's exchange and the cross-rank BatchNorm sums like the first pass (ADVICE r3:
    a thread-local context made it build a graph from the placeholder edge_index instead)."""
    world, layers = 2, 10
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, "allgather", "batch", mlp_layers, fused, q, layers,
                                               checkpoint, bwd)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    _compare_with_single_process(res, "batch", mlp_layers, layers, "reference")
