"""``deep_gcns_torch_amd.graphs.GraphedStep``: a whole training step of a model built from this package's modules -- the
reversible RevGCN with composed per-edge encoders -- captured as one hipGraph leaves the same parameters as the eager
steps (every libdgcn entry point is asynchronous on the caller's stream and neither allocates nor reads back)."""
import copy

import pytest
import torch

import rev_restated
from deep_gcns_torch_amd import synth

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("identity_dropout_mask")]


def test_graphed_training_step_equals_eager_steps():
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()
    from deep_gcns_torch_amd import ops
    from deep_gcns_torch_amd.graphs import GraphedStep
    dev = torch.device("cuda:0")
    # Adam divides by sqrt(v): last-bit differences of a near-zero gradient become O(lr) differences of the parameter,
    # so the comparison runs with the bit-reproducible item schedule (ops.ENC_STATIC_ITEMS; the default schedule groups
    # the encoder's dW | db partial sums differently from run to run)
    saved_static = ops.ENC_STATIC_ITEMS
    ops.ENC_STATIC_ITEMS = True
    n = 2000
    ei = synth.powerlaw_graph(n, 12_000, seed=9).to(dev)
    g = torch.Generator().manual_seed(2)
    table = torch.rand(n, 8, generator=g).to(dev)
    x = torch.rand(n, 8, generator=g).to(dev)
    ea = torch.rand(ei.size(1), 8, generator=g).to(dev)
    nidx = torch.arange(n, device=dev)
    y = (torch.rand(n, 16, generator=g) > 0.5).float().to(dev)
    torch.manual_seed(4)
    base = rev_restated.RevGCN(num_layers=3, hidden=128, num_tasks=16, aggr="max", dropout=0.0, node_table=table,
                               impl="product", composed_edges=True).to(dev).train()

    def make(model):
        opt = torch.optim.Adam(model.parameters(), lr=1e-2, capturable=True)

        def step():
            opt.zero_grad(set_to_none=True)
            pred, _ = model(x, nidx, ei, ea)
            torch.nn.functional.binary_cross_entropy_with_logits(pred, y).backward()
            opt.step()
        return step

    eager, graphed_model = copy.deepcopy(base), copy.deepcopy(base)
    eager.node_features = graphed_model.node_features = table
    step_e = make(eager)
    for _ in range(3 + 4):                          # 3 warm-up steps + 4 replays on the other side
        step_e()
    graphed = GraphedStep(make(graphed_model), warmup=3)      # 3 eager steps; the capture records, it does not run
    for _ in range(4):
        graphed()
    torch.cuda.synchronize()
    ops.ENC_STATIC_ITEMS = saved_static
    for (k, a), (_, b) in zip(graphed_model.named_parameters(), eager.named_parameters()):
        # deterministic kernels, the same arithmetic in the same order: equal to the last bit or two
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6, msg=lambda m, k=k: f"{k}: {m}")


def test_graphed_forward_backward_with_the_dynamic_item_schedule():
    """The default (dynamic, device-side counters) item schedule of the per-edge encoder kernels inside a replayed graph:
    the counters are re-armed by a kernel of every launch (a memset node was not reliable, see the test below), so a replay
    must do the same work as an eager pass.
    Forward outputs are schedule-independent (bit-equal); the weight gradients' partial sums are grouped by the schedule
    (rounding-level differences).  No optimizer here: Adam would turn those last bits into O(lr) differences."""
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()
    from deep_gcns_torch_amd import ops
    from deep_gcns_torch_amd.graphs import GraphedStep
    assert not ops.ENC_STATIC_ITEMS
    dev = torch.device("cuda:0")
    n = 2500
    ei = synth.powerlaw_graph(n, 20_000, seed=5).to(dev)
    g = torch.Generator().manual_seed(7)
    table = torch.rand(n, 8, generator=g).to(dev)
    x = torch.rand(n, 8, generator=g).to(dev)
    ea = torch.rand(ei.size(1), 8, generator=g).to(dev)
    nidx = torch.arange(n, device=dev)
    y = (torch.rand(n, 16, generator=g) > 0.5).float().to(dev)
    torch.manual_seed(1)
    model = rev_restated.RevGCN(num_layers=3, hidden=224, num_tasks=16, aggr="max", dropout=0.0, node_table=table,
                                impl="product", composed_edges=True).to(dev).train()
    held = {}

    def fwd_bwd():
        for p in model.parameters():
            p.grad = None
        pred, _ = model(x, nidx, ei, ea)
        torch.nn.functional.binary_cross_entropy_with_logits(pred, y).backward()
        held["pred"] = pred.detach()

    fwd_bwd()
    torch.cuda.synchronize()
    ref_pred = held["pred"].clone()
    ref_grads = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    graphed = GraphedStep(fwd_bwd, warmup=2)
    captured_pred = held["pred"]                      # the tensor the captured step writes on every replay
    for _ in range(3):
        captured_pred.fill_(float("nan"))             # a replay that skipped work would leave this behind
        graphed()
    torch.cuda.synchronize()
    assert torch.equal(captured_pred, ref_pred)
    assert ref_grads
    for k, p in model.named_parameters():
        if k in ref_grads:
            r = ref_grads[k]
            torch.testing.assert_close(p.grad, r, rtol=1e-4, atol=1e-6 * max(float(r.abs().max()), 1e-3),
                                       msg=lambda m, k=k: f"{k}: {m}")


def test_replayed_back_to_back_encoder_aggregations_lose_no_work_item():
    """Round-5 regression.  Consecutive per-edge-encoder aggregations of one captured step get the SAME workspace block
    from torch's allocator, i.e. the same work-item counters.  While the library re-armed them with hipMemsetAsync (a memset
    NODE of the captured graph), ~10 % of the replayed launches found the counters of a few queues non-zero at their first
    claims: the first item of those queues -- rows 0 .. 2 of the ogbn-proteins cluster graph -- was never processed and
    its output / arg-max ids were whatever the block held before (bench.py; an illegal address once the ids were used as
    addresses by the max backward).  Here: 24
    back-to-back launches (max: output and gradients both depend on the ids) in one graph, 30 replays, every output and
    every input gradient bit-equal to the eager pass; the buffers are poisoned between replays."""
    from deep_gcns_torch_amd import ops
    assert not ops.ENC_STATIC_ITEMS
    dev = torch.device("cuda:0")
    s = synth.SHAPES["proteins_cluster"]
    ei = synth.powerlaw_graph(s["n"], s["n_undirected"], s["seed"], device=dev)
    n, E, C, L = s["n"], ei.size(1), 112, 24
    gen = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(n, C, device=dev, generator=gen).requires_grad_(True)
    feat = torch.rand(E, 8, device=dev, generator=gen)
    W = (torch.randn(L, C, 8, device=dev, generator=gen) / 3).requires_grad_(True)
    b = torch.randn(L, C, device=dev, generator=gen).requires_grad_(True)
    probe = torch.randn(n, C, device=dev, generator=gen)
    outs = torch.empty(L, n, C, device=dev)
    gxs = torch.empty(L, n, C, device=dev)

    def run():
        for i in range(L):
            out = ops.gen_aggregate(x, ei, feat, aggr="max", edge_encoder=(W[i], b[i]))
            gx, = torch.autograd.grad(out, x, probe)
            outs[i].copy_(out.detach())
            gxs[i].copy_(gx)

    with ops.options(enc_max_winner_bwd=True):
        run()
        torch.cuda.synchronize()
        ref_o, ref_g = outs.clone(), gxs.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            run()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            run()
    for rep in range(30):
        outs.fill_(float("nan"))
        gxs.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(outs, ref_o), f"replay {rep}: outputs differ from the eager pass"
        assert torch.equal(gxs, ref_g), f"replay {rep}: input gradients differ from the eager pass"
