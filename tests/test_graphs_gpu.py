"""``deep_gcns_torch_amd.graphs.GraphedStep``: a whole training step of a model built from this package's modules -- the
reversible RevGCN with composed per-edge encoders -- captured as one hipGraph leaves the same parameters as the eager
steps (every libdgcn entry point is asynchronous on the caller's stream and neither allocates nor reads back)."""
import copy

import pytest
import torch

import rev_restated
from deep_gcns_torch_amd import synth

pytestmark = pytest.mark.gpu


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
