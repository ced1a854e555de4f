"""``deep_gcns_torch_amd.fuse``: the fast layer loops applied to UNCHANGED model classes from outside
(``install(fuse_models=True)`` / ``fuse.fuse_model``) are the same function of the same parameters as the model
file's own loop -- outputs, every gradient, the state_dict -- and fall back to the file's loop when an instance does not
qualify.  The classes are the restated model files (tests/arch_restated.py, tests/rev_restated.py: the reference's files
do not travel to the GPU box; tests/test_dropin.py checks the import hook against the real files in the build
container)."""
import copy

import pytest
import torch

import arch_restated
import rev_restated
from deep_gcns_torch_amd import synth

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _install():
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()


@pytest.mark.parametrize("layers,norm,mlp_layers,mode", [(4, "batch", 1, "full"), (10, "batch", 2, "full"),
                                                         (10, "layer", 1, "aggregation"), (10, "batch", 1, "aggregation")])
def test_fused_deepergcn_equals_the_model_files_loop(layers, norm, mlp_layers, mode):
    _install()
    from deep_gcns_torch_amd import fuse
    dev = _dev()
    n = 3000
    ei = synth.powerlaw_graph(n, 20_000, seed=5).to(dev)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, 24, generator=g).to(dev)
    y = torch.randint(0, 7, (n,), generator=g).to(dev)
    torch.manual_seed(3)
    kw = dict(num_layers=layers, in_channels=24, hidden=64, num_tasks=7, aggr="softmax_sg", t=0.1, norm=norm,
              mlp_layers=mlp_layers, dropout=0.0)
    plain = arch_restated.DeeperGCN(**kw).to(dev).train()
    fused = copy.deepcopy(plain)
    saved = fuse.CHECKPOINT
    fuse.CHECKPOINT = mode
    try:
        fuse.fuse_model(fused)
        assert type(fused).forward is fuse._deepergcn_forward and type(plain).forward is not fuse._deepergcn_forward
        assert list(fused.state_dict()) == list(plain.state_dict())
        assert fused.checkpoint_grad == (layers > 7)
        out_p = plain(x, ei)
        torch.nn.functional.nll_loss(out_p, y).backward()
        out_f = fused(x, ei)
        torch.nn.functional.nll_loss(out_f, y).backward()
    finally:
        fuse.CHECKPOINT = saved
    torch.testing.assert_close(out_f, out_p, rtol=1e-4, atol=1e-5)
    for (k, a), (_, b) in zip(fused.named_parameters(), plain.named_parameters()):
        # two fp32 evaluations of the same function (different kernels, different summation orders) through `layers`
        # normalised layers: elementwise, relative to the tensor's own scale
        # (a bias in front of a BatchNorm has a zero gradient: fp32 noise of 1e-9 on both sides, hence the floor)
        torch.testing.assert_close(a.grad, b.grad, rtol=1e-3, atol=1e-4 * float(b.grad.abs().max()) + 1e-7,
                                   msg=lambda m, k=k: f"{k}: {m}")
    for (k, a), (_, b) in zip(fused.named_buffers(), plain.named_buffers()):
        if "running" in k and ".mlp." not in k:          # (a BatchNorm inside a recomputed MLP updates twice per step)
            torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6, msg=k)
    # an instance that does not qualify takes the model file's own forward: another block type, CPU tensors
    fused.block = "plain"
    assert not fuse._deepergcn_qualifies(fused, x, ei)
    fused.block = "res+"
    assert fuse._deepergcn_qualifies(fused, x, ei) and not fuse._deepergcn_qualifies(fused, x.cpu(), ei.cpu())


@pytest.mark.parametrize("aggr", ["max", "power"])
def test_fused_revgcn_equals_the_model_files_forward(aggr):
    _install()
    from deep_gcns_torch_amd import fuse
    dev = _dev()
    n = 2000
    ei = synth.powerlaw_graph(n, 12_000, seed=9).to(dev)
    E = ei.size(1)
    g = torch.Generator().manual_seed(2)
    table = torch.rand(n, 8, generator=g).to(dev)
    x = torch.rand(n, 8, generator=g).to(dev)
    ea = torch.rand(E, 8, generator=g).to(dev)
    nidx = torch.arange(n, device=dev)
    y = (torch.rand(n, 16, generator=g) > 0.5).float().to(dev)
    torch.manual_seed(4)
    plain = rev_restated.RevGCNModelFile(num_layers=3, hidden=64, num_tasks=16, aggr=aggr, dropout=0.2, node_table=table,
                                         impl="product", learn_p=aggr == "power").to(dev).train()
    fused = copy.deepcopy(plain)
    fused.node_features = table
    fuse.fuse_model(fused)
    assert list(fused.state_dict()) == list(plain.state_dict())
    assert fuse._revgcn_qualifies(fused, x, ea)

    def step(m):
        torch.manual_seed(11)                      # the model draws its shared dropout mask inside forward
        torch.cuda.manual_seed(11)
        out = m(x, nidx, ei, ea)
        torch.nn.functional.binary_cross_entropy_with_logits(out, y).backward()
        return out.detach()
    out_p, out_f = step(plain), step(fused)
    # (A We) W_l in two roundings against A (We W_l) in one: the tolerance of tests/test_revgcn.py's composed path
    torch.testing.assert_close(out_f, out_p, rtol=2e-4, atol=2e-4)
    for (k, a), (_, b) in zip(fused.named_parameters(), plain.named_parameters()):
        assert a.grad is not None, k
        torch.testing.assert_close(a.grad, b.grad, rtol=2e-3, atol=2e-4 * max(1.0, float(b.grad.abs().max())),
                                   msg=lambda m, k=k: f"{k}: {m}")
    # integer edge features (no Linear composition possible): the file's own forward
    assert not fuse._revgcn_qualifies(fused, x, ea.long())
