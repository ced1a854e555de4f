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

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("identity_dropout_mask")]


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _install():
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()


@pytest.mark.parametrize("layers,norm,mlp_layers,mode", [(4, "batch", 1, "full"), (10, "batch", 2, "full"),
                                                         (10, "layer", 1, "aggregation"), (10, "batch", 1, "aggregation"),
                                                         (28, "batch", 1, "full")])      # config 3's depth, fwd AND bwd
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
    # Gradients.  A deep stack cannot be compared with its float64 evaluation (or two fp32 evaluations with each other)
    # elementwise: a ReLU whose pre-activation lies within fp32 rounding of zero takes different branches in two
    # evaluations, and one flipped element moves a weight gradient by percents (attribution.ReluDecisions).  The
    # yardstick is the float64 evaluation of the model file's loop on the host (oracle aggregation) ALONG THE DEVICE
    # RUN'S ReLU DECISIONS, one replay per route: every parameter gradient of both routes must match its replay to fp32
    # rounding.
    import attribution
    import config_replays
    from gcn_lib.sparse import torch_message

    def host_along(decisions, dtype=torch.float64):
        ref = copy.deepcopy(plain).cpu().to(dtype)
        for q in ref.parameters():
            q.grad = None
        ref.checkpoint_grad = False
        saved_prop = torch_message.GenMessagePassing.propagate

        def propagate(self, *a, **k):             # (the message ReLU lives inside the aggregation: not a replayed site)
            with decisions.suspended():
                return config_replays.oracle_propagate(self, *a, **k)
        torch_message.GenMessagePassing.propagate = propagate
        try:
            with decisions.replaying():
                torch.nn.functional.nll_loss(ref(x.cpu().to(dtype), ei.cpu()), y.cpu()).backward()
        finally:
            torch_message.GenMessagePassing.propagate = saved_prop
        return ref

    for route, model in (("model file's loop", plain), ("fused route", fused)):
        for q in model.parameters():
            q.grad = None
        dec = attribution.ReluDecisions()
        saved_mode = fuse.CHECKPOINT
        fuse.CHECKPOINT = mode
        try:
            # the recomputation of a checkpointed layer repeats its ReLU sites: record the first (no-grad) visit only
            model_ckpt = model.checkpoint_grad
            model.checkpoint_grad = False
            with dec.recording():
                loss = torch.nn.functional.nll_loss(model(x, ei), y)
            model.checkpoint_grad = model_ckpt
            for q in model.parameters():
                q.grad = None
            torch.nn.functional.nll_loss(model(x, ei), y).backward()       # the route as shipped, checkpointing on
        finally:
            fuse.CHECKPOINT = saved_mode
        ref64 = host_along(dec)
        errs = attribution.gradient_errors(model, ref64)
        worst = max(errs.items(), key=lambda kv: kv[1])
        # what fp32 rounding does to these gradients through `layers` normalised layers: the same replay (same branches) in
        # float32 on the host against the float64 one -- the device (fp32 partial sums per workgroup for the BatchNorm
        # statistics, six-product bf16 GEMMs, another summation order in the aggregation) has to stay within 3x that,
        # floor 1e-5.  Measured (profiles/r06_test_gates.json): 28 layers 6.5e-6 on the device, 8.4e-6 on the host.
        # Rounds 3 - 5 gated this at max(10 x the host replay, 3e-4) and measured up to 3.7e-3 at 28 layers, "explained" by
        # flipped decisions and later by rounding amplified through the BatchNorm layers.  It was the INSTRUMENT: its forced
        # ReLU handed an exact 0 instead of a tiny positive number to the message ReLU wherever the device had let a
        # pre-activation through that is rounding-level negative on the host (attribution.passed_value), which cut one
        # gradient term per such element -- in the device comparison AND in the host's float32 yardstick (7e-4)
        worst32 = max(attribution.gradient_errors(host_along(dec, torch.float32), ref64).values())
        from conftest import gate
        gate(f"fuse deepergcn {layers}-{norm}-{mlp_layers}-{mode}, {route}: worst parameter gradient vs float64 along its own ReLU "
             f"decisions, in units of max(3 x the host's float32 replay [{worst32:.2e}], 1e-5)",
             worst[1] / max(3 * worst32, 1e-5), 1.0, what=f"{worst[0]} {worst[1]:.3e}")
        print(f"[fuse {layers}-{norm}-{mlp_layers}-{mode}] {route}: {dec.n_decisions()} ReLU decisions replayed, worst "
              f"gradient error {worst[1]:.2e} of max |grad| (host float32 replay: {worst32:.2e})")
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
    from conftest import gate
    gate(f"fused revgcn3 {aggr}: prediction, fused route vs the model file's forward (max error in units of 1e-5 (1 + |ref|))",
         float(((out_f - out_p).abs() / (1e-5 + 1e-5 * out_p.abs())).max()), 1.0)          # measured 0.05
    worst, wk = 0.0, ""
    for (k, a), (_, b) in zip(fused.named_parameters(), plain.named_parameters()):
        assert a.grad is not None, k
        err = float((a.grad - b.grad).abs().max()) / max(1.0, float(b.grad.abs().max()))
        rel = float((a.grad - b.grad).abs().max()) / (float(b.grad.abs().max()) + 1e-30)
        if rel > worst:
            worst, wk = rel, k
        assert err < 2e-3, k
    # the two routes round the per-edge pre-activation differently ((A We) W_l in two roundings against A (We W_l) in one):
    # an arg-max / relu decision within that rounding flips and moves one gradient term -- 7.7e-7 when none does (round 5,
    # torch composition), 5.2e-4 with one flip (round 5, composition kernel)
    gate(f"fused revgcn3 {aggr}: worst parameter gradient, fused route vs the model file's forward (max error / max)", worst,
         2e-3, what=wk)
    # integer edge features (no Linear composition possible): the file's own forward
    assert not fuse._revgcn_qualifies(fused, x, ea.long())
