"""GPU: the fused node-wise layer pieces of SURVEY.md 8 f1 against float64 torch on the CPU.

  rows_linear            csrc/rows_linear.hip   Linear over node rows on the bf16 matrix pipe (fp32-faithful six-product
                                                split) with bias, residual, next-BatchNorm statistics, bias gradient
                                                  gcn_lib/sparse/torch_nn.py:50-71, torch_vertex.py:70-76
  pre_activation         csrc/rows_norm.hip     norm -> ReLU -> dropout in one apply pass (hash mask regenerated in the
                                                backward, or the reversible model's shared mask), ReLU mask recomputed
                                                  examples/ogb/ogbn_arxiv/model.py:96-99, eff_gcn_modules/rev/rev_layer.py:38-46
  res_plus_layer         blocks.py              h + conv(dropout(relu(norm(h)))) with the statistics handed from GEMM to norm
                                                  examples/ogb/ogbn_arxiv/model.py:90-106
Tolerances: fp32 results against fp64 references, 2e-5 relative to the natural scale of the terms (written at each check).
"""
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,K,C", [(5000, 128, 128), (3001, 112, 224), (4100, 224, 112), (2500, 64, 64),
                                      (2049, 100, 128), (3000, 256, 128), (3000, 128, 256), (2304, 16, 40),
                                      (169343, 128, 128)])
@pytest.mark.parametrize("with_res", [False, True])
def test_rows_linear_forward_backward(rows, K, C, with_res):
    from deep_gcns_torch_amd import node_ops
    dev = _dev()
    g = torch.Generator().manual_seed(rows + 7 * K + 13 * C)
    x = torch.randn(rows, K, generator=g)
    w = torch.randn(C, K, generator=g) / K ** 0.5
    b = torch.randn(C, generator=g)
    res = torch.randn(rows, C, generator=g) if with_res else None
    probe = torch.randn(rows, C, generator=g)
    assert node_ops.rows_linear_supported(x.to(dev), w.to(dev))

    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    rr = res.double().requires_grad_(True) if with_res else None
    yr = xr @ wr.t() + br + (rr if with_res else 0.0)
    (yr * probe.double()).sum().backward()

    xd, wd, bd = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    rd = res.to(dev).requires_grad_(True) if with_res else None
    y, stats = node_ops.rows_linear(xd, wd, bd, rd, want_stats=True)
    (y * probe.to(dev)).sum().backward()

    # |error| <= ~2e-7 * sum_k |x||w| per element (six-product split) + fp32 accumulation: compare on that scale
    nat = float((x.double().abs() @ w.double().abs().t()).max())
    torch.testing.assert_close(y.detach().cpu().double(), yr.detach(), rtol=1e-5, atol=2e-6 * nat)
    natg = float((probe.double().abs() @ w.double().abs()).max())
    torch.testing.assert_close(xd.grad.cpu().double(), xr.grad, rtol=1e-5, atol=2e-6 * natg)
    torch.testing.assert_close(wd.grad.cpu().double(), wr.grad, rtol=2e-5, atol=1e-5 * float(wr.grad.abs().max()))
    torch.testing.assert_close(bd.grad.cpu().double(), br.grad, rtol=2e-5, atol=1e-5 * float(br.grad.abs().max()))
    if with_res:
        torch.testing.assert_close(rd.grad.cpu().double(), rr.grad, rtol=0, atol=0)
    # statistics partials: sum y | sum y^2 over all workgroups
    tot = stats.double().sum(0).cpu()
    torch.testing.assert_close(tot[0], yr.detach().sum(0), rtol=1e-5, atol=1e-5 * nat * rows ** 0.5)
    torch.testing.assert_close(tot[1], (yr.detach() ** 2).sum(0), rtol=2e-5, atol=1e-6 * nat * nat * rows)


@pytest.mark.parametrize("rows,C,K,strided", [(5000, 128, 128, False), (100_003, 112, 224, True), (3001, 64, 64, False),
                                              (2500, 100, 40, False), (4100, 112, 224, False), (2049, 20, 132, False), (3000, 47, 128, False),
                                              (2111, 128, 256, False), (13253, 224, 112, False), (5000, 256, 64, True),
                                              (169343, 128, 128, False)])
def test_rows_tn_weight_gradient(rows, C, K, strided):
    """g^T x on the matrix pipe (csrc/rows_tn.hip) against float64: the weight gradient of the row-wise Linear layers and
    of the fused edge encoder (x = the strided per-group view of the (E, 2 hidden) embedding)."""
    from deep_gcns_torch_amd import node_ops
    dev = _dev()
    gen = torch.Generator().manual_seed(rows + C + K)
    g = torch.randn(rows, C, generator=gen)
    xf = torch.randn(rows, 2 * K if strided else K, generator=gen)
    x = xf[:, K:] if strided else xf
    ref = g.double().t() @ x.double()
    xd = xf.to(dev)
    out = node_ops.rows_tn(g.to(dev), xd[:, K:] if strided else xd)
    nat = float((g.double().abs().t() @ x.double().abs()).max())
    torch.testing.assert_close(out.cpu().double(), ref, rtol=1e-5, atol=2e-6 * nat)
    node_ops.ROWS_TN_KERNEL = False
    try:
        lib_out = node_ops.rows_tn(g.to(dev), xd[:, K:] if strided else xd)
    finally:
        node_ops.ROWS_TN_KERNEL = True
    torch.testing.assert_close(out, lib_out, rtol=1e-4, atol=1e-5 * nat)


@pytest.mark.parametrize("rows,C,K", [(5000, 128, 128), (13253, 224, 112), (13253, 112, 224), (4100, 112, 224),
                                      (2500, 100, 40), (2049, 20, 132), (3000, 44, 128), (2111, 128, 256),
                                      (5003, 256, 64), (2050, 132, 20), (169343, 128, 128)])
def test_rows_tn_bias_gradient_from_the_same_pass(rows, C, K):
    """``rows_tn(g, x, with_colsum=True)``: the loading waves of the weight-gradient kernel also leave ``g.sum(0)`` (the
    bias gradient of gcn_lib/sparse/torch_nn.py:50-71's Linear) -- in both operand orders (a g wider than 128 columns is
    the kernel's second operand and the result is written transposed), remainder rows and partial column blocks
    included; the matrix itself is bit-identical to the call without the sums."""
    from deep_gcns_torch_amd import node_ops
    dev = _dev()
    gen = torch.Generator().manual_seed(rows + 3 * C + K)
    g = torch.randn(rows, C, generator=gen) + 0.25
    x = torch.randn(rows, K, generator=gen)
    gd, xd = g.to(dev), x.to(dev)
    out, gb = node_ops.rows_tn(gd, xd, with_colsum=True)
    plain = node_ops.rows_tn(gd, xd)
    assert out.shape == (C, K) and gb.shape == (C,) and out.is_contiguous()
    assert torch.equal(out, plain)
    ref = g.double().t() @ x.double()
    nat = float((g.double().abs().t() @ x.double().abs()).max())
    torch.testing.assert_close(out.cpu().double(), ref, rtol=1e-5, atol=2e-6 * nat)
    ref_b = g.double().sum(0)
    torch.testing.assert_close(gb.cpu().double(), ref_b, rtol=1e-6, atol=2e-7 * float(g.double().abs().sum(0).max()))
    out2, gb2 = node_ops.rows_tn(gd, xd, with_colsum=True)
    assert torch.equal(out, out2) and torch.equal(gb, gb2)            # fixed summation order


def test_rows_linear_backward_takes_its_bias_gradient_from_rows_tn():
    """The Linear's backward is three launches (dX, g^T x, its partial sum) and its gradients match float64."""
    from deep_gcns_torch_amd import node_ops
    dev = _dev()
    gen = torch.Generator().manual_seed(11)
    for cin, cout in ((112, 224), (224, 112), (128, 128)):
        x = torch.randn(13253, cin, generator=gen)
        w = torch.randn(cout, cin, generator=gen) / cin ** 0.5
        b = torch.randn(cout, generator=gen)
        up = torch.randn(13253, cout, generator=gen)
        xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
        (torch.nn.functional.linear(xr, wr, br) * up.double()).sum().backward()
        xd, wd, bd = (t.to(dev).requires_grad_(True) for t in (x, w, b))
        y = node_ops.rows_linear(xd, wd, bd)
        calls = []
        orig = node_ops.rows_tn
        node_ops.rows_tn = lambda *a, **k: (calls.append(k), orig(*a, **k))[1]
        try:
            (y * up.to(dev)).sum().backward()
        finally:
            node_ops.rows_tn = orig
        assert calls == [{"with_colsum": True}]
        torch.testing.assert_close(xd.grad.cpu().double(), xr.grad, rtol=2e-5, atol=1e-5 * float(xr.grad.abs().max()))
        torch.testing.assert_close(wd.grad.cpu().double(), wr.grad, rtol=2e-5, atol=1e-5 * float(wr.grad.abs().max()))
        torch.testing.assert_close(bd.grad.cpu().double(), br.grad, rtol=2e-5, atol=1e-5 * float(br.grad.abs().max()))


@pytest.mark.parametrize("rows", [7, 32, 40, 65, 8 * 32 * 3 + 31])
def test_rows_tn_short_row_counts_through_the_abi(rows):
    """Whole steps, the zero-filled remainder step and workgroup ranges of dgcn_rows_tn_f32 at row counts the host wrapper
    hands to the library."""
    from deep_gcns_torch_amd import _lib
    dev = _dev()
    C, K = 112, 224
    gen = torch.Generator().manual_seed(rows)
    g = torch.randn(rows, C, generator=gen)
    x = torch.randn(rows, K, generator=gen)
    gd, xd = g.to(dev), x.to(dev)
    lib = _lib.load()
    nparts = lib.dgcn_rows_tn_num_partials(rows, C, K)
    assert nparts >= 1
    parts = torch.empty(nparts, C, K, device=dev)
    out = torch.full((C, K + 4), 7.0, device=dev)            # row stride K + 4: the padding stays untouched
    with _lib.device_ctx(dev):
        _lib.check(lib.dgcn_rows_tn_f32(gd.data_ptr(), C, xd.data_ptr(), K, rows, C, K, parts.data_ptr(), out.data_ptr(),
                                        K + 4, _lib.current_stream_handle(dev)), "dgcn_rows_tn_f32")
    ref = g.double().t() @ x.double()
    nat = float((g.double().abs().t() @ x.double().abs()).max())
    torch.testing.assert_close(out[:, :K].cpu().double(), ref, rtol=1e-5, atol=2e-6 * nat)
    assert bool((out[:, K:] == 7.0).all())


def test_rows_linear_strided_input_and_no_bias():
    from deep_gcns_torch_amd import node_ops
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    big = torch.randn(4000, 256, generator=g).to(dev)
    x = big[:, 128:]                                       # row stride 256, 16-byte aligned rows
    w = (torch.randn(96, 128, generator=g) / 11.0).to(dev)
    y = node_ops.rows_linear(x, w)
    ref = x.double().cpu() @ w.double().cpu().t()
    torch.testing.assert_close(y.cpu().double(), ref, rtol=1e-5, atol=1e-5)


def test_rows_linear_special_values():
    """Contract of the six-product split (csrc/bf16x6.h): finite inputs of any magnitude behave like fp32 (tiny values
    included -- the bf16 pipe keeps denormals); a non-finite input makes its whole output ROW non-finite (inf - inf in
    the split is NaN where a plain GEMM would give +-inf)."""
    from deep_gcns_torch_amd import node_ops
    dev = _dev()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2048, 64, generator=g)
    x[5] *= 1e-30
    x[6] *= 1e30
    w = torch.randn(64, 64, generator=g)
    y = node_ops.rows_linear(x.to(dev), w.to(dev)).cpu().double()
    ref = x.double() @ w.double().t()
    nat = x.double().abs() @ w.double().abs().t()
    assert torch.all((y - ref).abs() <= 1e-5 * nat + 1e-37)
    x[7, 3] = float("inf")
    y = node_ops.rows_linear(x.to(dev), w.to(dev)).cpu()
    assert not torch.isfinite(y[7]).any() and torch.isfinite(y[8]).all()


# ---------------------------------------------------------------------------------------------------------------------
def _load(ours, ref):
    ours.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in ref.state_dict().items()})


@pytest.mark.parametrize("kind", ["batch", "layer"])
@pytest.mark.parametrize("rows,C", [(3000, 128), (1025, 112), (777, 64), (513, 50)])
@pytest.mark.parametrize("drop", ["none", "hash", "mask", "mask_strided"])
def test_pre_activation_matches_reference_composition(kind, rows, C, drop):
    """norm -> relu -> dropout fused vs the three stock modules in float64 (the mask comes from the host replica of
    the kernels' hash, or is the shared tensor)."""
    from deep_gcns_torch_amd import node_ops
    if kind == "layer" and C % 4:
        pytest.skip("LayerNorm rows kernel: C % 4 == 0")
    dev = _dev()
    g = torch.Generator().manual_seed(rows + C)
    x = torch.randn(rows, C, generator=g) * 1.3 + 0.2
    probe = torch.randn(rows, C, generator=g)
    ref = (nn.BatchNorm1d(C) if kind == "batch" else nn.LayerNorm(C)).double().train()
    with torch.no_grad():
        ref.weight.copy_(torch.randn(C, generator=g).double())
        ref.bias.copy_(torch.randn(C, generator=g).double() * 0.3)
    ours = (node_ops.BatchNorm1d(C) if kind == "batch" else node_ops.LayerNorm(C))
    _load(ours, ref)
    ours = ours.to(dev).train()

    spec, factors = None, torch.ones(rows, C)
    if drop == "hash":
        spec = node_ops.DropSpec.hashed(0.3, seed=(12345, 678))
        factors = node_ops.hash_keep_factors(rows, C, spec.s0, spec.s1, spec.thr)
        assert abs(float((factors == 0).float().mean()) - 0.3) < 0.02          # the hash is a fair coin
        assert float(factors.max()) == pytest.approx(65536.0 / (65536 - spec.thr))
    elif drop in ("mask", "mask_strided"):
        factors = torch.zeros(rows, C).bernoulli_(0.8, generator=g) / 0.8
        if drop == "mask_strided":
            wide = torch.zeros(rows, 2 * C)
            wide[:, C:] = factors
            spec = node_ops.DropSpec.shared(wide.to(dev)[:, C:])               # a chunk view, row stride 2C
        else:
            spec = node_ops.DropSpec.shared(factors.to(dev))

    xr = x.double().requires_grad_(True)
    yr = torch.relu(ref(xr)) * factors.double()
    (yr * probe.double()).sum().backward()

    xd = x.to(dev).requires_grad_(True)
    y = ours(xd, fuse_relu=True, drop=spec)
    (y * probe.to(dev)).sum().backward()
    scale = max(1.0, float(yr.abs().max()))
    torch.testing.assert_close(y.detach().cpu().double(), yr.detach(), rtol=2e-5, atol=2e-6 * scale)
    if kind == "batch":
        var = x.double().var(0, unbiased=False)
        nat = float(probe.abs().max() * factors.max() * ref.weight.detach().abs().max() * (1.0 / torch.sqrt(var + ref.eps)).max())
    else:
        var = x.double().var(1, unbiased=False)
        nat = float(probe.abs().max() * factors.max() * ref.weight.detach().abs().max() * (1.0 / torch.sqrt(var + ref.eps)).max())
    # a pre-activation within fp32 rounding of zero may land on the other side of the ReLU than in float64: allow a
    # handful of such elements (each is off by one whole term)
    err = (xd.grad.cpu().double() - xr.grad).abs()
    bad = err > (2e-5 * xr.grad.abs() + 1e-5 * max(1.0, float(xr.grad.abs().max())) + 4e-6 * nat)
    assert int(bad.sum()) <= max(2, rows * C // 200000), f"{int(bad.sum())} gradient elements off"
    gw = max(1.0, float(ref.weight.grad.abs().max()))
    torch.testing.assert_close(ours.weight.grad.cpu().double(), ref.weight.grad, rtol=1e-4, atol=2e-5 * gw)
    torch.testing.assert_close(ours.bias.grad.cpu().double(), ref.bias.grad, rtol=1e-4, atol=2e-5 * gw)


def test_batchnorm_takes_statistics_from_the_gemm():
    """Lin -> BatchNorm1d(train): the BN finalize consumes the GEMM's per-workgroup partials; same output, same running
    statistics as with its own statistics pass."""
    from deep_gcns_torch_amd import node_ops
    dev = _dev()
    torch.manual_seed(3)
    x = torch.randn(20000, 128, device=dev)
    w = torch.randn(256, 128, device=dev) / 11.0
    b = torch.randn(256, device=dev)
    y, stats = node_ops.rows_linear(x, w, b, want_stats=True)
    bn1, bn2 = node_ops.BatchNorm1d(256).to(dev).train(), node_ops.BatchNorm1d(256).to(dev).train()
    o1 = bn1(y, fuse_relu=True, stats=stats)
    o2 = bn2(y, fuse_relu=True)
    torch.testing.assert_close(o1, o2, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(bn1.running_mean, bn2.running_mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(bn1.running_var, bn2.running_var, rtol=1e-5, atol=1e-6)
    assert int(bn1.num_batches_tracked) == 1


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("norm", ["batch", "layer"])
@pytest.mark.parametrize("use_checkpoint", [False, True])
def test_res_plus_layer_equals_the_reference_loop_body(norm, use_checkpoint):
    """blocks.res_plus_layer == norm -> relu -> (dropout p = 0) -> GENConv -> + h written with separate modules
    (examples/ogb/ogbn_arxiv/model.py:96-104), outputs and every gradient, two stacked layers so that the statistics
    hand-over is exercised."""
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()
    from deep_gcns_torch_amd import blocks, synth
    from gcn_lib.sparse.torch_nn import norm_layer
    from gcn_lib.sparse.torch_vertex import GENConv
    dev = _dev()
    torch.manual_seed(11)
    N, C = 6000, 128
    ei = synth.undirected_random_graph(N, 20000, seed=2, device=dev)
    norms = nn.ModuleList([norm_layer(norm, C) for _ in range(2)]).to(dev).train()
    convs = nn.ModuleList([GENConv(C, C, aggr="softmax_sg", t=0.1, norm=norm, mlp_layers=1 + i) for i in range(2)]).to(dev).train()
    h0 = torch.randn(N, C, device=dev)
    probe = torch.randn(N, C, device=dev)

    def run(fused):
        for m in list(norms) + list(convs):
            m.zero_grad(set_to_none=True)
        for m in norms:
            if hasattr(m, "reset_running_stats"):
                m.reset_running_stats()
        h = h0.clone().requires_grad_(True)
        x = h * 1.0
        stats = None
        for i in range(2):
            if fused:
                x, stats = blocks.res_plus_layer(norms[i], convs[i], x, ei, p=0.0, training=True, stats=stats,
                                                 use_checkpoint=use_checkpoint)
            else:
                h2 = torch.relu(norms[i](x))
                x = convs[i](h2, ei) + x
        (x * probe).sum().backward()
        grads = [p.grad.clone() for m in list(norms) + list(convs) for p in m.parameters()]
        return x.detach(), h.grad.clone(), grads

    y1, gh1, gp1 = run(True)
    y0, gh0, gp0 = run(False)
    torch.testing.assert_close(y1, y0, rtol=1e-4, atol=1e-4 * float(y0.abs().max()))
    torch.testing.assert_close(gh1, gh0, rtol=1e-4, atol=1e-4 * float(gh0.abs().max()))
    # a bias in front of a training-mode BatchNorm has an exactly-zero gradient (the mean is removed): what comes out is
    # rounding noise of the size of the OTHER gradients, so the absolute gate uses the layer-wide gradient scale
    gscale = max(float(b.abs().max()) for b in gp0)
    for a, b in zip(gp1, gp0):
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-4 * max(gscale, float(b.abs().max())))


def test_deepergcn_fused_layers_equal_the_plain_model():
    """The restated DeeperGCN ('res+', BatchNorm, checkpointing) with its layer loop through blocks.res_plus_layer
    against the reference-shaped loop: same parameters, same outputs / gradients / running statistics (dropout off)."""
    import os
    import sys
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import arch_restated
    from deep_gcns_torch_amd import synth
    dev = _dev()
    torch.manual_seed(5)
    N = 9000
    ei = synth.undirected_random_graph(N, 40000, seed=7, device=dev)
    plain = arch_restated.DeeperGCN(num_layers=9, in_channels=32, hidden=64, num_tasks=10).to(dev).train()
    fused = arch_restated.DeeperGCN(num_layers=9, in_channels=32, hidden=64, num_tasks=10, fused_layers=True).to(dev).train()
    fused.load_state_dict(plain.state_dict())
    x = torch.randn(N, 32, device=dev)
    y = torch.randint(0, 10, (N,), device=dev)
    o0 = plain(x, ei)
    torch.nn.functional.nll_loss(o0, y).backward()
    o1 = fused(x, ei)
    torch.nn.functional.nll_loss(o1, y).backward()
    torch.testing.assert_close(o1, o0, rtol=1e-3, atol=1e-4)
    # Gradients.  Nine stacked layers, each re-normalised by a training-mode BatchNorm: the two loops differ by fp32
    # rounding per layer (residual added in the GEMM epilogue, statistics summed in another order), and a pre-activation
    # within rounding of 0 crosses the ReLU in one of them: whole gradient terms move.  So neither loop is the yardstick
    # of the other: each is compared with the float64 evaluation of the model on the host ALONG ITS OWN ReLU decisions
    # (attribution.ReluDecisions; checkpointing off while recording, the recomputation would visit every site twice)
    import copy
    import attribution
    import config_replays
    from conftest import gate
    for route, model in (("plain loop", plain), ("res_plus_layer loop", fused)):
        sd = copy.deepcopy(model.state_dict())
        ck = model.checkpoint_grad
        model.checkpoint_grad = False
        dec = attribution.ReluDecisions()
        with dec.recording():
            model(x, ei)
        model.checkpoint_grad = ck
        model.load_state_dict(sd)                           # (the recording pass moved the running statistics)
        hosts = {}
        for dt in (torch.float64, torch.float32):
            host = arch_restated.DeeperGCN(num_layers=9, in_channels=32, hidden=64, num_tasks=10)
            host.load_state_dict(plain.state_dict())
            host = host.to(dt).train()
            host.checkpoint_grad = False
            attribution.float64_backward_along(
                dec, host, lambda mm, dt=dt: torch.nn.functional.nll_loss(mm(x.cpu().to(dt), ei.cpu()), y.cpu()),
                config_replays.oracle_propagate)
            hosts[dt] = host
        errs = attribution.gradient_errors(model, hosts[torch.float64])
        # the yardstick for "fp32 rounding through nine training-mode BatchNorm layers": the SAME replay in float32 on the
        # host (same branches, torch CPU kernels) against the float64 one
        errs32 = attribution.gradient_errors(hosts[torch.float32], hosts[torch.float64])
        worst = max(errs.items(), key=lambda kv: kv[1])
        worst32 = max(errs32.values())
        gate(f"deepergcn9 {route}: worst parameter gradient vs float64 along its own ReLU decisions, in units of max(3 x the "
             f"host's float32 replay of the same branches [{worst32:.2e}], 1e-5)", worst[1] / max(3 * worst32, 1e-5), 1.0,
             what=f"{worst[0]} {worst[1]:.3e}")
    for (n0, b0), (n1, b1) in zip(plain.named_buffers(), fused.named_buffers()):
        torch.testing.assert_close(b1.float(), b0.float(), rtol=1e-4, atol=1e-5, msg=n0)
    # with dropout: runs, finite, and about the right fraction of the pre-activations is dropped
    drop = arch_restated.DeeperGCN(num_layers=3, in_channels=32, hidden=64, num_tasks=10, dropout=0.5, fused_layers=True).to(dev).train()
    out = drop(x, ei)
    assert torch.isfinite(out).all()
    torch.nn.functional.nll_loss(out, y).backward()
    assert all(torch.isfinite(p.grad).all() for p in drop.parameters())


@pytest.mark.parametrize("aggr,kw", [("softmax_sg", {}), ("softmax", dict(learn_t=True)), ("power", dict(p=1.3, learn_p=True)),
                                     ("max", {})])
def test_checkpoint_with_kept_aggregation_equals_full_recomputation(aggr, kw):
    """blocks.res_plus_layer(use_checkpoint=True) keeps the aggregation's outputs of the first pass and recomputes the
    node-wise part only (ops.AggregationStash); "full" recomputes everything.  Same kernels on the same inputs either way:
    identical outputs and gradients, with dropout (the hash seed is drawn before the checkpoint) and learnable t / p."""
    import os
    import sys
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import arch_restated
    from deep_gcns_torch_amd import ops, synth
    dev = _dev()
    N = 6000
    ei = synth.undirected_random_graph(N, 30000, seed=11, device=dev)
    x = torch.randn(N, 32, generator=torch.Generator().manual_seed(3)).to(dev)
    y = torch.randint(0, 10, (N,), generator=torch.Generator().manual_seed(4)).to(dev)
    res = {}
    for mode in ("reference", "reference_full", "never"):
        torch.manual_seed(21)
        m = arch_restated.DeeperGCN(num_layers=9, in_channels=32, hidden=64, num_tasks=10, aggr=aggr, dropout=0.3,
                                    fused_layers=True, checkpoint=mode, **kw).to(dev).train()
        if aggr == "max":
            m.checkpoint_grad = mode != "never"              # (the reference checkpoints softmax / power stacks only)
        torch.manual_seed(22)                                # the dropout seeds of the layers
        out = m(x, ei)
        torch.nn.functional.nll_loss(out, y).backward()
        res[mode] = (out.detach(), [p.grad.clone() for p in m.parameters()])
    o_ref, g_ref = res["reference_full"]
    # learnable t / p: the launch that also writes the second moment is another instantiation of the kernel than the
    # no_grad launch of the full recomputation's first pass -- the passes then differ by fp32 rounding, not bit for bit
    exact = not (kw.get("learn_t") or kw.get("learn_p"))
    gscale = max(float(b.abs().max()) for b in g_ref)
    worst_ne = 0.0
    for mode in ("reference", "never"):
        o, g = res[mode]
        if exact:
            assert torch.equal(o, o_ref)
        else:
            torch.testing.assert_close(o, o_ref, rtol=1e-4, atol=1e-5)
        for a, b in zip(g, g_ref):
            if exact:
                assert torch.equal(a, b), mode
            else:
                floor = 1e-3 * gscale * b.numel() ** 0.5
                err = float((a - b).double().norm() / max(float(b.double().norm()), floor))
                worst_ne = max(worst_ne, err)
                assert err < 2e-4, (mode, err)
    if not exact:
        from conftest import gate
        gate(f"checkpoint modes, {aggr} with a learnable exponent / temperature: worst parameter gradient, kept aggregation "
             f"vs full recomputation (relative L2)", worst_ne, 2e-4)      # measured 2.6e-5


def test_aggregation_stash_refuses_a_different_recomputation():
    from deep_gcns_torch_amd import ops, synth
    dev = _dev()
    ei = synth.undirected_random_graph(500, 3000, seed=2, device=dev)
    x = torch.randn(500, 32, device=dev, requires_grad=True)
    st = ops.AggregationStash()
    with torch.no_grad(), ops.stash_aggregation(st, "record"):
        ops.gen_aggregate(x, ei, aggr="softmax", t=0.5)
    assert len(st.items) == 1
    with ops.stash_aggregation(st, "replay"):
        out = ops.gen_aggregate(x, ei, aggr="softmax", t=0.5)
        with pytest.raises(RuntimeError, match="more aggregations"):
            ops.gen_aggregate(x, ei, aggr="softmax", t=0.5)
    with ops.stash_aggregation(st, "replay"):
        with pytest.raises(RuntimeError, match="does not repeat"):
            ops.gen_aggregate(x, ei, aggr="max")
    ref = ops.gen_aggregate(x, ei, aggr="softmax", t=0.5)
    assert torch.equal(out, ref)
    g = torch.randn_like(out)
    g0, = torch.autograd.grad(out, x, g)
    g1, = torch.autograd.grad(ref, x, g)
    assert torch.equal(g0, g1)


@pytest.mark.parametrize("aggr,kw", [("softmax", dict(t=1.0)), ("max", {}), ("power", dict(p=1.0))])
@pytest.mark.parametrize("use_checkpoint", [False, True])
def test_composed_edge_embedding_in_a_res_plus_stack(aggr, kw, use_checkpoint):
    """The ogbn-proteins DeeperGCN pattern (examples/ogb/ogbn_proteins/model.py:90,107-128): a model-level
    Linear(8 -> hidden) edge encoder whose output every layer's GENConv encodes again with Linear(hidden -> hidden).
    blocks.ComposedEdgeEmbedding in place of the (E, hidden) tensor: same logits and the same gradients for BOTH Linear
    layers of every layer, with and without checkpointing."""
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()
    from deep_gcns_torch_amd import blocks, synth
    from gcn_lib.sparse.torch_nn import norm_layer
    from gcn_lib.sparse.torch_vertex import GENConv
    dev = _dev()
    torch.manual_seed(17)
    N, hidden, L = 5000, 64, 4
    ei = synth.powerlaw_graph(N, 40_000, seed=3, exponent=2.2).to(dev)
    E = ei.size(1)
    enc = torch.nn.Linear(8, hidden).to(dev)
    convs = torch.nn.ModuleList([GENConv(hidden, hidden, aggr=aggr, encode_edge=True, edge_feat_dim=hidden, norm="layer",
                                         mlp_layers=2, **kw) for _ in range(L)]).to(dev)
    norms = torch.nn.ModuleList([norm_layer("layer", hidden) for _ in range(L)]).to(dev)
    x0 = torch.randn(N, hidden, device=dev)
    ea = torch.rand(E, 8, device=dev)
    probe = torch.randn(N, hidden, device=dev)
    params = list(enc.parameters()) + list(convs.parameters()) + list(norms.parameters())

    def run(composed):
        for p in params:
            p.grad = None
        emb = blocks.ComposedEdgeEmbedding(enc, ea) if composed else enc(ea)
        h = convs[0](x0, ei, emb)
        for layer in range(1, L):
            h, _ = blocks.res_plus_layer(norms[layer - 1], convs[layer], h, ei, emb, p=0.0, training=True,
                                         want_stats=False, use_checkpoint=use_checkpoint)
        (h * probe).sum().backward()
        return h.detach(), [None if p.grad is None else p.grad.clone() for p in params]

    o1, g1 = run(True)
    o0, g0 = run(False)
    torch.testing.assert_close(o1, o0, rtol=2e-4, atol=2e-4 * float(o0.abs().max()))
    worst_c = 0.0
    for a, b, p in zip(g1, g0, params):
        assert (a is None) == (b is None)
        if b is not None:
            scale = float(b.abs().max()) + 1e-12
            worst_c = max(worst_c, float((a - b).abs().max()) / scale)
            assert float((a - b).abs().max()) / scale < 5e-4, tuple(p.shape)
    from conftest import gate
    gate(f"composed edge embedding in a res+ stack, {aggr}, checkpoint={use_checkpoint}: worst parameter gradient, composed vs "
         f"materialised embedding (max error / max)", worst_c, 5e-4)        # measured: 2e-6 (max), 1.2e-5 (power), 1.1e-4 (softmax)


@pytest.mark.parametrize("C,H,F,biases", [(112, 224, 8, (True, True)), (64, 64, 8, (True, False)), (32, 80, 3, (False, True)),
                                          (40, 256, 16, (False, False))])
def test_encoder_composition_kernels_match_the_two_linear_layers(C, H, F, biases):
    """blocks._ComposeEncoder (csrc/enc_compose.hip): (W', b') = (W_l We, W_l b_e + b_l) and the gradients of all four
    parameters, one launch each way, against the float64 composition of the two Linear layers
    (gcn_lib/sparse/torch_vertex.py:62-66 applied to model_rev.py:98's embedding)."""
    from deep_gcns_torch_amd import blocks
    from deep_gcns_torch_amd.blocks import ComposedEdgeEmbedding
    dev = _dev()
    g = torch.Generator().manual_seed(C + H + F)
    layer = torch.nn.Linear(H, C, bias=biases[0])
    enc = torch.nn.Linear(F, H, bias=biases[1])
    with torch.no_grad():
        for p in list(layer.parameters()) + list(enc.parameters()):
            p.copy_(torch.randn(p.shape, generator=g) / 4)
    pw, pb = torch.randn(C, F, generator=g), torch.randn(C, generator=g)
    l64, e64 = layer.double(), enc.double()
    w64 = l64.weight @ e64.weight
    b64 = None
    if biases[0] or biases[1]:
        b64 = (l64.weight @ e64.bias if biases[1] else 0) + (l64.bias if biases[0] else 0)
    loss = (w64 * pw.double()).sum() + ((b64 * pb.double()).sum() if b64 is not None else 0)
    params64 = list(l64.parameters()) + list(e64.parameters())
    grads64 = torch.autograd.grad(loss, params64)
    ld = torch.nn.Linear(H, C, bias=biases[0]).to(dev)
    ed = torch.nn.Linear(F, H, bias=biases[1]).to(dev)
    ld.load_state_dict({k: v.float() for k, v in l64.state_dict().items()})
    ed.load_state_dict({k: v.float() for k, v in e64.state_dict().items()})
    emb = ComposedEdgeEmbedding(ed, torch.rand(10, F, device=dev))
    assert blocks._compose_on_device(ld.weight, ld.bias, ed.weight, ed.bias)
    w, b = emb.composed(ld)
    torch.testing.assert_close(w.detach().cpu().double(), w64.detach(), rtol=1e-5, atol=1e-6)
    assert (b is None) == (b64 is None)
    lossd = (w * pw.to(dev)).sum()
    if b is not None:
        torch.testing.assert_close(b.detach().cpu().double(), b64.detach(), rtol=1e-5, atol=1e-6)
        lossd = lossd + (b * pb.to(dev)).sum()
    gradsd = torch.autograd.grad(lossd, list(ld.parameters()) + list(ed.parameters()))
    for a, r in zip(gradsd, grads64):
        torch.testing.assert_close(a.cpu().double(), r, rtol=1e-5, atol=1e-5 * float(r.abs().max()))


@pytest.mark.parametrize("shape", [(1024, 2, 112), (829, 112, 9), (52, 112, 224), (1, 7), (3, 5), (17, 1), (1000, 4097)])
def test_sum_partials_is_the_fixed_order_sum_over_the_first_axis(shape):
    """_lib.sum_partials (dgcn_reduce_partials_f32): one launch, bit-reproducible, against the float64 sum."""
    from deep_gcns_torch_amd import _lib
    dev = _dev()
    p = torch.randn(*shape, generator=torch.Generator().manual_seed(sum(shape))).to(dev)
    out = _lib.sum_partials(p)
    assert out.shape == p.shape[1:]
    ref = p.double().sum(0)
    scale = float(p.double().abs().sum(0).max())
    assert float((out.double() - ref).abs().max()) <= 1e-6 * max(scale, 1.0)
    assert torch.equal(out, _lib.sum_partials(p))
    assert torch.equal(_lib.sum_partials(p[:0]), torch.zeros_like(out))


def test_residual_gradient_view_scope_keeps_the_upstream_gradient_intact():
    """node_ops.residual_gradient_is_last_use: inside it the residual's gradient leaves rows_linear as a view of the
    upstream gradient (a leaf takes it over instead of cloning it); outside it -- or with a second consumer of the
    residual -- gradients accumulate as usual and the upstream gradient is never written to."""
    from deep_gcns_torch_amd import node_ops
    dev = _dev()
    torch.manual_seed(3)
    rows, K, C = 4099, 64, 64
    x = torch.randn(rows, K, device=dev)
    w = torch.randn(C, K, device=dev, requires_grad=True)
    b = torch.randn(C, device=dev, requires_grad=True)
    probe = torch.randn(rows, C, device=dev)

    def run(scoped, twice):
        r = torch.randn(rows, C, device=dev, generator=torch.Generator(device=dev).manual_seed(5)).requires_grad_(True)
        w.grad = b.grad = None
        if scoped:
            with node_ops.residual_gradient_is_last_use():
                y = node_ops.rows_linear(x, w, b, residual=r)
        else:
            y = node_ops.rows_linear(x, w, b, residual=r)
        g_seen = {}

        def keep(g):
            g_seen["g"] = g

        y.register_hook(keep)
        loss = (y * probe).sum() + ((r * 2.0).sum() if twice else 0.0)
        loss.backward()
        return r.grad, g_seen["g"]

    for scoped, twice in ((False, False), (False, True), (True, False)):
        rg, g = run(scoped, twice)
        assert torch.equal(g, probe)                                   # the upstream gradient is what it was
        assert torch.equal(rg, probe + 2.0 if twice else probe), (scoped, twice)
        if scoped:
            assert rg.data_ptr() == g.data_ptr()                       # taken over, not cloned


@pytest.mark.parametrize("shape", [(829, 112, 9), (52, 64, 9), (3, 5, 2), (1024, 7, 17), (1, 1, 2)])
def test_sum_partials_split_leaves_the_last_column_apart(shape):
    """_lib.sum_partials_split (dgcn_reduce_partials_split_f32) = sum_partials followed by the two slices, bit for bit."""
    from deep_gcns_torch_amd import _lib
    dev = _dev()
    p = torch.randn(*shape, generator=torch.Generator().manual_seed(sum(shape))).to(dev)
    whole = _lib.sum_partials(p)
    a, b = _lib.sum_partials_split(p)
    assert a.is_contiguous() and b.is_contiguous() and a.shape == (shape[1], shape[2] - 1) and b.shape == (shape[1],)
    assert torch.equal(a, whole[:, :-1]) and torch.equal(b, whole[:, -1])
    a0, b0 = _lib.sum_partials_split(p[:0])
    assert not a0.any() and not b0.any()


@pytest.mark.parametrize("rows,K,C,negate", [(13253, 224, 112, True), (13253, 224, 112, False), (5001, 64, 48, True),
                                             (2049, 128, 128, True)])
def test_rows_linear_writes_the_coupling_result_into_a_column_block(rows, K, C, negate):
    """``CouplingResidual``: ``res -/+ (x W^T + b)`` from the Linear's epilogue, written into a column block of a wider
    buffer (row stride 2 C) -- the additive coupling's ``x_i = y_i - F_i`` / ``y_i = x_i + F_i``
    (eff_gcn_modules/rev/memgcn.py:36-52) without an elementwise pass; the neighbouring block stays untouched, the
    returned tensor aliases the block and differentiates as F."""
    from deep_gcns_torch_amd import node_ops
    dev = _dev()
    gen = torch.Generator().manual_seed(rows + K)
    x = torch.randn(rows, K, generator=gen)
    w = torch.randn(C, K, generator=gen) / K ** 0.5
    b = torch.randn(C, generator=gen)
    res_full = torch.randn(rows, 2 * C, generator=gen)
    up = torch.randn(rows, C, generator=gen)
    f = x.double() @ w.double().t() + b.double()
    ref = res_full[:, C:].double() - f if negate else res_full[:, C:].double() + f
    buf = torch.full((rows, 2 * C), 7.0, device=dev)
    xd, wd, bd = (t.to(dev).requires_grad_(True) for t in (x, w, b))
    cr = node_ops.CouplingResidual(res_full.to(dev)[:, C:], buf[:, C:], negate=negate)
    y = node_ops.rows_linear(xd, wd, bd, cr)
    assert cr.used and y.data_ptr() == buf[:, C:].data_ptr() and y.stride() == (2 * C, 1)
    nat = float((x.double().abs() @ w.double().abs().t()).max()) + float(res_full.abs().max())
    torch.testing.assert_close(buf[:, C:].cpu().double(), ref, rtol=1e-5, atol=2e-6 * nat)
    assert bool((buf[:, :C] == 7.0).all())
    # the graph behind the returned values is F's: gradients of x W^T + b, whatever the sign of the epilogue
    gx, gw, gb = torch.autograd.grad(y, [xd, wd, bd], up.to(dev))
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    rx, rw, rb = torch.autograd.grad(torch.nn.functional.linear(xr, wr, br), [xr, wr, br], up.double())
    for got, want in ((gx, rx), (gw, rw), (gb, rb)):
        torch.testing.assert_close(got.cpu().double(), want, rtol=2e-5, atol=1e-5 * float(want.abs().max()))


def test_additive_coupling_folds_its_adds_into_the_last_linear():
    """GroupAdditiveCoupling over GENBlocks on device rows: forward and fused backward with the coupling's add / subtract
    folded into F_i's last Linear give the values of the elementwise form (FOLD_COUPLING = False) -- the rebuilt input,
    the input gradient, every weight gradient -- and launch no torch add / sub for them."""
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()
    from deep_gcns_torch_amd import ops, synth
    from deep_gcns_torch_amd.eff_gcn_modules.rev import memgcn, rev_layer
    dev = _dev()
    torch.manual_seed(3)
    N, C, g = 6000, 128, 2
    ei = synth.powerlaw_graph(N, 40_000, 2, device=dev)
    fms = torch.nn.ModuleList([rev_layer.GENBlock(C // g, C // g, aggr="max", norm="layer", mlp_layers=2) for _ in range(g)])
    coupling = memgcn.GroupAdditiveCoupling(fms, group=g).to(dev).train()
    x = torch.randn(N, C, device=dev)
    gy = torch.randn(N, C, device=dev)
    mask = (torch.rand(N, C // g, device=dev) > 0.2).float() / 0.8
    weights = tuple(p for p in coupling.parameters())
    res = {}
    for fold in (True, False):
        memgcn.FOLD_COUPLING = fold
        calls = []
        add, sub = torch.add, torch.sub
        torch.add = lambda *a, **k: (calls.append("add"), add(*a, **k))[1]
        torch.sub = lambda *a, **k: (calls.append("sub"), sub(*a, **k))[1]
        try:
            with torch.no_grad():
                y = coupling(x, ei, mask.repeat(1, g))
                xb, gx, wg = coupling.fused_backward(y, gy, ei, (mask.repeat(1, g),), weights, [], ops.edge_grad_sink)
        finally:
            torch.add, torch.sub = add, sub
            memgcn.FOLD_COUPLING = True
        res[fold] = (y, xb, gx, wg, calls)
    yf, xf, gxf, wgf, calls_f = res[True]
    yu, xu, gxu, wgu, calls_u = res[False]
    assert calls_u.count("sub") == g and calls_f.count("sub") == 0
    assert calls_f.count("add") == calls_u.count("add") - g          # the forward's y_i = x_i + F_i
    scale = float(yu.abs().max())
    torch.testing.assert_close(yf, yu, rtol=0, atol=2e-6 * scale)
    torch.testing.assert_close(xf, xu, rtol=0, atol=4e-6 * scale)
    torch.testing.assert_close(xf, x, rtol=0, atol=2e-5 * scale)     # the coupling is inverted
    torch.testing.assert_close(gxf, gxu, rtol=0, atol=1e-4 * float(gxu.abs().max()))
    for a, b in zip(wgf, wgu):
        torch.testing.assert_close(a, b, rtol=0, atol=1e-4 * float(b.abs().max()) + 1e-12)
