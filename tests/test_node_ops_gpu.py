"""GPU: node-wise BatchNorm1d [+ReLU] HIP kernels (csrc/rows_norm.hip) against stock torch.nn.BatchNorm1d evaluated
on the CPU in float64 -- outputs, running statistics, and all three gradients; training and eval; float4 and scalar
layouts; strided rows.  Tolerance 1e-5 relative (fp32 op, fp64 reference)."""
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu

RTOL = 2e-5


def _dev():
    return torch.device("cuda:0")


def _ref(x, bn_ref, relu, probe):
    xr = x.double().requires_grad_(True)
    y = bn_ref(xr)
    if relu:
        y = torch.relu(y)
    (y * probe.double()).sum().backward()
    return y.detach(), xr.grad


@pytest.mark.parametrize("rows,C", [(1000, 128), (4097, 256), (777, 100), (513, 50), (64, 7), (2, 128), (169343, 128)])
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("relu", [False, True])
def test_batchnorm_rows_matches_torch(rows, C, training, relu):
    from deep_gcns_torch_amd.node_ops import BatchNorm1d
    dev = _dev()
    g = torch.Generator().manual_seed(rows * 31 + C)
    x = torch.randn(rows, C, generator=g) * 1.7 + 0.3
    probe = torch.randn(rows, C, generator=g)
    ref = nn.BatchNorm1d(C).double()
    with torch.no_grad():
        ref.weight.copy_(torch.randn(C, generator=g).double())
        ref.bias.copy_(torch.randn(C, generator=g).double())
        ref.running_mean.copy_(torch.randn(C, generator=g).double() * 0.1)
        ref.running_var.copy_(torch.rand(C, generator=g).double() + 0.5)
    ours = BatchNorm1d(C)
    ours.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in ref.state_dict().items()})
    ours = ours.to(dev)
    ref.train(training)
    ours.train(training)
    y_ref, gx_ref = _ref(x, ref, relu, probe)
    xd = x.to(dev).requires_grad_(True)
    y = ours(xd, fuse_relu=relu) if relu else ours(xd)
    (y * probe.to(dev)).sum().backward()
    scale = max(1.0, float(y_ref.abs().max()))
    torch.testing.assert_close(y.detach().cpu().double(), y_ref, rtol=RTOL, atol=2e-6 * scale)
    # dx = scale*(g' - mean g' - xhat*mean(g' xhat)) cancels heavily when there are few rows: the error scales with
    # the size of the TERMS (|g| * |gamma| * invstd), not with the (much smaller) result
    var = x.double().var(0, unbiased=False) if training else ref.running_var
    nat = float(probe.abs().max() * ref.weight.detach().abs().max() * (1.0 / torch.sqrt(var + ref.eps)).max())
    gs = max(1.0, float(gx_ref.abs().max()))
    torch.testing.assert_close(xd.grad.cpu().double(), gx_ref, rtol=RTOL, atol=1e-5 * gs + 2e-6 * nat)
    gw = max(1.0, float(ref.weight.grad.abs().max()))
    rt = RTOL if rows > 8 else 3e-4      # two rows: xhat = +-1 up to eps/var, fp32 (x - mean)*invstd is ill-conditioned
    torch.testing.assert_close(ours.weight.grad.cpu().double(), ref.weight.grad, rtol=rt, atol=1e-5 * gw)
    torch.testing.assert_close(ours.bias.grad.cpu().double(), ref.bias.grad, rtol=rt, atol=1e-5 * gw)
    torch.testing.assert_close(ours.running_mean.cpu().double(), ref.running_mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(ours.running_var.cpu().double(), ref.running_var, rtol=1e-5, atol=1e-6)
    assert int(ours.num_batches_tracked) == int(ref.num_batches_tracked)


def test_batchnorm_variants_strided_no_affine_cumulative_momentum():
    from deep_gcns_torch_amd.node_ops import BatchNorm1d
    dev = _dev()
    g = torch.Generator().manual_seed(4)
    wide = torch.randn(300, 96, generator=g)
    x = wide[:, 16:80]                                            # row stride 96, 64 channels, 16-byte aligned
    for kw in (dict(affine=False), dict(momentum=None), dict(track_running_stats=False)):
        ref = nn.BatchNorm1d(64, **kw).double()
        ours = BatchNorm1d(64, **kw).to(dev)
        for _ in range(3):                                        # several steps: cumulative average changes
            y_ref = ref(x.double())
            y = ours(wide.to(dev)[:, 16:80])
        torch.testing.assert_close(y.cpu().double(), y_ref, rtol=RTOL, atol=1e-5)
        if ref.running_mean is not None:
            torch.testing.assert_close(ours.running_mean.cpu().double(), ref.running_mean, rtol=1e-5, atol=1e-6)
            torch.testing.assert_close(ours.running_var.cpu().double(), ref.running_var, rtol=1e-5, atol=1e-6)
        ref.eval(); ours.eval()
        torch.testing.assert_close(ours(x.to(dev)).cpu().double(), ref(x.double()), rtol=RTOL, atol=1e-5)
    # 3-D input and a single training row keep torch's behaviour (stock path)
    bn = BatchNorm1d(8).to(dev)
    assert bn(torch.randn(4, 8, 5, device=dev)).shape == (4, 8, 5)
    with pytest.raises(ValueError):
        bn(torch.randn(1, 8, device=dev))


def test_mlp_fuses_batchnorm_relu_and_matches_stock_sequential():
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()
    from gcn_lib.sparse.torch_nn import MLP, norm_layer
    from deep_gcns_torch_amd.node_ops import BatchNorm1d
    dev = _dev()
    assert isinstance(norm_layer("batch", 16), BatchNorm1d) and isinstance(norm_layer("batch", 16), nn.BatchNorm1d)
    torch.manual_seed(0)
    mlp = MLP([64, 128, 64], norm="batch", last_lin=True).to(dev)
    assert list(mlp.state_dict().keys())[:4] == ["0.weight", "0.bias", "1.weight", "1.bias"]
    stock = nn.Sequential(nn.Linear(64, 128), nn.BatchNorm1d(128), nn.ReLU(), nn.Linear(128, 64)).to(dev)
    stock.load_state_dict(mlp.state_dict())
    x = torch.randn(20000, 64, device=dev)
    xa = x.clone().requires_grad_(True)
    xb = x.clone().requires_grad_(True)
    ya, yb = mlp(xa), stock(xb)
    ya.square().mean().backward()
    yb.square().mean().backward()
    torch.testing.assert_close(ya, yb, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(xa.grad, xb.grad, rtol=1e-4, atol=1e-7)
    for (na, pa), (nb, pb) in zip(mlp.named_parameters(), stock.named_parameters()):
        torch.testing.assert_close(pa.grad, pb.grad, rtol=2e-4, atol=1e-6, msg=lambda m: f"{na}: {m}")


@pytest.mark.parametrize("rows,C", [(1000, 128), (4097, 256), (777, 100), (513, 64), (64, 8), (3, 16), (2000, 512), (300, 1024),
                                    (169343, 128)])
@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("affine", [True, False])
def test_layernorm_rows_matches_torch(rows, C, relu, affine):
    """HIP LayerNorm [+ReLU] against nn.LayerNorm on the CPU in float64: output, dx, dgamma, dbeta; every lanes-per-row /
    float4-per-lane layout (C = 8 ... 1024, incl. C/4 not a power of two)."""
    from deep_gcns_torch_amd.node_ops import LayerNorm
    dev = _dev()
    g = torch.Generator().manual_seed(rows + C)
    x = torch.randn(rows, C, generator=g) * 2.0 + 0.5
    probe = torch.randn(rows, C, generator=g)
    ref = nn.LayerNorm(C, elementwise_affine=affine).double()
    ours = LayerNorm(C, elementwise_affine=affine)
    if affine:
        with torch.no_grad():
            ref.weight.copy_(torch.randn(C, generator=g).double())
            ref.bias.copy_(torch.randn(C, generator=g).double())
        ours.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    ours = ours.to(dev)
    xr = x.double().requires_grad_(True)
    yr = ref(xr)
    if relu:
        yr = torch.relu(yr)
    (yr * probe.double()).sum().backward()
    xd = x.to(dev).requires_grad_(True)
    y = ours(xd, fuse_relu=True) if relu else ours(xd)
    (y * probe.to(dev)).sum().backward()
    torch.testing.assert_close(y.detach().cpu().double(), yr.detach(), rtol=RTOL, atol=1e-5)
    gs = max(1.0, float(xr.grad.abs().max()))
    torch.testing.assert_close(xd.grad.cpu().double(), xr.grad, rtol=1e-4, atol=2e-5 * gs)
    if affine:
        gw = max(1.0, float(ref.weight.grad.abs().max()))
        torch.testing.assert_close(ours.weight.grad.cpu().double(), ref.weight.grad, rtol=1e-4, atol=1e-5 * gw)
        torch.testing.assert_close(ours.bias.grad.cpu().double(), ref.bias.grad, rtol=1e-4, atol=1e-5 * gw)


def test_layernorm_module_fallbacks_and_mlp_fusion():
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()
    from gcn_lib.sparse.torch_nn import MLP, norm_layer
    from deep_gcns_torch_amd.node_ops import LayerNorm
    dev = _dev()
    ln = norm_layer("layer", 64)
    assert isinstance(ln, LayerNorm) and isinstance(ln, nn.LayerNorm) and list(ln.state_dict()) == ["weight", "bias"]
    ln = ln.to(dev)
    x3 = torch.randn(5, 7, 64, device=dev)                              # leading dims are flattened
    torch.testing.assert_close(ln(x3), nn.functional.layer_norm(x3, (64,), ln.weight, ln.bias, ln.eps), rtol=1e-5, atol=1e-5)
    odd = LayerNorm(50).to(dev)                                         # C % 4 != 0: stock path
    xo = torch.randn(9, 50, device=dev)
    torch.testing.assert_close(odd(xo), nn.functional.layer_norm(xo, (50,), odd.weight, odd.bias, odd.eps))
    torch.manual_seed(1)
    mlp = MLP([64, 128, 64], norm="layer", last_lin=True).to(dev)
    stock = nn.Sequential(nn.Linear(64, 128), nn.LayerNorm(128), nn.ReLU(), nn.Linear(128, 64)).to(dev)
    stock.load_state_dict(mlp.state_dict())
    x = torch.randn(5000, 64, device=dev)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    mlp(xa).square().mean().backward()
    stock(xb).square().mean().backward()
    torch.testing.assert_close(xa.grad, xb.grad, rtol=1e-4, atol=1e-7)
    for (na, pa), (nb, pb) in zip(mlp.named_parameters(), stock.named_parameters()):
        torch.testing.assert_close(pa.grad, pb.grad, rtol=2e-4, atol=1e-6, msg=lambda m: f"{na}: {m}")


@pytest.mark.parametrize("rows,C", [(1000, 128), (777, 100), (513, 64), (64, 8), (300, 1024), (169343, 128)])
@pytest.mark.parametrize("add_x", [False, True])
def test_msgnorm_rows_matches_reference_formula(rows, C, add_x):
    """MsgNorm (torch_message.py:95-99) [+ residual] against the stock composition in float64: y, dx, dmsg, dscale;
    includes an all-zero message row (normalize clamps the norm) and an all-zero x row."""
    import torch.nn.functional as F
    from deep_gcns_torch_amd import node_ops
    dev = _dev()
    g = torch.Generator().manual_seed(rows * 7 + C)
    x = torch.randn(rows, C, generator=g)
    m = torch.randn(rows, C, generator=g) * 3.0
    m[3] = 0.0
    x[5] = 0.0
    probe = torch.randn(rows, C, generator=g)
    s0 = torch.tensor([0.7])
    xr, mr, sr = x.double().requires_grad_(True), m.double().requires_grad_(True), s0.double().requires_grad_(True)
    yr = F.normalize(mr, p=2, dim=1) * xr.norm(p=2, dim=1, keepdim=True) * sr
    if add_x:
        yr = xr + yr
    (yr * probe.double()).sum().backward()
    xd, md = x.to(dev).requires_grad_(True), m.to(dev).requires_grad_(True)
    sd = s0.to(dev).requires_grad_(True)
    assert node_ops.msg_norm_supported(xd, md)
    y = node_ops.msg_norm_rows(xd, md, sd, add_x=add_x)
    (y * probe.to(dev)).sum().backward()
    torch.testing.assert_close(y.detach().cpu().double(), yr.detach(), rtol=RTOL, atol=1e-5)
    torch.testing.assert_close(md.grad.cpu().double(), mr.grad, rtol=1e-4, atol=1e-5 * float(mr.grad.abs().max()))
    gx_ref = xr.grad.clone()
    gx_ref[5] = torch.nan_to_num(gx_ref[5], nan=0.0) if not add_x else torch.nan_to_num(gx_ref[5], nan=0.0)
    if add_x:
        gx_ref[5] = probe[5].double()              # d||x||/dx at 0 is taken as 0 (torch yields nan for the norm term)
    torch.testing.assert_close(xd.grad.cpu().double(), gx_ref, rtol=1e-4, atol=1e-5 * float(gx_ref.abs().max()))
    torch.testing.assert_close(sd.grad.cpu().double(), sr.grad, rtol=1e-4, atol=1e-6 * float(sr.grad.abs().max()) + 1e-6)


def test_genconv_with_msgnorm_uses_the_fused_residual_and_matches_composition():
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()
    from deep_gcns_torch_amd import node_ops, synth
    from gcn_lib.sparse.torch_vertex import GENConv
    dev = _dev()
    torch.manual_seed(3)
    conv = GENConv(64, 64, aggr="softmax", t=1.0, learn_t=True, msg_norm=True, learn_msg_scale=True, norm="batch").to(dev)
    ei = synth.undirected_random_graph(3000, 20000, seed=2, device=dev)
    x = torch.randn(3000, 64, device=dev)

    def run(fused):
        saved = node_ops.msg_norm_supported
        if not fused:
            node_ops.msg_norm_supported = lambda *a: False
        try:
            conv.zero_grad()
            xa = x.clone().requires_grad_(True)
            conv(xa, ei).square().mean().backward()
            return xa.grad.clone(), conv.msg_norm.msg_scale.grad.clone(), conv.t.grad.clone()
        finally:
            node_ops.msg_norm_supported = saved
    a, b = run(True), run(False)
    for u, v in zip(a, b):
        torch.testing.assert_close(u, v, rtol=1e-3, atol=1e-6 * float(v.abs().max()) + 1e-9)
