"""csrc/edgeconv_bwd.hip: the input gradient and the [dW | db] partials of the dense EdgeConv2d layer from dP | dQ
(reference: autograd through BasicConv's 1x1 Conv2d over cat([x_i, x_j - x_i]), gcn_lib/dense/torch_vertex.py:31-35,
gcn_lib/dense/torch_nn.py:48-60; SURVEY.md Appendix A "Dense EdgeConv": dx = (W1-W2)^T dP + W2^T dQ, dW1 = dP x^T,
dW2 = (dQ - dP) x^T, db = sum dP).  Through the C ABI against the float64 formulas, and the layer as a whole against the
float64 torch graph of the reference's formulation."""
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 2e-6          # max error / (sum of |terms| of the dot product): an fp32 fma chain over <= 32,768 terms


def _dev():
    return torch.device("cuda:0")


SHAPES = [  # B, C, N, Cout
    (8, 64, 4096, 64),      # config 2's backbone layer
    (8, 9, 4096, 64),       # its head (in_channels = 9)
    (2, 64, 1000, 64),      # N not a multiple of 16 / 64: chunks straddle samples
    (3, 96, 515, 80),       # two channel blocks, two row blocks, ragged everything
    (1, 4, 17, 4),
    (2, 33, 70, 18),        # odd C: 2C = 66, the scalar load path of the input kernel (K = 36 is a multiple of 4)
    (1, 16, 64, 7),         # K = 14: not a multiple of 4
    (2, 96, 128, 80),       # N % 64 == 0 with 2 Cout > 128 and C > 64: the staged store under the chunked contraction,
                            # two channel blocks; the weight kernel's 16-byte loads with two row and two channel blocks
    (1, 64, 64, 72),        # 2 Cout = 144: one point-tile group, contraction in two chunks
]


@pytest.mark.parametrize("B,C,N,Cout", SHAPES)
@pytest.mark.parametrize("res,g_layout", [(None, "slice"), (1.0, "slice"), (0.5, "slice"), (0.5, "permuted")])
def test_edgeconv_input_gradient_kernel(B, C, N, Cout, res, g_layout):
    from deep_gcns_torch_amd import _lib
    lib = _lib.load()
    dev = _dev()
    gen = torch.Generator().manual_seed(B * 1000 + C * 10 + Cout)
    dpq = torch.randn(B, N, 2 * Cout, generator=gen).to(dev)
    w = torch.randn(Cout, 2 * C, generator=gen).to(dev)
    # the upstream gradient as a channel slice of a wider tensor: element strides, not contiguous
    if g_layout == "slice":
        gfull = torch.randn(B, C + 3, N, generator=gen).to(dev)
        g = gfull[:, 1:C + 1]
    else:                                                                   # point-major memory behind a (B, C, N) view
        g = torch.randn(B, N, C + 3, generator=gen).to(dev).permute(0, 2, 1)[:, 1:C + 1]
    dx = torch.full((B, C, N), float("nan"), device=dev)
    with _lib.device_ctx(dev):
        _lib.check(lib.dgcn_edgeconv_bwd_input_f32(
            dpq.data_ptr(), w.data_ptr(), g.data_ptr() if res is not None else None, g.stride(0), g.stride(1), g.stride(2),
            float(res or 0.0), B, C, N, Cout, dx.data_ptr(), _lib.current_stream_handle(dev)), "dgcn_edgeconv_bwd_input_f32")
    torch.cuda.synchronize()
    w1, w2 = w[:, :C].double(), w[:, C:].double()
    dP, dQ = dpq[..., :Cout].double(), dpq[..., Cout:].double()
    ref = (dP @ (w1 - w2) + dQ @ w2).transpose(1, 2)                       # (B, C, N)
    scale = (dP.abs() @ (w1 - w2).abs() + dQ.abs() @ w2.abs()).transpose(1, 2)
    if res is not None:
        ref = ref + res * g.double()
        scale = scale + abs(res) * g.double().abs()
    err = ((dx.double() - ref).abs() / scale.clamp(min=1e-30)).max().item()
    assert torch.isfinite(dx).all()
    assert err < TOL, err


@pytest.mark.parametrize("B,C,N,Cout", SHAPES)
def test_edgeconv_weight_gradient_kernel(B, C, N, Cout):
    from deep_gcns_torch_amd import _lib
    lib = _lib.load()
    dev = _dev()
    gen = torch.Generator().manual_seed(B * 1000 + C * 10 + Cout + 1)
    dpq = torch.randn(B, N, 2 * Cout, generator=gen).to(dev)
    xfull = torch.randn(B, C + 5, N, generator=gen).to(dev)
    x = xfull[:, 2:C + 2]                                                   # channel slice: strided samples
    nparts = lib.dgcn_edgeconv_bwd_weight_num_partials(B, N)
    assert 1 <= nparts <= 256
    width = Cout * 2 * C + Cout
    parts = torch.full((nparts, width), float("nan"), device=dev)
    with _lib.device_ctx(dev):
        _lib.check(lib.dgcn_edgeconv_bwd_weight_f32(
            dpq.data_ptr(), x.data_ptr(), x.stride(0), x.stride(1), x.stride(2), B, C, N, Cout, parts.data_ptr(),
            _lib.current_stream_handle(dev)), "dgcn_edgeconv_bwd_weight_f32")
    assert torch.isfinite(parts).all()                                      # every element of every block is written
    got = _lib.sum_partials(parts).double()
    xr = x.double().permute(0, 2, 1).reshape(B * N, C)
    dP, dQ = dpq[..., :Cout].double().reshape(B * N, Cout), dpq[..., Cout:].double().reshape(B * N, Cout)
    dW = torch.cat([dP.t() @ xr, (dQ - dP).t() @ xr], dim=1)                # (Cout, 2C): the Conv2d weight's layout
    sW = torch.cat([dP.abs().t() @ xr.abs(), (dQ.abs() + dP.abs()).t() @ xr.abs()], dim=1)
    db, sb = dP.sum(0), dP.abs().sum(0)
    errw = ((got[:Cout * 2 * C].view(Cout, 2 * C) - dW).abs() / sW.clamp(min=1e-30)).max().item()
    errb = ((got[Cout * 2 * C:] - db).abs() / sb.clamp(min=1e-30)).max().item()
    assert errw < TOL and errb < TOL, (errw, errb)
    # fixed order: bit-reproducible
    parts2 = torch.empty_like(parts)
    with _lib.device_ctx(dev):
        _lib.check(lib.dgcn_edgeconv_bwd_weight_f32(
            dpq.data_ptr(), x.data_ptr(), x.stride(0), x.stride(1), x.stride(2), B, C, N, Cout, parts2.data_ptr(),
            _lib.current_stream_handle(dev)), "dgcn_edgeconv_bwd_weight_f32")
    assert torch.equal(parts, parts2)


def test_edgeconv_backward_argument_checks():
    from deep_gcns_torch_amd import _lib
    lib = _lib.load()
    dev = _dev()
    t = torch.zeros(64, device=dev)
    s = _lib.current_stream_handle(dev)
    assert lib.dgcn_edgeconv_bwd_input_f32(None, t.data_ptr(), None, 0, 0, 0, 0.0, 1, 4, 4, 4, t.data_ptr(), s) != 0
    assert lib.dgcn_edgeconv_bwd_input_f32(t.data_ptr(), t.data_ptr(), None, 0, 0, 0, 0.0, 1, 0, 4, 4, t.data_ptr(), s) != 0
    assert lib.dgcn_edgeconv_bwd_input_f32(t.data_ptr(), t.data_ptr(), None, 0, 0, 0, 0.0, 0, 4, 4, 4, t.data_ptr(), s) == 0
    assert lib.dgcn_edgeconv_bwd_weight_f32(t.data_ptr(), None, 0, 0, 0, 1, 4, 4, 4, t.data_ptr(), s) != 0
    assert lib.dgcn_edgeconv_bwd_weight_f32(t.data_ptr(), t.data_ptr(), 16, 4, 1, 1, 4, 4, -1, t.data_ptr(), s) != 0
    assert lib.dgcn_edgeconv_bwd_weight_num_partials(0, 5) == 0


@pytest.mark.parametrize("B,C,N,Cout,k,res", [(2, 64, 1024, 64, 16, 1.0), (2, 9, 600, 64, 8, None), (1, 24, 333, 40, 5, None),
                                              # C > 64 and 2 Cout > 128: the P | Q producer's chunked contraction and
                                              # second column block, the backward kernels' second blocks
                                              (1, 96, 256, 80, 6, None), (2, 80, 128, 80, 4, 0.5)])
@pytest.mark.parametrize("norm", ["batch", None])
def test_edgeconv_layer_gradients_match_the_float64_graph_and_the_library_glue(B, C, N, Cout, k, res, norm):
    """The whole layer (forward + backward through the C ABI) against the reference's formulation in float64, and the
    new kernels against the library-call glue they replace (same dP | dQ, so the two agree to fp32 rounding)."""
    from deep_gcns_torch_amd import dense_ops
    from deep_gcns_torch_amd.gcn_lib.dense.torch_vertex import EdgeConv2d
    dev = _dev()
    torch.manual_seed(C + N)
    conv = EdgeConv2d(C, Cout, act="relu", norm=norm, bias=True).to(dev).train()
    x = torch.randn(B, C, N, 1, device=dev)
    idx = torch.stack([torch.randint(0, N, (B, N, k), device=dev),
                       torch.arange(N, device=dev).view(1, N, 1).expand(B, N, k)])
    probe = torch.randn(B, Cout, N, 1, device=dev)

    def run(kernels):
        dense_ops.EDGECONV_BWD_KERNELS = kernels
        try:
            xx = x.clone().requires_grad_(True)
            for p in conv.parameters():
                p.grad = None
            y = conv(xx, idx, res_scale=res)                     # (the skip connection rides in the layer's last kernel)
            (y * probe).sum().backward()
            return [xx.grad.clone()] + [p.grad.clone() for p in conv.parameters()]
        finally:
            dense_ops.EDGECONV_BWD_KERNELS = True

    new, old = run(True), run(False)
    for a, b in zip(new, old):
        scale = float(b.abs().max()) + 1e-30
        assert float((a - b).abs().max()) <= 2e-5 * scale, (a.shape, float((a - b).abs().max()), scale)

    # float64 graph of the reference's formulation (batched_index_select + cat + Conv2d + act + BN + max)
    conv64 = EdgeConv2d(C, Cout, act="relu", norm=norm, bias=True).double().to(dev).train()
    conv64.load_state_dict({k_: v.double() for k_, v in conv.state_dict().items()})
    x64 = x.double().requires_grad_(True)
    xi = x64.squeeze(-1)                                                    # (B, C, N)
    xj = torch.gather(xi.unsqueeze(2).expand(B, C, N, N), 3, idx[0].unsqueeze(1).expand(B, C, N, k)) if N <= 1024 else None
    if xj is None:
        pytest.skip("float64 gather graph sized for N <= 1024")
    feat = torch.cat([xi.unsqueeze(-1).expand(B, C, N, k), xj - xi.unsqueeze(-1)], dim=1)
    y64 = conv64.nn(feat).max(-1, keepdim=True)[0]
    if res is not None and C == Cout:
        y64 = y64 + res * x64
    (y64 * probe.double()).sum().backward()
    ref = [x64.grad] + [p.grad for p in conv64.parameters()]
    for a, b in zip(new, ref):
        scale = float(b.abs().max()) + 1e-30
        # relu kinks / arg-max ties move single terms between two fp32 evaluations: 1e-3 of the gradient's scale
        assert float((a.double() - b).abs().max()) <= 1e-3 * scale, (a.shape, float((a.double() - b).abs().max()), scale)
