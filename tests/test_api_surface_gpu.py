"""GPU: the less-travelled parts of the drop-in API against the oracle: utils.pyg_util.scatter_,
GenMessagePassing.aggregate on materialised messages, sparse dynamic blocks (kNN + conv on flat clouds),
dense Plain/Dense blocks, BondEncoder edge features, EdgeConv2d with a channel count the fused path
does not take (composed path)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _install():
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()


@pytest.mark.parametrize("name", ["add", "mean", "max", "min"])
def test_scatter_shim(name):
    _install()
    from utils.pyg_util import scatter_
    from deep_gcns_torch_amd import synth
    from oracle import sparse_ref
    dev = _dev()
    ei = synth.tricky_graph()
    g = torch.Generator().manual_seed(4)
    src = torch.randn(ei.size(1), 20, generator=g) * 3
    src[5] = -30000.0                                                   # exercises the < -10000 -> 0 fix-up
    a = src.clone().requires_grad_(True)
    ref = sparse_ref.scatter_(name, a, ei[1], dim_size=257)
    ref.sum().backward()
    b = src.to(dev).requires_grad_(True)
    out = scatter_(name, b, ei[1].to(dev), dim_size=257)
    out.sum().backward()
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(b.grad.cpu(), a.grad, rtol=1e-4, atol=1e-6)


def test_aggregate_on_materialised_messages_and_bond_encoder():
    _install()
    from gcn_lib.sparse.torch_vertex import GENConv
    from deep_gcns_torch_amd import synth
    from oracle import sparse_ref
    dev = _dev()
    ei = synth.tricky_graph(n=64, e=700, hub_deg=300, seed=7)
    torch.manual_seed(0)
    conv = GENConv(16, 16, aggr="softmax", t=0.5, norm="layer", encode_edge=True, bond_encoder=True).to(dev)
    msgs = torch.rand(700, 16)
    out = conv.aggregate(msgs.to(dev), ei[1].to(dev), dim_size=64)
    ref = sparse_ref.gen_aggregate_messages(msgs, ei[1], 64, aggr="softmax", t=0.5)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-4, atol=1e-6)
    x = torch.randn(64, 16)
    bonds = torch.stack([torch.randint(0, 5, (700,)), torch.randint(0, 6, (700,)), torch.randint(0, 2, (700,))], 1)
    y = conv(x.to(dev), ei.to(dev), bonds.to(dev))
    cpu = GENConv(16, 16, aggr="softmax", t=0.5, norm="layer", encode_edge=True, bond_encoder=True)
    cpu.load_state_dict(conv.state_dict())
    emb = cpu.edge_encoder(bonds)
    ref = cpu.mlp(x + sparse_ref.gen_propagate(x, ei, emb, aggr="softmax", t=0.5))
    torch.testing.assert_close(y.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("conv", ["mr", "edge"])
def test_sparse_dynamic_blocks_on_flat_clouds(conv):
    """sem_seg_sparse style: features (B*N, C) + batch vector -> ResDynBlock (kNN on features, then conv)."""
    _install()
    from gcn_lib.sparse import DenseDynBlock, PlainDynBlock, ResDynBlock
    from gcn_lib.sparse.torch_edge import knn_graph_matrix
    from oracle import sparse_ref
    dev = _dev()
    torch.manual_seed(1)
    B, N, C = 2, 96, 12
    x = torch.randn(B * N, C)
    batch = torch.arange(B).repeat_interleave(N)
    blk = ResDynBlock(C, 6, 2, conv, "relu", "batch", True, 1).to(dev).train()
    out, b2 = blk(x.to(dev), batch.to(dev))
    assert out.shape == (B * N, C) and b2 is not None
    # same edges on the CPU through the oracle
    ei = knn_graph_matrix(x.to(dev), 12, batch.to(dev))[:, ::2].cpu()
    cpu = ResDynBlock(C, 6, 2, conv, "relu", "batch", True, 1).train()
    cpu.load_state_dict({k: v for k, v in blk.state_dict().items() if "num_batches" not in k and "running" not in k}, strict=False)
    nn_ = cpu.body.gconv.nn
    ref = (sparse_ref.mrconv_forward(x, ei, nn_) if conv == "mr" else sparse_ref.edgeconv_forward(x, ei, nn_)) + x
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=2e-4, atol=2e-5)
    assert PlainDynBlock(C, 6, 1, conv, "relu", None).to(dev)(x.to(dev), batch.to(dev))[0].shape == (B * N, C)
    assert DenseDynBlock(C, 8, 6, 1, conv, "relu", None).to(dev)(x.to(dev), batch.to(dev))[0].shape == (B * N, C + 8)


def test_dense_blocks_and_composed_edgeconv():
    _install()
    from gcn_lib.dense import DenseDynBlock2d, EdgeConv2d, PlainDynBlock2d, dense_knn_matrix
    from oracle import dense_ref
    dev = _dev()
    torch.manual_seed(2)
    x = torch.randn(2, 8, 64, 1)
    assert PlainDynBlock2d(8, 4, 2, "mr", "relu", "batch").to(dev)(x.to(dev)).shape == (2, 8, 64, 1)
    assert DenseDynBlock2d(8, 12, 4, 1, "edge", "leakyrelu", None).to(dev)(x.to(dev)).shape == (2, 20, 64, 1)
    conv = EdgeConv2d(8, 10, "relu", "batch", True).train()              # 10 % 4 != 0 -> composed path
    ei = dense_knn_matrix(x.to(dev), 5).cpu()
    xr = x.clone().requires_grad_(True)
    ref = dense_ref.edgeconv2d(xr, ei, conv.nn)
    ref.sum().backward()
    dconv = EdgeConv2d(8, 10, "relu", "batch", True)
    dconv.load_state_dict({k: v for k, v in conv.state_dict().items() if "running" not in k and "num_batches" not in k}, strict=False)
    dconv.to(dev).train()
    xd = x.to(dev).requires_grad_(True)
    out = dconv(xd, ei.to(dev))
    out.sum().backward()
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("aggr,norm,act", [("add", None, "relu"), ("mean", "batch", "relu"), ("max", "layer", "relu"),
                                           ("max", "batch", "prelu"), ("add", "batch", "leakyrelu")])
def test_sparse_edgconv_every_option_of_the_reference(aggr, norm, act):
    """EdgConv(aggr != 'max') and layer norm / PReLU inside it (gcn_lib/sparse/torch_vertex.py:106-114 passes them
    straight to tg.nn.EdgeConv): evaluated per edge, reduced by the HIP aggregation kernel, against the oracle."""
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()
    from deep_gcns_torch_amd import synth
    from gcn_lib.sparse.torch_vertex import EdgConv
    from oracle import thirdparty as tp
    dev = torch.device("cuda:0")
    ei = synth.tricky_graph()
    torch.manual_seed(3)
    m = EdgConv(24, 40, act, norm, True, aggr)
    x = torch.randn(257, 24)
    probe = torch.randn(257, 40)
    xr = x.clone().requires_grad_(True)
    m.train()
    ref = tp.EdgeConv(m.nn, aggr)(xr, ei)                      # the same nn module, per-edge on the CPU
    (ref * probe).sum().backward()
    ref_grads = {k: p.grad.clone() for k, p in m.named_parameters()}
    m.zero_grad()
    import copy
    md = copy.deepcopy(m).to(dev).train()
    if norm == "batch":                                        # the CPU pass above already advanced the running stats
        for mod in md.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.reset_running_stats()
    xd = x.to(dev).requires_grad_(True)
    out = md(xd, ei.to(dev))
    (out * probe.to(dev)).sum().backward()
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=1e-3, atol=1e-4 * float(xr.grad.abs().max()) + 1e-6)
    # one scale for all parameters: the bias in front of a BatchNorm has a gradient of exactly zero in real numbers
    # (rounding noise in both implementations), so its own magnitude is no yardstick
    scale = max(float(g.abs().max()) for g in ref_grads.values())
    for k, p in md.named_parameters():
        g = ref_grads[k]
        torch.testing.assert_close(p.grad.cpu(), g, rtol=1e-3, atol=2e-4 * scale + 1e-6)


def test_ragged_batches_through_knn_graph():
    """torch_cluster-style knn_graph on clouds of different sizes (sorted batch vector): one launch per cloud; exact
    neighbour distances on lattice clouds, ids offset per cloud, self excluded."""
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()
    from deep_gcns_torch_amd import synth
    from gcn_lib.sparse.torch_edge import knn_graph
    dev = torch.device("cuda:0")
    sizes = [150, 97, 230]
    clouds = [synth.lattice_cloud(1, 5, n, seed=40 + i)[0, :, :, 0].t().contiguous() for i, n in enumerate(sizes)]
    x = torch.cat(clouds)
    batch = torch.cat([torch.full((n,), i) for i, n in enumerate(sizes)])
    k = 7
    ei = knn_graph(x.to(dev), k, batch.to(dev)).cpu()
    assert ei.shape == (2, sum(sizes) * k)
    start = 0
    for c, n in zip(clouds, sizes):
        d = torch.cdist(c.double(), c.double()) ** 2
        d.fill_diagonal_(float("inf"))
        want = torch.sort(d, dim=1).values[:, :k]
        sl = slice(start * k, (start + n) * k)
        nb = ei[0, sl].view(n, k) - start
        ct = ei[1, sl].view(n, k) - start
        assert bool((nb >= 0).all()) and bool((nb < n).all())
        assert torch.equal(ct, torch.arange(n).view(n, 1).expand(n, k))
        got = torch.gather(d, 1, nb)
        torch.testing.assert_close(got, want, rtol=0, atol=1e-9)
        start += n



@pytest.mark.parametrize("act,norm", [("relu", "instance"), ("prelu", "batch"), ("prelu", None)])
def test_dense_edgeconv_options_outside_the_vertex_split(act, norm):
    """EdgeConv2d(norm='instance') / act='prelu' (gcn_lib/dense/torch_nn.py:9-33) run the reference's per-edge
    formulation on library ops: compared with the same stack evaluated on the CPU, parameters and their gradients too."""
    import copy
    from gcn_lib.dense import EdgeConv2d, DenseDilatedKnnGraph
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    m = EdgeConv2d(12, 20, act, norm, True).train()
    x = torch.randn(2, 12, 96, 1)
    probe = torch.randn(2, 20, 96, 1)
    md = copy.deepcopy(m).to(dev)
    ei = DenseDilatedKnnGraph(6, 2)(x.to(dev))
    xr = x.clone().requires_grad_(True)
    eic = ei.cpu()
    xi = torch.gather(xr.squeeze(-1), 2, eic[1].reshape(2, 1, -1).expand(2, 12, -1)).view(2, 12, 96, 6)
    xj = torch.gather(xr.squeeze(-1), 2, eic[0].reshape(2, 1, -1).expand(2, 12, -1)).view(2, 12, 96, 6)
    ref = m.nn(torch.cat([xi, xj - xi], dim=1)).max(-1, keepdim=True)[0]
    (ref * probe).sum().backward()
    xd = x.to(dev).requires_grad_(True)
    out = md(xd, ei)
    (out * probe.to(dev)).sum().backward()
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=1e-3, atol=1e-5)
    scale = max(float(p.grad.abs().max()) for p in m.parameters())
    for (k, p), (_, q) in zip(md.named_parameters(), m.named_parameters()):
        torch.testing.assert_close(p.grad.cpu(), q.grad, rtol=1e-3, atol=2e-4 * scale + 1e-6)
