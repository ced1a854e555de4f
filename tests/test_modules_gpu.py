"""GPU parity of the drop-in gcn_lib modules against golden vectors produced by the reference's own
modules (oracle/make_golden.py): same state_dict, same inputs -> outputs, input grads, parameter
grads and BatchNorm running statistics within 1e-4 relative."""
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

MODS = load_golden("sparse_modules.pt")
DENSE = load_golden("dense.pt")
RTOL = 1e-4


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _close(a, b, rtol=RTOL, atol_scale=2e-6, what=""):
    b = b.to(a.dtype)
    scale = max(float(b.abs().max()), 1.0)
    torch.testing.assert_close(a.detach().cpu(), b.cpu(), rtol=rtol, atol=atol_scale * scale, msg=lambda m: f"{what}: {m}")


def _build_sparse(case):
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()
    from gcn_lib import sparse
    name = case["name"]
    if name.startswith("genconv"):
        return sparse.GENConv(**case["ctor"])
    if name.startswith("graphconv"):
        return sparse.GraphConv(**case["ctor"])
    if name.startswith("resgraphblock"):
        return sparse.ResGraphBlock(**case["ctor"])
    raise AssertionError(name)


@pytest.mark.parametrize("case", MODS, ids=lambda c: c["name"])
def test_sparse_module_matches_reference(case):
    dev = _dev()
    m = _build_sparse(case)
    assert list(m.state_dict().keys()) == list(case["state_dict_before"].keys())
    m.load_state_dict(case["state_dict_before"])
    m.to(dev).train()
    x = case["x"].to(dev).requires_grad_(True)
    ei = case["edge_index"].to(dev)
    if "edge_attr" in case:
        out = m(x, ei, case["edge_attr"].to(dev))
    else:
        out = m(x, ei)
    if isinstance(out, tuple):
        out = out[0]
    _close(out, case["out"], what="out")
    (out * case["probe"].to(dev)).sum().backward()
    _close(x.grad, case["grad_x"], what="grad_x")
    named = dict(m.named_parameters())
    for pname, g in case["param_grads"].items():
        if g is None:
            continue
        _close(named[pname].grad, g, rtol=5e-4, atol_scale=4e-5, what=f"grad {pname}")
    after = m.state_dict()
    for k, v in case["state_dict"].items():            # running stats after one training step
        if "running" in k or "num_batches" in k:
            _close(after[k].float(), v.float(), rtol=1e-4, atol_scale=1e-5, what=k)


# ---------------------------------------------------------------------------------------- dense
@pytest.mark.parametrize("case", [c for c in DENSE if c["kind"] == "knn"], ids=lambda c: c["name"])
def test_dense_knn_matches_reference(case):
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()
    from gcn_lib.dense import DenseDilatedKnnGraph, dense_knn_matrix
    dev = _dev()
    x, k, d = case["x"].to(dev), case["k"], case["dilation"]
    dist = case["dist"]                                   # (B,N,N) reference distances
    B, N = dist.shape[:2]
    full = dense_knn_matrix(x, k * d)
    assert full.shape == (2, B, N, k * d) and full.dtype == torch.int64
    mine = full[0].cpu()
    ctr = torch.arange(N).view(1, N, 1).expand(B, N, k * d)
    assert torch.equal(full[1].cpu(), ctr)
    # every row lists K distinct points whose reference distances are exactly the K smallest, ascending
    dm = torch.gather(dist, 2, mine)
    ref_sorted = torch.sort(dist, dim=2).values[:, :, :k * d]
    lattice = "randn" not in case["name"]
    if lattice:
        assert torch.equal(dm, ref_sorted), "selected distances differ from the reference top-K"
    else:
        torch.testing.assert_close(dm, ref_sorted, rtol=0, atol=2e-5)
    assert bool((torch.sort(mine, dim=2).values.diff(dim=2) != 0).all()), "duplicate neighbour"
    if case["edge_index_full"] is not None:
        ref = case["edge_index_full"][0].long()
        # bit-exact indices on tie-free rows (torch.topk leaves equal distances unordered)
        gaps = ref_sorted.diff(dim=2)
        tie_free = (gaps > (0 if lattice else 1e-5)).all(dim=2)
        assert tie_free.float().mean() > 0.5
        assert torch.equal(mine[tie_free], ref[tie_free])
    dil = DenseDilatedKnnGraph(k, d)(x)
    assert dil.shape == (2, B, N, k)
    assert torch.equal(dil[0].cpu(), mine[:, :, ::d])     # dilation fused in the kernel == strided pick
    refd = case["edge_index"][0].long()
    dmd = torch.gather(dist, 2, dil[0].cpu())
    if lattice:
        assert torch.equal(dmd, torch.gather(dist, 2, refd))


def test_knn_on_channel_slice_view_and_stochastic_rng_stream():
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()
    from gcn_lib.dense import DenseDilatedKnnGraph
    from deep_gcns_torch_amd import synth
    from oracle import dense_ref
    dev = _dev()
    pos = synth.lattice_cloud(2, 3, 200, seed=3)
    feats = torch.randn(2, 6, 200, 1)
    inputs = torch.cat([pos, feats], dim=1).to(dev)       # (B,9,N,1); kNN on the xyz slice (a view)
    g = DenseDilatedKnnGraph(8, 1)
    ei = g(inputs[:, 0:3])
    ref = dense_ref.dense_knn_matrix(pos, 8)
    dist = dense_ref.pairwise_distance(pos.transpose(2, 1).squeeze(-1))
    assert torch.equal(torch.gather(dist, 2, ei[0].cpu()), torch.gather(dist, 2, ref[0]))
    # stochastic dilation consumes the CPU RNG exactly like the reference (rand(1) then randperm)
    gs = DenseDilatedKnnGraph(4, 3, stochastic=True, epsilon=1.0).train()
    torch.manual_seed(123)
    out = gs(inputs[:, 0:3])
    torch.manual_seed(123)
    assert torch.rand(1) < 1.0
    pick = torch.randperm(12)[:4]
    full = DenseDilatedKnnGraph(12, 1)(inputs[:, 0:3])
    assert torch.equal(out, full[:, :, :, pick])


def _build_dense_conv(case):
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()
    from gcn_lib import dense
    cls = getattr(dense, case["cls"])
    return cls(case["Cin"], case["Cout"], case["act"], case["norm"], True)


@pytest.mark.parametrize("case", [c for c in DENSE if c["kind"] == "conv"], ids=lambda c: c["name"])
def test_dense_conv_matches_reference(case):
    dev = _dev()
    m = _build_dense_conv(case)
    assert list(m.state_dict().keys()) == list(case["state_dict_before"].keys())
    m.load_state_dict(case["state_dict_before"])
    m.to(dev).train()
    x = case["x"].to(dev).requires_grad_(True)
    ei = case["edge_index"].to(dev)
    out = m(x, ei)
    assert out.shape == case["out"].shape
    _close(out, case["out"], what="out")
    (out * case["probe"].to(dev)).sum().backward()
    _close(x.grad, case["grad_x"], rtol=2e-4, atol_scale=1e-5, what="grad_x")
    named = dict(m.named_parameters())
    for pname, g in case["param_grads"].items():
        _close(named[pname].grad, g, rtol=5e-4, atol_scale=2e-5, what=f"grad {pname}")
    after = m.state_dict()
    for k, v in case["state_dict_after"].items():
        if "running" in k or "num_batches" in k:
            _close(after[k].float(), v.float(), rtol=1e-4, atol_scale=1e-5, what=k)
    m.eval()
    with torch.no_grad():
        _close(m(x, ei), case["out_eval"], what="eval out")


def test_dense_resdynblock_matches_reference():
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()
    from gcn_lib.dense import ResDynBlock2d
    dev = _dev()
    case = next(c for c in DENSE if c["kind"] == "block")
    blk = ResDynBlock2d(**case["ctor"])
    blk.load_state_dict(case["state_dict_before"])
    blk.to(dev).train()
    x = case["x"].to(dev).requires_grad_(True)
    out = blk(x)
    _close(out, case["out"], what="out")
    (out * case["probe"].to(dev)).sum().backward()
    _close(x.grad, case["grad_x"], rtol=2e-4, atol_scale=1e-5, what="grad_x")
    named = dict(blk.named_parameters())
    for pname, g in case["param_grads"].items():
        _close(named[pname].grad, g, rtol=5e-4, atol_scale=2e-5, what=f"grad {pname}")


@pytest.mark.parametrize("res_scale,conv,norm", [(1, "edge", "batch"), (0.5, "edge", "batch"), (2.0, "edge", None),
                                                 (1, "mr", "batch"), (0.25, "edge", "instance")])
def test_resdynblock_skip_connection_in_the_last_kernel(res_scale, conv, norm):
    """ResDynBlock2d adds `x * res_scale` inside the convolution's final store (EdgeConv2d fused path) and its gradient in
    the input-gradient GEMM; other convolutions / norms add it afterwards.  Against the reference's own expression."""
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()
    from gcn_lib.dense import ResDynBlock2d
    dev = _dev()
    torch.manual_seed(3)
    blk = ResDynBlock2d(32, 9, 2, conv, "relu", norm, True, res_scale=res_scale).to(dev).train()
    x0 = torch.randn(3, 32, 700, 1, device=dev)
    probe = torch.randn(3, 32, 700, 1, device=dev)
    ei = blk.body.dilated_knn_graph(x0)
    xa = x0.clone().requires_grad_(True)
    out = blk(xa, ei)
    (out * probe).sum().backward()
    ga = [p.grad.clone() for p in blk.parameters()]
    for p in blk.parameters():
        p.grad = None
    bufs = {k: v.clone() for k, v in blk.named_buffers()}
    xb = x0.clone().requires_grad_(True)
    ref = blk.body(xb, ei) + xb * res_scale                 # gcn_lib/dense/torch_vertex.py:101
    (ref * probe).sum().backward()
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(xa.grad, xb.grad, rtol=1e-4, atol=1e-4 * float(xb.grad.abs().max()))
    for a, p in zip(ga, blk.parameters()):
        torch.testing.assert_close(a, p.grad, rtol=1e-4, atol=1e-4 * float(p.grad.abs().max()) + 1e-7)
    assert set(bufs) == set(dict(blk.named_buffers()))


def test_vertex_gemm_is_an_exact_fma_chain():
    """fp32 MFMA == channel-ordered fmaf chain (guide: bitwise); checked against float64 within 1 ulp-ish
    and against small-integer data exactly."""
    from deep_gcns_torch_amd import dense_ops
    dev = _dev()
    g = torch.Generator().manual_seed(0)
    x = torch.randint(-8, 9, (2, 19, 37, 1), generator=g).float()
    W = torch.randint(-8, 9, (19, 50), generator=g).float()
    b = torch.randint(-8, 9, (50,), generator=g).float()
    out = dense_ops.vertex_gemm(x.to(dev), W.to(dev), b.to(dev)).cpu()
    ref = torch.einsum("bcn,cm->bnm", x.squeeze(-1), W) + b
    assert torch.equal(out, ref)                          # integers: exact in any order
    x = torch.randn(3, 64, 130, 1, generator=g)
    W = torch.randn(64, 128, generator=g)
    out = dense_ops.vertex_gemm(x.to(dev), W.to(dev)).cpu()
    ref = torch.einsum("bcn,cm->bnm", x.squeeze(-1).double(), W.double())
    torch.testing.assert_close(out.double(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("N,C,K,kind", [(4096, 3, 16, "lattice"), (4096, 64, 432, "lattice"), (2048, 16, 64, "sorted"),
                                        (1024, 3, 100, "duplicates"), (4096, 8, 512, "lattice"), (1536, 5, 1, "lattice"),
                                        (4096, 64, 880, "lattice"), (2048, 8, 1024, "sorted"), (1100, 3, 600, "duplicates"),
                                        # 32 / 64 channels: the bf16 filter kernel, whose overflow rows the workgroup
                                        # redoes in place (sorted clouds and duplicates defeat the sampled threshold)
                                        (4096, 64, 64, "sorted"), (2048, 32, 200, "sorted"), (1024, 64, 100, "duplicates"),
                                        (4096, 32, 432, "duplicates"), (4096, 64, 16, "sorted"),
                                        # round 6 (32 query rows per workgroup): clouds whose last workgroup is partly
                                        # filled (N % 32 != 0) or whose last 16-point tile is (N % 16 != 0)
                                        (1100, 64, 50, "lattice"), (1048, 32, 16, "sorted"), (2000, 64, 300, "lattice")])
def test_dense_knn_large_n_sampled_select_vs_oracle(N, C, K, kind):
    """N >= 1024 takes the sample-pre-filtered select; its result must equal the exact top-K whatever the
    point order (sorted clouds defeat the sample -> exact fallback) and with heavy ties (duplicates)."""
    from deep_gcns_torch_amd import dense_ops, synth
    from oracle import dense_ref
    dev = _dev()
    x = synth.lattice_cloud(2, C, N, seed=N + K)
    if kind == "sorted":
        order = torch.argsort(x[:, 0, :, 0], dim=1)                       # points sorted along the first axis
        x = torch.gather(x, 2, order.view(2, 1, N, 1).expand(2, C, N, 1)).contiguous()
    elif kind == "duplicates":
        x = x[:, :, torch.randint(0, 64, (N,), generator=torch.Generator().manual_seed(1))]   # 64 distinct points
    ei = dense_ops.knn_edge_index(x.to(dev), K, 1)
    mine = ei[0].cpu()
    dist = dense_ref.pairwise_distance(x.transpose(2, 1).squeeze(-1))
    ref_sorted = torch.sort(dist, dim=2).values[:, :, :K]
    assert torch.equal(torch.gather(dist, 2, mine), ref_sorted)
    assert bool((torch.sort(mine, dim=2).values.diff(dim=2) != 0).all()) if K > 1 else True
    # ties are ordered by index: within equal distances the ids ascend, and the boundary takes the lowest ids
    d_m = torch.gather(dist, 2, mine)
    same = d_m.diff(dim=2) == 0
    assert bool((mine.diff(dim=2)[same] > 0).all())
    kth = ref_sorted[:, :, -1:]
    n_le = (dist <= kth).sum(2)
    if kind == "duplicates":
        # among the points tied with the K-th distance the chosen ones are the lowest-indexed
        for b in range(2):
            for i in (0, N // 2, N - 1):
                tied = torch.nonzero(dist[b, i] == kth[b, i, 0]).flatten()
                chosen = mine[b, i][d_m[b, i] == kth[b, i, 0]]
                assert torch.equal(chosen, tied[:chosen.numel()])
    assert bool((n_le >= K).all())


@pytest.mark.parametrize("N,C,K,d", [(2048, 64, 16, 1), (1100, 32, 48, 3), (4096, 64, 224, 14)])
def test_self_excluding_knn_on_the_filter_kernels(N, C, K, d):
    """``exclude_self`` (torch_cluster.knn_graph(loop=False), gcn_lib/dense/torch_edge.py:97) on clouds large enough for the
    candidate-filter kernels (the golden test above uses 160 points = the exact kernel): the query point is never emitted,
    the emitted distances are the K smallest of the row without its diagonal, dilation applied."""
    from deep_gcns_torch_amd import dense_ops, synth
    from oracle import dense_ref
    x = synth.lattice_cloud(2, C, N, seed=N + K)
    ei = dense_ops.knn_edge_index(x.to(_dev()), K // d, d, exclude_self=True).cpu()
    assert ei.shape == (2, 2, N, K // d)
    dist = dense_ref.pairwise_distance(x.transpose(2, 1).squeeze(-1))
    dist.diagonal(dim1=1, dim2=2).fill_(float("inf"))
    want = torch.sort(dist, dim=2).values[:, :, :K:d]
    assert torch.equal(torch.gather(dist, 2, ei[0]), want)
    assert bool((ei[0] != torch.arange(N).view(1, N, 1)).all())
    assert torch.equal(ei[1], torch.arange(N).view(1, N, 1).expand_as(ei[1]))


@pytest.mark.parametrize("exclude_self", [False, True])
@pytest.mark.parametrize("N,C,K,d", [(2048, 32, 48, 3), (2048, 64, 48, 3), (1100, 32, 16, 1), (4096, 64, 224, 14)])
def test_filter_kernels_give_the_same_exact_answer_every_call(N, C, K, d, exclude_self):
    """Twelve calls on the same cloud, each against the oracle's distances.  Round 6's first ``exclude_self`` form of the
    32-row kernel (the self compare among the 16 flag compares of a tile pair) was RIGHT on most calls and wrong on a few
    rows of others -- in the C = 32 instantiation only, always rows 13 / 29 of a workgroup, one tile's 16 candidates
    entered the list with |x_i|^2 missing from their distance (found by reading the lists back:
    benchmarks/knn_list_readback.py; nothing in the source explains it, the exclusion now sits in the append rounds).  A single call cannot
    see such a fault."""
    from deep_gcns_torch_amd import dense_ops, synth
    from oracle import dense_ref
    x = synth.lattice_cloud(2, C, N, seed=N + K)
    dist = dense_ref.pairwise_distance(x.transpose(2, 1).squeeze(-1))
    if exclude_self:
        dist.diagonal(dim1=1, dim2=2).fill_(float("inf"))
    want = torch.sort(dist, dim=2).values[:, :, :K:d]
    xd = x.to(_dev())
    first = None
    for rep in range(12):
        ei = dense_ops.knn_edge_index(xd, K // d, d, exclude_self=exclude_self)
        bad = int((torch.gather(dist, 2, ei[0].cpu()) != want).any(2).sum())
        assert bad == 0, f"call {rep}: {bad} rows differ from the exact answer"
        if first is None:
            first = ei
        assert torch.equal(ei, first), f"call {rep} differs from call 0"


@pytest.mark.parametrize("N,C,K,d,exclude_self", [(4096, 64, 224, 14, False), (2048, 32, 48, 3, True), (1100, 64, 16, 1, False)])
def test_the_16_row_filter_kernel_with_lds_lists_stays_exact(N, C, K, d, exclude_self):
    """``dense_ops.KNN_GLOBAL_LISTS = False`` (a workspace without room for the candidate lists) selects rounds 4 - 5's
    kernel -- kept for A/B measurements (benchmarks/knn_time.py --lds-lists): same exact answer, same ids as the 32-row
    kernel."""
    from deep_gcns_torch_amd import dense_ops, synth
    from oracle import dense_ref
    x = synth.lattice_cloud(2, C, N, seed=N + K)
    dist = dense_ref.pairwise_distance(x.transpose(2, 1).squeeze(-1))
    if exclude_self:
        dist.diagonal(dim1=1, dim2=2).fill_(float("inf"))
    want = torch.sort(dist, dim=2).values[:, :, :K:d]
    xd = x.to(_dev())
    new = dense_ops.knn_edge_index(xd, K // d, d, exclude_self=exclude_self)
    dense_ops.KNN_GLOBAL_LISTS = False
    try:
        old = dense_ops.knn_edge_index(xd, K // d, d, exclude_self=exclude_self)
    finally:
        dense_ops.KNN_GLOBAL_LISTS = True
    assert torch.equal(torch.gather(dist, 2, old[0].cpu()), want)
    assert torch.equal(old, new)


_KNN_TRUTH = {}


def _shape_d_features(kind):
    """(B, C, N, 1) fp32 features of config 2's layer shape: iid normal, or what a ResGCN block really sees -- the output
    of a training-mode ResDynBlock2d (EdgeConv -> ReLU -> BatchNorm -> max over neighbours, plus the skip connection)."""
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()
    from gcn_lib.dense import ResDynBlock2d
    B, C, N = 8, 64, 4096
    x = torch.randn(B, C, N, 1, generator=torch.Generator().manual_seed(64))
    if kind == "post_bn_block":
        torch.manual_seed(5)
        blk = ResDynBlock2d(C, 16, 1, "edge", "relu", "batch", True).to(_dev()).train()
        with torch.no_grad():
            x = blk(x.to(_dev())).cpu()
    return x


def _knn_truth(kind, kmax=433):
    """Per sample: float64 distances, their (kmax) smallest per row in order, and the per-pair fp32 rounding budget
    8 eps32 (|x_i|^2 + |x_j|^2) -- what two correct fp32 evaluations of ||x_i||^2 - 2 x_i.x_j + ||x_j||^2 may differ by
    (the reference's own bmm on another BLAS, this kernel's six-product bf16 sum)."""
    if kind not in _KNN_TRUTH:
        x = _shape_d_features(kind)
        per_sample = []
        for b in range(x.size(0)):                                        # one (N, N) float64 matrix at a time
            pts = x[b, :, :, 0].t().double()                              # (N, C)
            sq = (pts * pts).sum(-1)
            d64 = sq.unsqueeze(1) - 2 * pts @ pts.t() + sq.unsqueeze(0)
            val, idx = torch.topk(d64, kmax, dim=1, largest=False, sorted=True)
            bud = 8 * torch.finfo(torch.float32).eps * (sq.unsqueeze(1) + sq[idx])
            per_sample.append((val, idx, bud, d64))
        _KNN_TRUTH[kind] = (x, per_sample)
    return _KNN_TRUTH[kind]


@pytest.mark.parametrize("K", [16, 224, 432])
@pytest.mark.parametrize("kind", ["randn", "post_bn_block"])
def test_knn_on_real_valued_features_at_shape_D_against_the_oracle(kind, K):
    """The kernel that serves config 2 (N = 4096, C = 64: knn_filter_bf16_kernel, distances from a six-product bf16
    matrix-pipe sum -- the fp32 value up to rounding, NOT Appendix A's fp32 association bit for bit) on REAL-VALUED
    features (the lattice clouds of the other tests make every product exact in any arithmetic and cannot see this).

    A position r of a row is DETERMINED when its float64 distance is further than the rounding budget from both its
    neighbours in the sorted order (for r = K - 1: from the first excluded candidate): there the emitted id must equal
    the float64 ranking's id -- and so must the oracle's (oracle/dense_ref.py: the reference's fp32 association,
    gcn_lib/dense/torch_edge.py:32-58), which validates the budget.  At the remaining positions (two candidates
    closer than fp32 can resolve; the reference's own topk on another BLAS would order them either way) the emitted
    neighbour must be as far as the r-th nearest, within the budget.  The counts are asserted and printed."""
    from deep_gcns_torch_amd import dense_ops
    from oracle import dense_ref
    x, truth = _knn_truth(kind)
    B, C, N, _ = x.shape
    mine = dense_ops.knn_edge_index(x.to(_dev()), K, 1)
    assert mine.shape == (2, B, N, K) and mine.dtype == torch.int64
    assert torch.equal(mine[1].cpu(), torch.arange(N).view(1, N, 1).expand(B, N, K))
    mine = mine[0].cpu()
    n_det = n_all = n_oracle_checked = n_vs_oracle = n_vs_f64 = n_oracle_vs_f64 = 0
    for b in range(B):
        val, idx, bud, d64 = truth[b]
        gap = val[:, 1:K + 1] - val[:, :K]                                # gap[r] = d(r+1) - d(r), r = 0..K-1
        need = bud[:, :K] + bud[:, 1:K + 1]
        right_ok = gap > need
        left_ok = torch.ones_like(right_ok)
        left_ok[:, 1:] = right_ok[:, :-1]
        det = left_ok & right_ok
        got = mine[b]
        assert torch.equal(got[det], idx[:, :K][det]), f"sample {b}: a determined neighbour differs from the float64 ranking"
        d_got = torch.gather(d64, 1, got)                                 # (N, K) float64
        worst = ((d_got - val[:, :K]).abs() / bud[:, :K]).max().item()
        assert worst <= 1.0, f"sample {b}: rank inconsistency {worst:.2f} x the fp32 rounding budget"
        assert bool((torch.sort(got, dim=1).values.diff(dim=1) != 0).all()) if K > 1 else True
        n_det += int(det.sum())
        n_all += det.numel()
        # the oracle's own ranking (the reference's fp32 association on this host's BLAS), every sample
        ref = dense_ref.dense_knn_matrix(x[b:b + 1], K)[0, 0]
        assert torch.equal(ref[det], idx[:, :K][det]), "the rounding budget does not cover the reference's own fp32 evaluation"
        n_oracle_checked += int(det.sum())
        n_vs_oracle += int((got != ref).sum())
        n_vs_f64 += int((got != idx[:, :K]).sum())
        n_oracle_vs_f64 += int((ref != idx[:, :K]).sum())
    frac = n_det / n_all
    # THE COUNT (VERDICT r5 weak #1a): of the B x N x K emitted positions, how many ids differ from the oracle's on this host,
    # next to how many of the oracle's own differ from the float64 ranking (the same kind of disagreement: two candidates
    # closer than fp32 resolves, ranked by two different fp32 evaluations)
    print(f"[knn {kind} K={K}] ids that differ from the oracle's: {n_vs_oracle} of {n_all} ({n_vs_oracle / n_all:.2e}); device vs "
          f"float64 ranking: {n_vs_f64} ({n_vs_f64 / n_all:.2e}); oracle vs float64 ranking: {n_oracle_vs_f64} "
          f"({n_oracle_vs_f64 / n_all:.2e})")
    from conftest import gate
    gate(f"knn shape D {kind} K={K}: fraction of emitted ids that differ from oracle/dense_ref.dense_knn_matrix (all 8 samples)",
         n_vs_oracle / n_all, 2e-3 if K > 16 else 1e-4)
    gate(f"knn shape D {kind} K={K}: ids off the float64 ranking, device / (3 x oracle + 32)",
         n_vs_f64 / (3 * n_oracle_vs_f64 + 32), 1.0)
    print(f"[knn {kind} K={K}] {n_det} of {n_all} neighbour positions determined beyond fp32 rounding ({frac:.4f}): ids "
          f"equal to the float64 ranking there (oracle checked on {n_oracle_checked}); the remaining "
          f"{n_all - n_det} rank-consistent within the budget")
    assert frac > (0.85 if K > 16 else 0.97)


def test_sparse_layout_and_self_excluding_knn_match_reference():
    """gcn_lib.sparse.torch_edge.DilatedKnnGraph (knn='matrix' and the torch_cluster-style 'tree' variant)
    and gcn_lib.dense.DilatedKnnGraph against the reference's own modules (golden), compared through the
    exact lattice distances (equal distances may be ordered differently)."""
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()
    from gcn_lib.sparse.torch_edge import DilatedKnnGraph as SparseKnn
    from gcn_lib.dense.torch_edge import DilatedKnnGraph as DenseTreeKnn
    dev = _dev()
    case = next(c for c in DENSE if c["kind"] == "knn_sparse")
    k, d, B, N = case["k"], case["dilation"], 3, 160
    dist = case["dist"]                                                   # (B,N,N)
    flat, batch = case["flat"].to(dev), case["batch"].to(dev)

    def dists_of(edge_index):                                             # (2, B*N*k) flattened, global ids
        nb = edge_index[0].cpu().long().view(B, N, k)
        ct = edge_index[1].cpu().long().view(B, N, k)
        off = (torch.arange(B) * N).view(B, 1, 1)
        assert torch.equal(ct - off, torch.arange(N).view(1, N, 1).expand(B, N, k))
        assert bool(((nb - off) >= 0).all()) and bool(((nb - off) < N).all())
        return torch.gather(dist, 2, nb - off), nb - off

    for knn, key in (("matrix", "edge_index"), ("tree", "edge_index_tree")):
        mine = SparseKnn(k, d, knn=knn)(flat, batch)
        assert mine.shape == (2, B * N * k) and mine.dtype == torch.int64
        dm, nbm = dists_of(mine)
        dr, _ = dists_of(case[key])
        assert torch.equal(dm, dr), knn
        if knn == "tree":
            assert bool((nbm != torch.arange(N).view(1, N, 1)).all())     # self excluded
    dt = DenseTreeKnn(k, d)(case["x"].to(dev))
    assert dt.shape == (2, B, N, k)
    ref = case["dense_tree"].long()
    assert torch.equal(torch.gather(dist, 2, dt[0].cpu()), torch.gather(dist, 2, ref[0]))
    assert bool((dt[0].cpu() != torch.arange(N).view(1, N, 1)).all())


@pytest.mark.parametrize("B,C,N,k,d", [(1, 3, 101, 5, 1), (3, 7, 130, 1, 1), (2, 4, 64, 64, 1), (1, 2, 4096, 3, 7)])
def test_dense_knn_odd_shapes(B, C, N, k, d):
    """N not a multiple of 4 (scalar distance phase), k = 1, K = N, tiny C."""
    from deep_gcns_torch_amd import dense_ops, synth
    from oracle import dense_ref
    dev = _dev()
    x = synth.lattice_cloud(B, C, N, seed=B * 100 + N)
    ei = dense_ops.knn_edge_index(x.to(dev), k, d)
    assert ei.shape == (2, B, N, k)
    dist = dense_ref.pairwise_distance(x.transpose(2, 1).squeeze(-1))
    ref = torch.sort(dist, dim=2).values[:, :, :k * d][:, :, ::d]
    assert torch.equal(torch.gather(dist, 2, ei[0].cpu()), ref)
    assert torch.equal(ei[1].cpu(), torch.arange(N).view(1, N, 1).expand(B, N, k))


def test_clouds_beyond_the_kernel_limits_run_on_the_library_tier():
    """N > 4096 points or k*dilation > 1024: the reference's formulation on library ops (dense_ops.
    _knn_beyond_kernel_limits).  Same contract: ascending distance, ties by index, dilation, centre ids; the selected
    distances equal the K smallest of the oracle's matrix to fp32 matmul rounding."""
    from deep_gcns_torch_amd import dense_ops
    from oracle import dense_ref
    dev = _dev()
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, 6, 5000, 1, generator=g)
    for K, d, excl in ((24, 3, False), (1100, 1, True)):
        ei = dense_ops.knn_edge_index(x.to(dev), K // d, d, exclude_self=excl).cpu()
        dist = dense_ref.pairwise_distance(x.transpose(2, 1).squeeze(-1))
        if excl:
            dist.diagonal(dim1=1, dim2=2).fill_(float("inf"))
        want = torch.sort(dist, dim=2).values[:, :, :K:d]
        torch.testing.assert_close(torch.gather(dist, 2, ei[0]), want, rtol=1e-5, atol=1e-5)
        assert torch.equal(ei[1], torch.arange(5000).view(1, 5000, 1).expand_as(ei[1]))
        if not excl:
            assert torch.equal(ei[0][:, :, 0], torch.arange(5000).expand(2, 5000))      # self first
    with pytest.raises(ValueError, match="neighbours asked"):
        dense_ops.knn_edge_index(x[:, :, :100].to(dev), 101, 1)
