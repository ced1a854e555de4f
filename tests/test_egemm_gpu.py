"""GPU parity of the fused edge-GEMM + aggregation kernel (csrc/gen_aggr_egemm.hip, dgcn_gen_aggr_egemm_fwd_f32)
against the oracle's statement of what the reference does for a GENConv with ``encode_edge=True`` on wide edge
features (gcn_lib/sparse/torch_vertex.py:56-68,78-85):

    edge_emb = Linear(edge_feat_dim -> C)(edge_attr);  m = relu(x_j + edge_emb) + eps;  out = AGGR(m)

Outputs and every gradient (x, edge features, encoder weight and bias, learnable t / p) for all aggregators, the
channel / feature widths of the reference's models (hidden 64, 80, 128, 224, 256 with group 2), strided
per-group views of the model-level embedding, and the graph shapes that stress the item / partial-row logic
(hub rows spanning many 64-edge items, empty rows anywhere, fewer than 64 edges, pre-sorted edge lists)."""
import pytest
import torch

from deep_gcns_torch_amd import synth

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _ref(x, ei, feat, W, b, n, aggr, kw):
    from oracle import sparse_ref
    emb = torch.nn.functional.linear(feat, W, b)
    return sparse_ref.gen_propagate(x, ei, emb, aggr=aggr, dim_size=n, **kw)


def _run_case(ei, n, C, K, aggr, kw, seed=0, strided=False, bias=True, add_root=False, rtol=1e-4, dead_channels=0,
              attribute=False):
    """``attribute``: at the cluster shape (E x C = 88 M pre-activations) about one pre-activation in 1e6 lies within
    fp32 rounding of the ReLU kink -- z = x_j + W f_e + b lands on either side depending on the summation order of the
    GEMM (six-product matrix-pipe sum here, a BLAS on the host), that edge then passes its gradient in one evaluation
    and not in the other -- and max aggregation meets near-ties.  The forward is replayed in float64 on the host, the
    (edge, channel) pairs on a kink / in a tie are marked, and every gradient element is held to the rounding tolerance
    PLUS the |gradient terms| of the marked pairs that feed it (tests/attribution.py): zero for all but a few hundred
    elements, and derived from the data, not a budget."""
    from deep_gcns_torch_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(seed)
    E = ei.size(1)
    x = torch.randn(n, C, generator=g)
    W = torch.randn(C, K, generator=g) / K ** 0.5
    b = torch.randn(C, generator=g) if bias else None
    if dead_channels:               # these channels see z < 0 on every edge: m = eps, the relu floor
        b[:dead_channels] = -1e3
    probe = torch.randn(n, C, generator=g)
    if strided:                     # (E, K) chunk of an (E, 2K) embedding: row stride 2K, offset K
        full = torch.randn(E, 2 * K, generator=g)
    else:
        full = torch.randn(E, K, generator=g)
    learn_t = bool(kw.get("learn_t"))
    learn_p = bool(kw.pop("learn_p", False))

    def leaves(device):
        xs = x.clone().to(device).requires_grad_(True)
        fs = full.clone().to(device).requires_grad_(True)
        Ws = W.clone().to(device).requires_grad_(True)
        bs = None if b is None else b.clone().to(device).requires_grad_(True)
        k2 = dict(kw)
        if learn_t:
            k2["t"] = torch.tensor([kw["t"]], device=device, requires_grad=True)
        if learn_p:
            k2["p"] = torch.tensor([kw["p"]], device=device, requires_grad=True)
        return xs, fs, Ws, bs, k2

    xr, fr, Wr, br, kr = leaves("cpu")
    featr = fr[:, K:] if strided else fr
    ref = _ref(xr, ei, featr, Wr, br, n, aggr, kr)
    if add_root:
        ref = ref + xr
    (ref * probe).sum().backward()

    xd, fd, Wd, bd, kd = leaves(dev)
    featd = fd[:, K:] if strided else fd
    assert ops.encoder_fusable(xd, featd, Wd)
    if learn_p:
        kd["learn_p"] = True
    out = ops.gen_aggregate(xd, ei.to(dev), featd, aggr=aggr, edge_encoder=(Wd, bd), dim_size=n,
                            add_root=add_root, **kd)
    (out * probe.to(dev)).sum().backward()

    extra = {}
    if attribute:
        import attribution
        bounds = attribution.sparse_flip_bounds(x, ei, full[:, K:] if strided else full, W, b, n, aggr, probe,
                                                t=kw.get("t", 1.0), p=kw.get("p", 1.0), learn_t=learn_t)
        # the marking is neither empty nor degenerate: a few pairs per million
        assert 0 < bounds["n_kink"] <= 2e-5 * bounds["n_pairs"], bounds["n_kink"]
        rows, vals = bounds["grad_feat"]
        if strided:                 # the gradient of the (E, 2K) parent: nothing reaches the other group's columns
            vals = torch.cat([torch.zeros_like(vals), vals], dim=1)
        extra = dict(grad_x=bounds["grad_x"], grad_feat=(rows, vals), grad_W=bounds["grad_W"], grad_b=bounds["grad_b"])

    def close(a, r, what, rt=rtol, scale_atol=2e-5):
        r = r.to(torch.float32)
        atol = scale_atol * max(float(r.abs().max()), 1e-6)
        a = a.detach().cpu()
        if attribute:
            import attribution
            attribution.assert_explained(a, r, extra.get(what), rt, atol, what)
            return
        torch.testing.assert_close(a, r, rtol=rt, atol=atol, msg=lambda m: f"{what}: {m}")

    close(out, ref.detach(), "out")
    close(xd.grad, xr.grad, "grad_x", 2e-4)
    close(fd.grad, fr.grad, "grad_feat", 2e-4)
    close(Wd.grad, Wr.grad, "grad_W", 5e-4, 1e-4)
    if b is not None:
        close(bd.grad, br.grad, "grad_b", 5e-4, 1e-4)
    if learn_t:
        close(kd["t"].grad, kr["t"].grad, "grad_t", 1e-3, 1e-4)
    if learn_p:
        close(kd["p"].grad, kr["p"].grad, "grad_p", 1e-3, 1e-4)
    return out


AGGRS = [("softmax", dict(t=0.7)), ("softmax", dict(t=0.9, learn_t=True)), ("softmax_sg", dict(t=0.1)),
         ("power", dict(p=2.0)), ("power", dict(p=1.0)), ("power", dict(p=1.5, learn_p=True)), ("power", dict(p=1.0, learn_p=True)),
         ("max", {}), ("add", {}), ("mean", {})]


@pytest.mark.parametrize("aggr,kw", AGGRS, ids=lambda v: v if isinstance(v, str) else "-".join(f"{k}{x}" for k, x in v.items()))
def test_all_aggregators_revgnn_wide_width(aggr, kw):
    """hidden = 224, group = 2: K = 224 features, C = 112 channels; 2100-edge hub row, isolated nodes, duplicates."""
    _run_case(synth.tricky_graph(), 257, 112, 224, aggr, dict(kw), seed=1)


@pytest.mark.parametrize("C,K", [(32, 64), (40, 80), (64, 128), (128, 256), (64, 64), (20, 48), (4, 16), (112, 224)])
@pytest.mark.parametrize("aggr,kw", [("max", {}), ("power", dict(p=1.0, learn_p=True)), ("softmax", dict(t=1.0))])
def test_widths_of_the_reference_models(C, K, aggr, kw):
    """hidden 64 / 80 (RevGNN-Deep) / 128 / 256 with two groups, ungrouped hidden 64, odd tiles (C = 20: partly
    filled 16-channel tile; K = 48: half-filled 32-float chunk), the smallest shape, and a strided group view."""
    small = synth.tricky_graph(n=64, e=700, hub_deg=300, seed=7)
    _run_case(small, 64, C, K, aggr, dict(kw), seed=C + K, strided=(C == 112 or C == 32))


def test_strided_group_view_no_bias_and_fused_root():
    _run_case(synth.tricky_graph(), 257, 112, 224, "max", {}, seed=3, strided=True, bias=False, add_root=True)
    _run_case(synth.tricky_graph(), 257, 32, 64, "softmax_sg", dict(t=0.1), seed=4, strided=True, add_root=True)
    _run_case(synth.tricky_graph(), 257, 32, 64, "power", dict(p=1.0), seed=5, add_root=True)


def _graph_cases():
    g = torch.Generator().manual_seed(11)
    out = {}
    out["three_edges"] = (torch.tensor([[0, 1, 2], [3, 3, 1]]), 5)
    out["one_edge"] = (torch.tensor([[4], [2]]), 6)
    out["exactly_64"] = (torch.stack([torch.randint(0, 10, (64,), generator=g), torch.randint(0, 10, (64,), generator=g)]), 10)
    out["exactly_128_sorted"] = (torch.stack([torch.randint(0, 40, (128,), generator=g),
                                              torch.sort(torch.randint(0, 40, (128,), generator=g)).values]), 40)
    out["single_hub_all_items"] = (torch.stack([torch.randint(0, 50, (1000,), generator=g), torch.full((1000,), 17)]), 50)
    dst = torch.randint(100, 200, (3000,), generator=g)            # rows 0..99 and 200..299 are empty
    out["empty_head_and_tail"] = (torch.stack([torch.randint(0, 300, (3000,), generator=g), dst]), 300)
    # item boundaries that coincide with row boundaries: 20 rows of exactly 64 edges
    dst = torch.arange(20).repeat_interleave(64)
    out["rows_aligned_to_items"] = (torch.stack([torch.randint(0, 20, (1280,), generator=g), dst[torch.randperm(1280, generator=g)]]), 20)
    # degree-1 rows (many row changes inside one batch) followed by a long row
    dst = torch.cat([torch.arange(90), torch.full((200,), 95)])
    out["unit_rows_then_long"] = (torch.stack([torch.randint(0, 100, (290,), generator=g), dst]), 100)
    return out


@pytest.mark.parametrize("name", list(_graph_cases()))
@pytest.mark.parametrize("aggr,kw", [("softmax", dict(t=0.5)), ("max", {}), ("power", dict(p=2.0)), ("mean", {})])
def test_item_and_partial_row_logic(name, aggr, kw):
    ei, n = _graph_cases()[name]
    _run_case(ei, n, 32, 64, aggr, dict(kw), seed=21)


def test_bit_reproducible_and_matches_unfused_path():
    """No atomics: two launches agree bit for bit; and the fused kernel agrees with this package's own unfused path
    (stock GEMM + (E, C) embedding + aggregation kernel) to fp32 rounding."""
    from deep_gcns_torch_amd import ops
    dev = _dev()
    s = synth.SHAPES["proteins_cluster"]
    ei = synth.powerlaw_graph(s["n"], s["n_undirected"], s["seed"], device=dev)
    E = ei.size(1)
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(s["n"], 112, device=dev, generator=g)
    feat = torch.randn(E, 448, device=dev, generator=g)[:, :224]
    W = torch.randn(112, 224, device=dev, generator=g) / 15
    b = torch.randn(112, device=dev, generator=g)
    for aggr, kw in (("max", {}), ("power", dict(p=1.0)), ("softmax", dict(t=1.0))):
        with torch.no_grad():
            a = ops.gen_aggregate(x, ei, feat, aggr=aggr, edge_encoder=(W, b), **kw)
            a2 = ops.gen_aggregate(x, ei, feat, aggr=aggr, edge_encoder=(W, b), **kw)
            u = ops.gen_aggregate(x, ei, torch.nn.functional.linear(feat, W, b), aggr=aggr, **kw)
        assert torch.equal(a, a2)
        torch.testing.assert_close(a, u, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("C,K", [(112, 224), (32, 64), (128, 256), (20, 48)])
def test_max_winner_backward_equals_the_dense_route(C, K):
    """csrc/egemm_max_bwd.hip (walk the (row, channel) winners; no (E, C) gradient) against the dense route (dz written,
    dz @ W and dz^T F on the matrix pipe) from the same forward: every gradient agrees to fp32 rounding; into a running
    sink (ops.edge_grad_sink, what the reversible backward uses) the winners' rows receive the same sums and the rows of
    edges that win nothing keep their bits."""
    from deep_gcns_torch_amd import ops
    dev = _dev()
    n = 3000
    ei = synth.powerlaw_graph(n, 40_000, seed=11).to(dev)
    E = ei.size(1)
    g = torch.Generator().manual_seed(C)
    x = torch.randn(n, C, generator=g).to(dev)
    full = torch.randn(E, 2 * K, generator=g).to(dev)
    W = (torch.randn(C, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(C, generator=g)
    b[:3] = -1e3                                   # three channels at the relu floor everywhere: arg-max id -1
    b = b.to(dev)
    probe = torch.randn(n, C, generator=g).to(dev)
    saved = ops.EGEMM_MAX_WINNER_BWD

    def run(winner, sink=None):
        ops.EGEMM_MAX_WINNER_BWD = winner
        try:
            xs, fs = x.clone().requires_grad_(True), full.clone().requires_grad_(True)
            Ws, bs = W.clone().requires_grad_(True), b.clone().requires_grad_(True)
            feat = fs[:, K:]
            if sink is None:
                out = ops.gen_aggregate(xs, ei, feat, aggr="max", edge_encoder=(Ws, bs), dim_size=n, add_root=True)
                (out * probe).sum().backward()
                return out.detach(), xs.grad, fs.grad[:, K:], Ws.grad, bs.grad
            with ops.edge_grad_sink(feat, sink):
                out = ops.gen_aggregate(xs, ei, feat, aggr="max", edge_encoder=(Ws, bs), dim_size=n, add_root=True)
                torch.autograd.grad((out * probe).sum(), [xs, Ws, bs, feat], allow_unused=True)
            return sink
        finally:
            ops.EGEMM_MAX_WINNER_BWD = saved

    dense = run(False)
    win = run(True)
    assert torch.equal(win[0], dense[0]) and torch.equal(win[1], dense[1]) and torch.equal(win[4], dense[4])
    for a, r, what in ((win[2], dense[2], "grad_feat"), (win[3], dense[3], "grad_W")):
        torch.testing.assert_close(a, r, rtol=1e-5, atol=2e-6 * float(r.abs().max()), msg=lambda m, what=what: f"{what}: {m}")
    assert torch.equal(fs_zero := (win[2] == 0).all(1), (dense[2] == 0).all(1)) and 0 < int(fs_zero.sum()) < E
    # the running sum of the reversible backward
    before = torch.randn(E, K, generator=g).to(dev)
    after = run(True, before.clone())
    torch.testing.assert_close(after - before, win[2], rtol=1e-4, atol=1e-5 * float(win[2].abs().max()))
    assert torch.equal(after[fs_zero], before[fs_zero])


@pytest.mark.parametrize("aggr,kw", [("max", {}), ("softmax", dict(t=0.7)), ("power", dict(p=2.0))])
def test_channels_at_the_relu_floor(aggr, kw):
    """Channels whose pre-activation is negative on EVERY edge (m = eps everywhere): no gradient flows through them.
    Max runs its backward without saved pre-activations there -- the forward writes arg-max id -1 for such channels
    (csrc/gen_aggr_egemm.hip eg_finalize) -- so this is the case that would break if that marking were wrong."""
    out = _run_case(synth.tricky_graph(), 257, 112, 224, aggr, dict(kw), seed=4, dead_channels=20)
    assert out.shape == (257, 112)


@pytest.mark.parametrize("aggr,kw", [("max", {}), ("power", dict(p=1.0, learn_p=True)), ("softmax", dict(t=1.0))],
                         ids=["max", "power-p1-learn_p", "softmax"])
def test_config5_layer_at_the_cluster_shape_against_the_oracle(aggr, kw):
    """One GENConv aggregation of BASELINE config 5 at its real size -- ogbn-proteins cluster N = 13,253, E = 791,225
    (power-law), hidden 224 / group 2: K = 224 features read as the strided per-group view of the (E, 448) model-level
    embedding, C = 112 channels -- against oracle/sparse_ref.py on this box's host cores (the reference's
    edge_encoder -> message -> aggregate chain, gcn_lib/sparse/torch_vertex.py:56-68): output and the gradients of
    x, the edge features, the encoder weight and bias (and p)."""
    s = synth.SHAPES["proteins_cluster"]
    ei = synth.powerlaw_graph(s["n"], s["n_undirected"], s["seed"])
    assert ei.size(1) == 791225 and s["n"] == 13253
    _run_case(ei, s["n"], 112, 224, aggr, dict(kw), seed=5, strided=True, attribute=True)


def test_non_finite_and_tiny_operands_contract():
    """The six-product bf16 split (csrc/bf16x6.h) on special values.  Contract:
    * finite operands of any magnitude, down to 1e-30-scale features and weights, give the fp32 result (the bf16 matrix
      pipe keeps denormal fragments: nothing is flushed);
    * non-finite features or weights are OUTSIDE the contract.  What happens is defined and tested here so that nobody
      has to guess: inf - top16(inf) = NaN in the split, so the pre-activation z of an affected edge is NaN in every
      channel, relu (v_max_f32, IEEE maxNum) turns it into 0 and the edge contributes m = eps -- where the reference's
      fp32 GEMM gives z = +-inf, i.e. m = inf or eps per channel.  Edges that do not touch the value are unaffected."""
    from deep_gcns_torch_amd import ops
    dev = _dev()
    ei = synth.tricky_graph(n=64, e=700, hub_deg=300, seed=7)
    n, C, K, E = 64, 32, 64, ei.size(1)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(n, C, generator=g)
    feat = torch.randn(E, K, generator=g)
    W = torch.randn(C, K, generator=g) / 8
    b = torch.randn(C, generator=g)
    # tiny: scale features by 1e-30 and one weight row by 1e-8 -> z = x + (tiny) exactly as in fp32
    feat_t = feat * 1e-30
    W_t = W.clone()
    W_t[3] *= 1e-8
    ref = _ref(x, ei, feat_t, W_t, b, n, "add", {})
    out = ops.gen_aggregate(x.to(dev), ei.to(dev), feat_t.to(dev), aggr="add", edge_encoder=(W_t.to(dev), b.to(dev)), dim_size=n)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=1e-5)
    # non-finite: one edge's feature row carries +inf -> that edge's message is eps in every channel
    bad_edge = 10
    feat_i = feat.clone()
    feat_i[bad_edge, 5] = float("inf")
    out = ops.gen_aggregate(x.to(dev), ei.to(dev), feat_i.to(dev), aggr="add", edge_encoder=(W.to(dev), b.to(dev)), dim_size=n).cpu()
    keep = torch.ones(E, dtype=torch.bool)
    keep[bad_edge] = False
    expect = _ref(x, ei[:, keep], feat[keep], W, b, n, "add", {})
    expect[int(ei[1, bad_edge])] += 1e-7
    torch.testing.assert_close(out, expect, rtol=1e-4, atol=1e-5)
    assert not torch.isfinite(_ref(x, ei, feat_i, W, b, n, "add", {})).all()      # the reference: inf in that row
