"""GPU parity: libdgcn's fused aggregation (through the C ABI) vs
  - the golden vectors produced by the reference's own code, and
  - the CPU oracle on seeded graphs, incl. size-independent properties at larger sizes.
Tolerance: 1e-4 relative (BASELINE.json north_star) for fp32 aggregation outputs.
"""
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

RTOL = 1e-4
AGG = load_golden("sparse_aggregate.pt")


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_library_shares_torch_hip_runtime():
    from deep_gcns_torch_amd import ops
    ops.selftest(_dev())
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):       # non-default stream: launches must follow torch's current stream
        ops.selftest(_dev())
    s.synchronize()


def _run_case(case, graph_tensor):
    from deep_gcns_torch_amd import ops
    dev = _dev()
    kw = dict(case["kw"])
    aggr = case["aggr"]
    x = case["x"].to(dev).requires_grad_(True)
    ea = None if case["edge_attr"] is None else case["edge_attr"].to(dev).requires_grad_(True)
    learn_t = bool(kw.get("learn_t", False)) and aggr in ("softmax", "softmax_sum")
    learn_p = bool(kw.get("learn_p", False))
    t = torch.tensor([kw.get("t", 1.0)], device=dev, requires_grad=learn_t) if learn_t else kw.get("t", 1.0)
    p = torch.tensor([kw.get("p", 1.0)], device=dev, requires_grad=learn_p) if learn_p else kw.get("p", 1.0)
    out = ops.gen_aggregate(x, graph_tensor, ea, aggr=aggr, t=t, p=p, learn_t=learn_t, learn_p=learn_p,
                            dim_size=case["n"])
    y = None
    if aggr.endswith("_sum"):  # degree scaling is the caller's epilogue (torch_message.py:60-63)
        y = torch.tensor([float(kw.get("y", 0.0))], device=dev, requires_grad=True)
        deg = torch.bincount(graph_tensor[1], minlength=case["n"]).float().unsqueeze(1)
        out = torch.pow(deg, torch.sigmoid(y)) * out
    (out * case["probe"].to(dev)).sum().backward()
    return out, x, ea, t, p, y


@pytest.mark.parametrize("case", AGG["cases"], ids=lambda c: c["name"])
def test_matches_reference_golden(case):
    dev = _dev()
    ei = AGG["graphs"][case["graph"]].to(dev)
    out, x, ea, t, p, y = _run_case(case, ei)
    torch.testing.assert_close(out.detach().cpu(), case["out"], rtol=RTOL, atol=1e-6)
    gscale = case["grad_x"].abs().max().item()
    torch.testing.assert_close(x.grad.cpu(), case["grad_x"], rtol=RTOL, atol=1e-5 * max(gscale, 1.0))
    if ea is not None:
        torch.testing.assert_close(ea.grad.cpu(), case["grad_edge_attr"], rtol=RTOL, atol=1e-6)
    if "grad_t" in case:
        torch.testing.assert_close(t.grad.cpu(), case["grad_t"], rtol=1e-3, atol=1e-4)
    if "grad_p" in case:
        torch.testing.assert_close(p.grad.cpu(), case["grad_p"], rtol=1e-3, atol=1e-4)
    if "grad_y" in case and y is not None:
        torch.testing.assert_close(y.grad.cpu(), case["grad_y"], rtol=1e-3, atol=1e-4)


def test_sorted_edge_list_and_graph_object_give_same_result():
    from deep_gcns_torch_amd import ops
    from deep_gcns_torch_amd.graph import Graph
    dev = _dev()
    case = next(c for c in AGG["cases"] if c["name"] == "tricky64_softmax_sg")
    ei = AGG["graphs"]["tricky"]
    order = torch.sort(ei[1], stable=True).indices
    x = case["x"].to(dev)
    a = ops.gen_aggregate(x, ei.to(dev), aggr="softmax_sg")
    b = ops.gen_aggregate(x, ei[:, order].contiguous().to(dev), aggr="softmax_sg")
    c = ops.gen_aggregate(x, Graph.from_edge_index(ei.to(dev), 257), aggr="softmax_sg")
    assert torch.equal(a, b) and torch.equal(a, c)      # deterministic: bit-identical
    torch.testing.assert_close(a.cpu(), case["out"], rtol=RTOL, atol=1e-6)


def test_known_answers_on_device():
    from deep_gcns_torch_amd import ops
    dev = _dev()
    x = torch.tensor([[1.0] * 4, [3.0] * 4, [0.0] * 4], device=dev)
    ei = torch.tensor([[0, 1], [2, 2]], device=dev)   # messages 1 and 3 into node 2
    f = lambda **k: ops.gen_aggregate(x, ei, relu_eps=False, **k)
    assert f(aggr="softmax", t=1.0)[2, 0].item() == pytest.approx(2.761594, abs=1e-5)
    assert f(aggr="softmax", t=50.0)[2, 0].item() == pytest.approx(3.0, abs=1e-6)
    assert f(aggr="power", p=2.0)[2, 0].item() == pytest.approx(5 ** 0.5, abs=1e-5)
    assert f(aggr="mean")[2, 0].item() == 2.0 and f(aggr="add")[2, 0].item() == 4.0
    assert f(aggr="max")[2, 0].item() == 3.0
    for aggr in ("softmax", "add", "mean", "max"):
        assert f(aggr=aggr)[0, 0].item() == 0.0      # isolated node
    assert f(aggr="power", p=2.0)[0, 0].item() == pytest.approx((1e-7) ** 0.5, rel=1e-4)
    # message floor relu(x)+eps (torch_vertex.py:85)
    xm = torch.tensor([[-2.0] * 4, [0.5] * 4], device=dev)
    e1 = torch.tensor([[0], [1]], device=dev)
    assert ops.gen_aggregate(xm, e1, aggr="add")[1, 0].item() == pytest.approx(1e-7, rel=1e-6)


@pytest.mark.parametrize("C,with_ea", [(128, False), (64, True), (32, False), (32, True), (16, False), (16, True)])
@pytest.mark.parametrize("aggr,kw", [("softmax_sg", dict(t=0.1)), ("power", dict(p=2.0)), ("power", dict(p=1.0)), ("max", {}),
                                     ("mean", {})])
def test_vs_oracle_powerlaw_graph(aggr, kw, C, with_ea):
    """Mid-size power-law graph (hubs on both sides -> split rows in both walks) vs the CPU oracle.
    C = 32 / 16 take the sub-group kernels (2 / 4 rows side by side in one wave), with and without edge features.
    The oracle is evaluated in float64: on hub rows (1e4+ edges) the fp32 sequential scatter_add of
    the reference path itself carries ~1e-4 relative rounding noise, so fp32-vs-fp32 would compare
    two roundings; fp64 is the value both approximate.  Tolerance stays 1e-4 relative."""
    from deep_gcns_torch_amd import ops, synth
    from oracle import sparse_ref
    dev = _dev()
    n = 20000
    ei = synth.powerlaw_graph(n, 150_000, seed=9, exponent=2.1)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, C, generator=g)
    probe = torch.randn(n, C, generator=g)
    ea = torch.randn(ei.size(1), C, generator=g) if with_ea else None
    xr = x.double().requires_grad_(True)
    er = ea.double().requires_grad_(True) if with_ea else None
    ref = sparse_ref.gen_propagate(xr, ei, edge_attr=er, aggr=aggr, **kw)
    (ref * probe.double()).sum().backward()
    xd = x.to(dev).requires_grad_(True)
    ed = ea.to(dev).requires_grad_(True) if with_ea else None
    out = ops.gen_aggregate(xd, ei.to(dev), edge_attr=ed, aggr=aggr, **kw)
    (out * probe.to(dev)).sum().backward()
    torch.testing.assert_close(out.detach().cpu().double(), ref.detach(), rtol=RTOL, atol=1e-6)
    gs = xr.grad.abs().max().item()
    torch.testing.assert_close(xd.grad.cpu().double(), xr.grad, rtol=RTOL, atol=2e-6 * max(gs, 1.0))
    if with_ea:
        torch.testing.assert_close(ed.grad.cpu().double(), er.grad, rtol=RTOL, atol=2e-6 * max(gs, 1.0))
    # ... and against the reference's own fp32 path (same oracle in float32: sequential scatter_add on the CPU).
    # Its hub rows (1e4+ edges summed in index order) carry up to a few 1e-4 of relative rounding noise themselves,
    # so the gate here is 5e-4 relative on the outputs / 5e-4 of the gradient scale (VERDICT r1, weak 1 iv).
    x32 = x.clone().requires_grad_(True)
    e32 = ea.clone().requires_grad_(True) if with_ea else None
    ref32 = sparse_ref.gen_propagate(x32, ei, edge_attr=e32, aggr=aggr, **kw)
    (ref32 * probe).sum().backward()
    torch.testing.assert_close(out.detach().cpu(), ref32.detach(), rtol=5e-4, atol=1e-5)
    torch.testing.assert_close(xd.grad.cpu(), x32.grad, rtol=5e-4, atol=5e-4 * max(gs, 1.0))


@pytest.mark.parametrize("with_ea", [False, True])
@pytest.mark.parametrize("aggr,kw", [("softmax_sg", dict(t=0.1)), ("softmax", dict(t=0.8, learn_t=True)), ("power", dict(p=2.0)),
                                     ("max", {}), ("mean", {}), ("add", {})])
def test_low_degree_128_channel_rows_two_per_wave(aggr, kw, with_ea):
    """C = 128 on a low-degree graph with many rows (ogbn-arxiv's regime: < 32 edges per row, >= 24,576 rows) takes the
    two-rows-per-wave layout (LPR = 32, one edge group per row) in both walks; also the fp32 oracle comparison the
    power-law test leaves out (the fp64 oracle is not the reference's fp32 path) -- degrees here are
    small, so fp32 summation order costs ~1e-6."""
    from deep_gcns_torch_amd import ops, synth
    from oracle import sparse_ref
    dev = _dev()
    n, C = 30000, 128
    ei = synth.undirected_random_graph(n, 180_000, seed=21)           # 13 edges per row incl. self loops
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(n, C, generator=gen)
    probe = torch.randn(n, C, generator=gen)
    ea = 0.5 * torch.randn(ei.size(1), C, generator=gen) if with_ea else None
    kw = dict(kw)
    tr = td = None
    if kw.get("learn_t"):
        tr = torch.tensor([kw["t"]], requires_grad=True)
        td = torch.tensor([kw["t"]], device=dev, requires_grad=True)
    xr = x.clone().requires_grad_(True)
    er = ea.clone().requires_grad_(True) if with_ea else None
    ref = sparse_ref.gen_propagate(xr, ei, edge_attr=er, aggr=aggr, **(dict(kw, t=tr) if tr is not None else kw))
    (ref * probe).sum().backward()
    xd = x.to(dev).requires_grad_(True)
    ed = ea.to(dev).requires_grad_(True) if with_ea else None
    out = ops.gen_aggregate(xd, ei.to(dev), edge_attr=ed, aggr=aggr, **(dict(kw, t=td) if td is not None else kw))
    (out * probe.to(dev)).sum().backward()
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=RTOL, atol=1e-5)
    gs = float(xr.grad.abs().max())
    torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=RTOL, atol=1e-5 * max(gs, 1.0))
    if with_ea:
        torch.testing.assert_close(ed.grad.cpu(), er.grad, rtol=RTOL, atol=1e-5 * max(gs, 1.0))
    if td is not None:
        torch.testing.assert_close(td.grad.cpu(), tr.grad, rtol=1e-3, atol=1e-3 * float(tr.grad.abs().max()))


def test_arxiv_shape_properties():
    """Full ogbn-arxiv-shaped input: size-independent properties instead of a CPU replay."""
    from deep_gcns_torch_amd import ops, synth
    dev = _dev()
    s = synth.SHAPES["arxiv"]
    ei = synth.undirected_random_graph(s["n"], s["n_undirected"], s["seed"], device=dev)
    n, C = s["n"], s["channels"]
    assert ei.size(1) == 2_484_941
    x = torch.randn(n, C, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    sm = ops.gen_aggregate(x, ei, aggr="softmax_sg", t=0.1)
    mx = ops.gen_aggregate(x, ei, aggr="max")
    mn = ops.gen_aggregate(x, ei, aggr="mean")
    ad = ops.gen_aggregate(x, ei, aggr="add")
    deg = torch.bincount(ei[1], minlength=n).float().unsqueeze(1)
    # softmax aggregation is a convex combination: mean <= softmax(t>0) <= max  (t>0 tilts to max)
    assert torch.all(sm <= mx * (1 + 1e-5) + 1e-6) and torch.all(sm >= mn * (1 - 1e-5) - 1e-6)
    torch.testing.assert_close(ad, mn * deg, rtol=1e-4, atol=1e-5)
    # t -> 0 recovers the mean; the softmax-weighted mean is non-decreasing in t (d/dt = variance >= 0)
    torch.testing.assert_close(ops.gen_aggregate(x, ei, aggr="softmax_sg", t=1e-7), mn, rtol=1e-4, atol=1e-5)
    hot = ops.gen_aggregate(x, ei, aggr="softmax_sg", t=200.0)
    assert torch.all(hot >= sm * (1 - 1e-5) - 1e-6) and torch.all(hot <= mx * (1 + 1e-5) + 1e-6)
    # edge-order invariance is exact (same CSR after the stable sort of a dst-preserving shuffle)
    perm = torch.randperm(ei.size(1), device=dev)
    sm2 = ops.gen_aggregate(x, ei[:, perm].contiguous(), aggr="softmax_sg", t=0.1)
    torch.testing.assert_close(sm2, sm, rtol=1e-5, atol=1e-6)
    # linearity of 'add' backward: grad of sum(out) w.r.t. x = out-degree * relu'(x)
    xg = x.clone().requires_grad_(True)
    ops.gen_aggregate(xg, ei, aggr="add").sum().backward()
    odeg = torch.bincount(ei[0], minlength=n).float().unsqueeze(1)
    torch.testing.assert_close(xg.grad, odeg * (x > 0).float() * torch.ones(1, C, device=dev), rtol=1e-5, atol=1e-5)
    # run-to-run bit reproducibility
    assert torch.equal(ops.gen_aggregate(x, ei, aggr="softmax_sg", t=0.1), sm)


def test_products_shape_properties():
    """BASELINE config 4 at FULL size (N = 2,449,029, E = 126,167,309, C = 128): the CPU oracle cannot replay this, so
    parity is checked through size-independent properties -- convexity bounds, add = mean * degree, exact gradients
    of the linear aggregators, the softmax gradient identity sum_c-free check on sampled rows against a torch
    gather, bit reproducibility."""
    from deep_gcns_torch_amd import ops, synth
    from deep_gcns_torch_amd.graph import Graph
    dev = _dev()
    s = synth.SHAPES["products"]
    n, C = s["n"], s["channels"]
    ei = synth.undirected_random_graph(n, s["n_undirected"], s["seed"], device=dev)
    assert ei.size(1) == 126_167_309
    graph = Graph.from_edge_index(ei, n)
    x = torch.randn(n, C, device=dev, generator=torch.Generator(device=dev).manual_seed(2))
    sm = ops.gen_aggregate(x, graph, aggr="softmax_sg", t=0.1)
    mx = ops.gen_aggregate(x, graph, aggr="max")
    mn = ops.gen_aggregate(x, graph, aggr="mean")
    assert torch.all(sm <= mx * (1 + 1e-5) + 1e-6) and torch.all(sm >= mn * (1 - 1e-5) - 1e-6)
    del mx
    ad = ops.gen_aggregate(x, graph, aggr="add")
    torch.testing.assert_close(ad, mn * graph.deg.unsqueeze(1), rtol=1e-4, atol=1e-4)
    del ad, mn
    # sampled rows against a direct torch evaluation of the reference formula (softmax over the row's neighbours)
    rows = torch.randint(0, n, (64,), device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    rp = graph.rowptr.long()
    for r in rows.tolist():
        nb = graph.col[rp[r]:rp[r + 1]].long()
        if nb.numel() == 0:
            assert float(sm[r].abs().max()) == 0.0
            continue
        m = torch.relu(x[nb]).double() + 1e-7
        w = torch.softmax(0.1 * m, dim=0)
        torch.testing.assert_close(sm[r].double(), (w * m).sum(0), rtol=1e-4, atol=1e-7)
    # backward: 'add' is linear (grad = out-degree * relu'), softmax_sg gradient sums match a finite directional check
    xg = x.clone().requires_grad_(True)
    ops.gen_aggregate(xg, graph, aggr="add").sum().backward()
    odeg = (graph.t_rowptr[1:] - graph.t_rowptr[:-1]).float().unsqueeze(1)
    torch.testing.assert_close(xg.grad, odeg * (x > 0).float(), rtol=1e-5, atol=1e-5)
    xg.grad = None
    probe = torch.randn(n, C, device=dev, generator=torch.Generator(device=dev).manual_seed(4))
    out = ops.gen_aggregate(xg, graph, aggr="softmax_sg", t=0.1)
    (out * probe).sum().backward()
    # sampled SOURCE rows: softmax_sg treats the weights as constants (torch_message.py:55-58), so
    # grad_x[s] = relu'(x_s) * sum_{e: s -> i} w_e * probe_i  with  w_e = softmax over i's in-edges, evaluated with torch
    trp = graph.t_rowptr.long()
    for sidx in rows[:16].tolist():
        dsts = graph.t_col[trp[sidx]:trp[sidx + 1]].long()
        acc = torch.zeros(C, device=dev, dtype=torch.float64)
        ms = torch.relu(x[sidx]).double() + 1e-7
        for i in dsts.tolist():
            nb = graph.col[rp[i]:rp[i + 1]].long()
            m = torch.relu(x[nb]).double() + 1e-7
            lse = torch.logsumexp(0.1 * m, dim=0)
            acc += torch.exp(0.1 * ms - lse) * probe[i].double()
        want = acc * (x[sidx] > 0).double()
        torch.testing.assert_close(xg.grad[sidx].double(), want, rtol=1e-4, atol=1e-6)
    assert torch.equal(ops.gen_aggregate(x, graph, aggr="softmax_sg", t=0.1), sm)
    # max at full size: the bit-mask backward (the default here: the arg-max table is 1.25 GB) equals the row walk bit
    # for bit, and sampled source rows equal  relu'(x_s) * sum over the (destination, channel) pairs this source wins
    del out, sm
    grads = {}
    old_thr = ops.MAX_MASK_MIN_TABLE_BYTES
    try:
        for name, thr in (("mask", 0), ("rows", 1 << 60)):
            ops.MAX_MASK_MIN_TABLE_BYTES = thr
            xg = x.clone().requires_grad_(True)
            omx = ops.gen_aggregate(xg, graph, aggr="max")
            (omx * probe).sum().backward()
            grads[name] = xg.grad
            del xg
    finally:
        ops.MAX_MASK_MIN_TABLE_BYTES = old_thr
    assert torch.equal(grads["mask"], grads["rows"])
    ep = graph.eperm.long() if graph.eperm is not None else None
    for sidx in rows[:8].tolist():
        dsts = graph.t_col[trp[sidx]:trp[sidx + 1]].long()
        eids = graph.t_eperm[trp[sidx]:trp[sidx + 1]].long()
        acc = torch.zeros(C, device=dev, dtype=torch.float64)
        for i, e in zip(dsts.tolist(), eids.tolist()):
            nb = graph.col[rp[i]:rp[i + 1]].long()
            ids = ep[rp[i]:rp[i + 1]] if ep is not None else torch.arange(rp[i], rp[i + 1], device=dev)
            m = torch.relu(x[nb]) + 1e-7
            best = m.max(0).values
            first = torch.where(m == best, ids.unsqueeze(1).expand_as(m), torch.full_like(m, 1 << 40, dtype=torch.long)).min(0).values
            acc += (first == e).double() * probe[i].double()              # ties: the first edge (lowest id) only
        want = acc * (x[sidx] > 0).double()
        torch.testing.assert_close(grads["mask"][sidx].double(), want, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("kind", ["uniform", "powerlaw"])
def test_products_shape_destination_range_against_the_oracle(kind):
    """BASELINE config 4 at FULL size against the ORACLE ("chunk by destination range").
    The aggregation rows are independent, so a contiguous range of destination rows (>= 1/32 of them; on the power-law
    graph the range that holds the largest hub -- ~300 k in-edges, cut into work items and merged) is replayed on the host
    by oracle/sparse_ref.gen_propagate from the device's own inputs: the range's output rows elementwise at 1e-4, and,
    with the probe restricted to the range, grad_x of ALL 2.4 M source rows elementwise (softmax_sg: the relu mask depends
    on x alone, so no pre-activation sits within rounding of a kink), for softmax_sg and max."""
    from deep_gcns_torch_amd import ops, synth
    from deep_gcns_torch_amd.graph import Graph
    from oracle import sparse_ref
    dev = _dev()
    s = synth.SHAPES["products"]
    n, C = s["n"], s["channels"]
    gen = synth.undirected_random_graph if kind == "uniform" else synth.powerlaw_graph
    ei = gen(n, s["n_undirected"], s["seed"], device=dev)
    assert ei.size(1) == 126_167_309
    graph = Graph.from_edge_index(ei, n)
    rp = graph.rowptr.long()
    deg = rp[1:] - rp[:-1]
    want_rows = n // 32 + 1
    if kind == "uniform":
        lo = n // 3
    else:
        hub = int(deg.argmax())
        assert int(deg[hub]) > 100_000                      # split into work items + merge kernel at full size
        lo = max(0, min(hub - want_rows // 2, n - want_rows))
    hi = lo + want_rows
    e0, e1 = int(rp[lo]), int(rp[hi])
    assert (e1 - e0) >= 0.8 * ei.size(1) / 32 and e1 - e0 < 8_000_000
    x = torch.randn(n, C, device=dev, generator=torch.Generator(device=dev).manual_seed(2))
    probe = torch.zeros(n, C, device=dev)
    probe[lo:hi] = torch.randn(hi - lo, C, device=dev, generator=torch.Generator(device=dev).manual_seed(4))
    # the range's edges in CSR order (stable sort: original order within a row, which is what first-max needs)
    src_r = graph.col[e0:e1].long().cpu()
    dst_r = torch.repeat_interleave(torch.arange(hi - lo), deg[lo:hi].cpu())
    ei_r = torch.stack([src_r, dst_r])
    xh = x.cpu()
    ph = probe[lo:hi].cpu()
    del ei
    for aggr, kw in (("softmax_sg", dict(t=0.1)), ("max", {})):
        xg = x.clone().requires_grad_(True)
        out = ops.gen_aggregate(xg, graph, aggr=aggr, **kw)
        (out * probe).sum().backward()
        # the oracle in the reference's precision, except for rows of more than 4096 in-edges (the power-law hub: 306 k):
        # torch's float32 scatter-add runs through such a row sequentially and is itself 2e-3 off (measured against its own
        # float64 evaluation; the device folds 64-edge pieces pairwise) -- those rows are replayed in float64.  Rows are
        # independent, so the two edge sets are two oracle calls whose outputs and gradients add.
        heavy = (deg[lo:hi].cpu() > 4096)[dst_r]
        ref = torch.zeros(hi - lo, C, dtype=torch.float64)
        gref = torch.zeros(n, C, dtype=torch.float64)
        for sel, dt in ((~heavy, torch.float32), (heavy, torch.float64)):
            if not bool(sel.any()):
                continue
            xr = xh.to(dt, copy=True).requires_grad_(True)
            part = sparse_ref.gen_propagate(xr, ei_r[:, sel], aggr=aggr, dim_size=hi - lo, **kw)
            (part * ph.to(dt)).sum().backward()
            ref += part.detach().double()
            gref += xr.grad.double()
            del xr, part
        ref = ref.float()
        torch.testing.assert_close(out[lo:hi].detach().cpu(), ref.detach(), rtol=RTOL, atol=1e-6)
        got, want = xg.grad.cpu(), gref.float()
        # max: a (row, channel) whose best messages coincide (duplicate edges of the symmetrised graph, the relu floor)
        # sends its gradient to the FIRST such edge on both sides (CSR order = original order within a row): nothing to
        # excuse, everything is compared
        gscale = float(want.abs().max())
        torch.testing.assert_close(got, want, rtol=RTOL, atol=1e-5 * max(gscale, 1.0))
        del xg, out, ref, gref, got, want


@pytest.mark.parametrize("t,expect_shifted", [(0.1, True), (1.0, True), (40.0, False)])
def test_single_gather_softmax_backward_and_its_device_side_fallback(t, expect_shifted):
    """The softmax backward gathers ONE pre-scaled row per edge when every |L_i| < 80 (checked by the forward
    kernel), otherwise (decided on the device, no host sync) two rows.  Both must match the float64 oracle."""
    from deep_gcns_torch_amd import ops
    from oracle import sparse_ref
    dev = _dev()
    ei = AGG["graphs"]["tricky"]
    g = torch.Generator().manual_seed(11)
    x = torch.randn(257, 64, generator=g) * 2.0
    probe = torch.randn(257, 64, generator=g)
    xr = x.double().requires_grad_(True)
    ref = sparse_ref.gen_propagate(xr, ei, aggr="softmax_sg", t=t)
    (ref * probe.double()).sum().backward()
    grads = {}
    for flag in (True, False):
        ops.SINGLE_GATHER_SOFTMAX_BWD = flag
        try:
            xd = x.to(dev).requires_grad_(True)
            out = ops.gen_aggregate(xd, ei.to(dev), aggr="softmax_sg", t=t)
            (out * probe.to(dev)).sum().backward()
        finally:
            ops.SINGLE_GATHER_SOFTMAX_BWD = True
        grads[flag] = xd.grad.cpu()
        gs = xr.grad.abs().max().item()
        torch.testing.assert_close(grads[flag].double(), xr.grad, rtol=RTOL, atol=2e-6 * max(gs, 1.0))
    # range check itself
    xd = x.to(dev)
    from deep_gcns_torch_amd.graph import Graph
    m = torch.relu(x) + 1e-7
    lse = torch.zeros(257, 64, dtype=torch.float64)
    s = (t * m.double())[ei[0]]
    mx = torch.zeros(257, 64, dtype=torch.float64).scatter_reduce_(0, ei[1].view(-1, 1).expand(-1, 64), s, "amax", include_self=False)
    den = torch.zeros(257, 64, dtype=torch.float64).index_add_(0, ei[1], torch.exp(s - mx[ei[1]]))
    has = torch.bincount(ei[1], minlength=257) > 0
    lse[has] = (mx + torch.log(den.clamp_min(1e-300)))[has]
    assert (float(lse.abs().max()) < 80.0) == expect_shifted
    if not expect_shifted:
        assert torch.equal(grads[True], grads[False])     # same two-gather code path, bit-identical


def test_degenerate_graphs():
    """Empty edge list, a single node, all edges into one node, C not a multiple of 4, one channel."""
    from deep_gcns_torch_amd import ops
    from oracle import sparse_ref
    dev = _dev()
    x = torch.randn(5, 8, device=dev, requires_grad=True)
    empty = torch.zeros(2, 0, dtype=torch.long, device=dev)
    for aggr in ("add", "mean", "max", "softmax", "power"):
        out = ops.gen_aggregate(x, empty, aggr=aggr, dim_size=5)
        ref = sparse_ref.gen_propagate(x.detach().cpu(), empty.cpu(), aggr=aggr, dim_size=5)
        torch.testing.assert_close(out.detach().cpu(), ref, rtol=1e-5, atol=1e-9)
        out.sum().backward()
        assert torch.count_nonzero(x.grad) == 0
        x.grad = None
    one = torch.randn(1, 3, device=dev)                       # single node with a self loop, C = 3
    loop = torch.zeros(2, 1, dtype=torch.long, device=dev)
    torch.testing.assert_close(ops.gen_aggregate(one, loop, aggr="softmax").cpu(),
                               sparse_ref.gen_propagate(one.cpu(), loop.cpu(), aggr="softmax"), rtol=1e-5, atol=1e-7)
    g = torch.Generator().manual_seed(0)
    xs = torch.randn(300, 1, generator=g)                     # one channel, star graph into node 0
    star = torch.stack([torch.arange(300), torch.zeros(300, dtype=torch.long)])
    for aggr, kw in (("softmax", dict(t=3.0)), ("max", {}), ("power", dict(p=3.0))):
        xr = xs.clone().requires_grad_(True)
        ref = sparse_ref.gen_propagate(xr, star, aggr=aggr, **kw)
        ref.sum().backward()
        xd = xs.to(dev).requires_grad_(True)
        out = ops.gen_aggregate(xd, star.to(dev), aggr=aggr, **kw)
        out.sum().backward()
        torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-7)
        torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=1e-4, atol=1e-6)


def test_strided_inputs_and_non_default_stream():
    """x given as a column slice (row stride > C) and launches on a side stream (library follows torch's
    current stream; the caller synchronises streams as usual)."""
    from deep_gcns_torch_amd import ops
    dev = _dev()
    ei = AGG["graphs"]["tricky"].to(dev)
    big = torch.randn(257, 192, device=dev)
    x = big[:, 64:128]                                         # stride (192, 1): rows are 16-B aligned
    ref = ops.gen_aggregate(x.contiguous(), ei, aggr="softmax_sg", t=0.5)
    assert torch.equal(ops.gen_aggregate(x, ei, aggr="softmax_sg", t=0.5), ref)
    xu = big[:, 1:65]                                          # misaligned rows -> scalar-lane kernel, same result
    torch.testing.assert_close(ops.gen_aggregate(xu, ei, aggr="softmax_sg", t=0.5),
                               ops.gen_aggregate(xu.contiguous(), ei, aggr="softmax_sg", t=0.5), rtol=1e-5, atol=1e-6)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        out = ops.gen_aggregate(x, ei, aggr="softmax_sg", t=0.5)
    s.synchronize()
    assert torch.equal(out, ref)


@pytest.mark.parametrize("aggr,kw", [("softmax_sg", dict(t=0.3)), ("power", dict(p=2.0)), ("max", {}), ("mean", {}), ("add", {})])
@pytest.mark.parametrize("C,with_ea", [(128, False), (16, True), (50, False)])
def test_add_root_equals_x_plus_aggregate(aggr, kw, C, with_ea):
    """out = x + AGGR(...) from the kernel epilogue (and grad_x = g + backward) against the two-op composition, on a
    power-law graph (split hub rows take the merge kernels), float4 / sub-group / scalar layouts."""
    from deep_gcns_torch_amd import ops, synth
    dev = _dev()
    n = 5000
    ei = synth.powerlaw_graph(n, 40_000, seed=11, exponent=2.1).to(dev)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(n, C, generator=g).to(dev)
    probe = torch.randn(n, C, generator=g).to(dev)
    ea = torch.randn(ei.size(1), C, generator=g).to(dev) if with_ea else None
    xa = x.clone().requires_grad_(True)
    ea_a = ea.clone().requires_grad_(True) if with_ea else None
    fused = ops.gen_aggregate(xa, ei, ea_a, aggr=aggr, add_root=True, **kw)
    (fused * probe).sum().backward()
    xb = x.clone().requires_grad_(True)
    ea_b = ea.clone().requires_grad_(True) if with_ea else None
    comp = xb + ops.gen_aggregate(xb, ei, ea_b, aggr=aggr, **kw)
    (comp * probe).sum().backward()
    torch.testing.assert_close(fused, comp, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(xa.grad, xb.grad, rtol=1e-5, atol=1e-5)
    if with_ea:
        torch.testing.assert_close(ea_a.grad, ea_b.grad, rtol=1e-6, atol=1e-6)
    with pytest.raises(ValueError):
        ops.gen_aggregate(xa, ei, aggr="softmax", t=torch.ones(1, device=dev, requires_grad=True), learn_t=True, add_root=True)


def test_narrow_edge_encoders_take_the_stock_path():
    """An edge encoder on RAW 8-wide edge features (a GENConv built directly on ogbn-proteins' edge_attr) matches no
    reference call site -- every reference model encodes at model level and hands each layer the (E, hidden) embedding,
    the shape served by the matrix-core kernel (tests/test_egemm_gpu.py) -- so the predicate refuses it and GENConv
    applies its Linear and aggregates the (E, C) rows: still correct against the oracle."""
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()
    from deep_gcns_torch_amd import ops, synth
    from gcn_lib.sparse.torch_vertex import GENConv
    from oracle import sparse_ref
    dev = _dev()
    n, C = 500, 32
    ei = synth.powerlaw_graph(n, 3_000, seed=13, exponent=2.1)
    g = torch.Generator().manual_seed(C)
    x = torch.randn(n, C, generator=g)
    feat = torch.randn(ei.size(1), 8, generator=g)
    conv = GENConv(C, C, aggr="softmax", t=1.0, encode_edge=True, edge_feat_dim=8, norm="layer", mlp_layers=1)
    assert not ops.encoder_fusable(x.to(dev), feat.to(dev), conv.edge_encoder.weight.to(dev))
    assert not ops.encoder_fusable(x.to(dev), torch.zeros(ei.size(1), 7).to(dev), conv.edge_encoder.weight.to(dev))
    emb = conv.edge_encoder(feat)
    ref = conv.mlp(x + sparse_ref.gen_propagate(x, ei, emb, aggr="softmax", t=1.0))
    out = conv.to(dev)(x.to(dev), ei.to(dev), feat.to(dev))
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("aggr,kw,C", [("softmax_sg", dict(t=0.3), 64), ("softmax", dict(t=1.0, learn_t=True), 112),
                                       ("power", dict(p=2.0), 32), ("max", {}, 16), ("mean", {}, 256), ("max", {}, 64)])
def test_per_edge_encoder_kernels_match_linear_then_aggregate(aggr, kw, C):
    """The per-edge encoder kernels (dgcn_gen_aggr_enc_*: relu(x_j + W f_e + b) + eps from 8 raw features per edge, no
    (E, C) array) that serve blocks.ComposedEdgeEmbedding -- W, b = the composition of the model-level and the per-layer
    edge encoder -- against the two-step form (Linear, then aggregation of the (E, C) rows): outputs, gradients w.r.t.
    x, the weight and the bias (per-workgroup partial sums), learnable t; hub rows, sub-group and padded-lane layouts."""
    from deep_gcns_torch_amd import ops, synth
    dev = _dev()
    n = 4000
    ei = synth.powerlaw_graph(n, 30_000, seed=13, exponent=2.1).to(dev)
    E = ei.size(1)
    g = torch.Generator().manual_seed(C)
    x = torch.randn(n, C, generator=g).to(dev)
    feat = torch.randn(E, 8, generator=g).to(dev)
    W = (torch.randn(C, 8, generator=g) * 0.5).to(dev)
    b = (torch.randn(C, generator=g) * 0.5).to(dev)
    probe = torch.randn(n, C, generator=g).to(dev)
    kw = dict(kw)
    if kw.get("learn_t"):
        kw["t"] = torch.tensor([kw["t"]], device=dev, requires_grad=True)
    assert ops.encoder_fusable(x, feat, W, narrow=True) and not ops.encoder_fusable(x, feat, W)

    def run(fused):
        xa, Wa, ba = x.clone().requires_grad_(True), W.clone().requires_grad_(True), b.clone().requires_grad_(True)
        ta = kw.get("t")
        if isinstance(ta, torch.Tensor):
            ta = ta.detach().clone().requires_grad_(True)
        k2 = dict(kw, t=ta) if ta is not None else dict(kw)
        if fused:
            out = ops.gen_aggregate(xa, ei, feat, aggr=aggr, edge_encoder=(Wa, ba), **k2)
        else:
            out = ops.gen_aggregate(xa, ei, torch.nn.functional.linear(feat, Wa, ba), aggr=aggr, **k2)
        (out * probe).sum().backward()
        return out.detach(), xa.grad, Wa.grad, ba.grad, (ta.grad if isinstance(ta, torch.Tensor) else None)

    of, gxf, gwf, gbf, gtf = run(True)
    oc, gxc, gwc, gbc, gtc = run(False)
    torch.testing.assert_close(of, oc, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(gxf, gxc, rtol=1e-4, atol=1e-5 * float(gxc.abs().max()))
    torch.testing.assert_close(gwf, gwc, rtol=1e-4, atol=2e-5 * float(gwc.abs().max()))
    torch.testing.assert_close(gbf, gbc, rtol=1e-4, atol=2e-5 * float(gbc.abs().max()))
    if gtc is not None:
        torch.testing.assert_close(gtf, gtc, rtol=1e-4, atol=1e-5 * float(gtc.abs().max()))
    # ... and against the ORACLE (oracle/sparse_ref.py on the host: the reference's edge_encoder -> message -> aggregate
    # chain, gcn_lib/sparse/torch_vertex.py:56-68), not only against this package's own unfused path
    from oracle import sparse_ref
    xr, Wr, br = (t.detach().cpu().clone().requires_grad_(True) for t in (x, W, b))
    kr = {k: (v.detach().cpu().clone().requires_grad_(True) if isinstance(v, torch.Tensor) else v) for k, v in kw.items()}
    ref = sparse_ref.gen_propagate(xr, ei.cpu(), torch.nn.functional.linear(feat.cpu(), Wr, br), aggr=aggr, dim_size=n, **kr)
    (ref * probe.cpu()).sum().backward()
    torch.testing.assert_close(of.cpu(), ref.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(gxf.cpu(), xr.grad, rtol=1e-4, atol=1e-5 * float(xr.grad.abs().max()))
    torch.testing.assert_close(gwf.cpu(), Wr.grad, rtol=1e-4, atol=1e-4 * float(Wr.grad.abs().max()))
    torch.testing.assert_close(gbf.cpu(), br.grad, rtol=1e-4, atol=1e-4 * float(br.grad.abs().max()))
    if gtc is not None:
        torch.testing.assert_close(gtf.cpu(), kr["t"].grad, rtol=1e-4, atol=1e-4 * float(kr["t"].grad.abs().max()))
    # shapes the fused path does not take are refused by the predicate (the module then builds the embedding)
    assert not ops.encoder_fusable(x, torch.zeros(E, 7, device=dev), W, narrow=True)
    # raw features that require grad: the predicate says no, the op refuses loudly (no silent None gradient)
    fr = feat.clone().requires_grad_(True)
    assert not ops.encoder_fusable(x, fr, W, narrow=True)
    with pytest.raises(ValueError, match="no gradient w.r.t. the raw edge features"):
        ops.gen_aggregate(x, ei, fr, aggr=aggr, edge_encoder=(W, b), **{k: v for k, v in kw.items() if k != "t" or not isinstance(v, torch.Tensor)})


@pytest.mark.parametrize("C", [16, 32, 64, 100, 128, 130, 256])
@pytest.mark.parametrize("sorted_input", [False, True])
def test_max_backward_bit_mask_path_equals_the_row_walk(C, sorted_input, monkeypatch):
    """dgcn_gen_aggr_max_bwd_f32 (per-edge arg-max bit masks, CSR-position indexed) against the one-launch walk that
    gathers arg-max rows -- hubs split on both sides, duplicate edges (ties -> first edge only), unsorted and
    destination-sorted edge lists (eperm = None), channel counts around the 32-bit word boundaries: bit-identical."""
    from deep_gcns_torch_amd import ops, synth
    dev = _dev()
    n = 6000
    ei = synth.powerlaw_graph(n, 60_000, seed=11, exponent=2.1)
    ei = torch.cat([ei, ei[:, :500]], dim=1)                           # duplicate edges: exact ties
    if sorted_input:
        ei = ei[:, torch.argsort(ei[1], stable=True)]
    ei = ei.to(dev)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, C, generator=g).to(dev)
    probe = torch.randn(n, C, generator=g).to(dev)
    grads = {}

    def run(name):
        for add_root in (False, True):
            xd = x.clone().requires_grad_(True)
            out = ops.gen_aggregate(xd, ei, aggr="max", add_root=add_root)
            (out * probe).sum().backward()
            grads[(name, add_root)] = xd.grad.clone()
        xd = x.clone().requires_grad_(True)
        out = ops.gen_aggregate(xd, ei, aggr="max", relu_eps=False)
        (out * probe).sum().backward()
        grads[(name, "raw")] = xd.grad.clone()

    monkeypatch.setattr(ops, "MAX_MASK_MIN_TABLE_BYTES", 1 << 60)
    run("rows")
    monkeypatch.setattr(ops, "MAX_MASK_MIN_TABLE_BYTES", 0)
    run("mask")
    for k in (False, True, "raw"):
        assert torch.equal(grads[("mask", k)], grads[("rows", k)]), k


@pytest.mark.parametrize("C", [112, 40, 256])
def test_per_edge_encoder_max_backward_from_the_winners(C):
    """Max aggregation with the per-edge encoder: the backward that takes grad_x from the plain walk over the forward's
    arg-max ids (-1 marks the relu floor) and dW' | db' from the (row, channel) winners (dgcn_enc_max_bwd_weight_f32)
    against the per-edge encoder walk that recomputes z for every edge -- same forward, channels at the relu floor on
    every edge, rows without edges, the fused root term."""
    from deep_gcns_torch_amd import ops, synth
    dev = _dev()
    n = 3000
    ei = synth.powerlaw_graph(n, 25_000, seed=3, exponent=2.1)
    ei = ei[:, ei[1] < n - 40].to(dev)                    # the last 40 destination rows have no edges
    E = ei.size(1)
    g = torch.Generator().manual_seed(C)
    x = torch.randn(n, C, generator=g).to(dev)
    feat = torch.randn(E, 8, generator=g).to(dev)
    W = (torch.randn(C, 8, generator=g) * 0.5).to(dev)
    b = torch.randn(C, generator=g) * 0.5
    b[:5] = -1e3                                           # z < 0 on every edge: m = eps, no gradient
    b = b.to(dev)
    probe = torch.randn(n, C, generator=g).to(dev)
    saved = ops.ENC_MAX_WINNER_BWD

    def run(flag):
        ops.ENC_MAX_WINNER_BWD = flag
        try:
            xa, Wa, ba = x.clone().requires_grad_(True), W.clone().requires_grad_(True), b.clone().requires_grad_(True)
            out = ops.gen_aggregate(xa, ei, feat, aggr="max", edge_encoder=(Wa, ba), dim_size=n, add_root=True)
            (out * probe).sum().backward()
            return out.detach(), xa.grad, Wa.grad, ba.grad
        finally:
            ops.ENC_MAX_WINNER_BWD = saved

    walk, win = run(False), run(True)
    assert torch.equal(win[0], walk[0])
    for a, r, what in zip(win[1:], walk[1:], ("grad_x", "grad_W", "grad_b")):
        torch.testing.assert_close(a, r, rtol=1e-5, atol=2e-6 * float(r.abs().max()), msg=lambda m, what=what: f"{what}: {m}")
    assert float(win[2][:5].abs().max()) == 0.0 and float(win[3][:5].abs().max()) == 0.0     # the dead channels
    # only the weights need a gradient: no walk at all
    ops.ENC_MAX_WINNER_BWD = True
    try:
        Wa, ba = W.clone().requires_grad_(True), b.clone().requires_grad_(True)
        out = ops.gen_aggregate(x, ei, feat, aggr="max", edge_encoder=(Wa, ba), dim_size=n, add_root=True)
        gW, gb = torch.autograd.grad((out * probe).sum(), [Wa, ba])
    finally:
        ops.ENC_MAX_WINNER_BWD = saved
    assert torch.equal(gW, win[2]) and torch.equal(gb, win[3])


def test_encoder_walk_item_schedules():
    """The per-edge encoder kernels hand their work items out from device-side counters (default) or deal them by wave
    index (ops.ENC_STATIC_ITEMS / DGCN_FLAG_STATIC_ITEMS).  Outputs and grad_x: the same bits under both schedules and
    from run to run (an item's result does not depend on who computes it).  dW | db: bit-reproducible under the static
    schedule; under the dynamic one equal to rounding (per-workgroup partial sums are grouped by the schedule)."""
    from deep_gcns_torch_amd import ops, synth
    dev = _dev()
    n, C = 6000, 112
    ei = synth.powerlaw_graph(n, 120_000, seed=21, exponent=2.2).to(dev)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, C, generator=g).to(dev)
    feat = torch.rand(ei.size(1), 8, generator=g).to(dev)
    W = (torch.randn(C, 8, generator=g) / 3).to(dev)
    b = torch.randn(C, generator=g).to(dev)
    probe = torch.randn(n, C, generator=g).to(dev)

    def run(static):
        saved = ops.ENC_STATIC_ITEMS
        ops.ENC_STATIC_ITEMS = static
        try:
            xa, Wa, ba = (t.clone().requires_grad_(True) for t in (x, W, b))
            out = ops.gen_aggregate(xa, ei, feat, aggr="softmax", t=1.0, edge_encoder=(Wa, ba))
            (out * probe).sum().backward()
            return out.detach(), xa.grad, Wa.grad, ba.grad
        finally:
            ops.ENC_STATIC_ITEMS = saved
    s1, s2, d1, d2 = run(True), run(True), run(False), run(False)
    for a, b_ in zip(s1, s2):
        assert torch.equal(a, b_)                                  # static: every result bit for bit
    for k in (0, 1):
        assert torch.equal(d1[k], d2[k]) and torch.equal(d1[k], s1[k])      # out, grad_x: schedule-independent bits
    for k in (2, 3):
        torch.testing.assert_close(d1[k], s1[k], rtol=1e-5, atol=1e-5 * float(s1[k].abs().max()))
        torch.testing.assert_close(d1[k], d2[k], rtol=1e-5, atol=1e-5 * float(s1[k].abs().max()))
