"""The instrument itself (tests/attribution.py) on the CPU: planted ReLU kinks and arg-max ties are found, explain the
gradient difference between two evaluations that resolve them differently, and nothing else is excused."""
import pytest
import torch

import attribution
from deep_gcns_torch_amd import synth
from oracle import dense_ref, sparse_ref


def _sparse_eval(x, ei, feat, W, b, n, aggr, probe, **kw):
    xs, fs, Ws, bs = (t.clone().requires_grad_(True) for t in (x, feat, W, b))
    out = sparse_ref.gen_propagate(xs, ei, torch.nn.functional.linear(fs, Ws, bs), aggr=aggr, dim_size=n, **kw)
    (out * probe).sum().backward()
    return out.detach(), dict(grad_x=xs.grad, grad_feat=fs.grad, grad_W=Ws.grad, grad_b=bs.grad)


@pytest.mark.parametrize("aggr,kw", [("softmax", dict(t=0.7)), ("power", dict(p=1.0)), ("max", {}), ("mean", {})])
def test_planted_relu_kink_is_marked_and_explains_the_difference(aggr, kw):
    n, C, K = 258, 16, 32
    # node 257 has exactly one outgoing edge (moving x[257, c] moves one pre-activation only) into a low-degree row
    ei = torch.cat([synth.tricky_graph(), torch.tensor([[257], [3]])], dim=1)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n, C, generator=g)
    feat = torch.randn(ei.size(1), K, generator=g)
    W = torch.randn(C, K, generator=g) / K ** 0.5
    b = torch.randn(C, generator=g)
    probe = torch.randn(n, C, generator=g)
    s, e = 257, ei.size(1) - 1
    c = 3
    z64 = x[s, c].double() + feat[e].double() @ W[c].double() + b[c].double()
    xa, xb = x.clone(), x.clone()
    xa[s, c] = (x[s, c].double() - z64 + 3e-7).float()          # z = +3e-7: passes its gradient
    xb[s, c] = (x[s, c].double() - z64 - 3e-7).float()          # z = -3e-7: does not
    out_a, ga = _sparse_eval(xa, ei, feat, W, b, n, aggr, probe, **kw)
    out_b, gb = _sparse_eval(xb, ei, feat, W, b, n, aggr, probe, **kw)
    bounds = attribution.sparse_flip_bounds(xa, ei, feat, W, b, n, aggr, probe, **kw)
    assert 1 <= bounds["n_kink"] <= 5                         # the planted pair (+ the odd natural one among 66 k)
    torch.testing.assert_close(out_a, out_b, rtol=1e-5, atol=1e-5)             # the forward is continuous
    moved = float((ga["grad_x"] - gb["grad_x"]).abs().max())
    if aggr == "max" and moved == 0.0:
        pytest.skip("the planted edge is not the arg-max of its row: no gradient to move")
    assert moved > 1e-3                                                       # the gradient is not
    for what in ("grad_x", "grad_feat", "grad_W", "grad_b"):
        used = attribution.assert_explained(ga[what], gb[what], bounds[what], 1e-5, 1e-6, what)
        assert used >= 1
        with pytest.raises(AssertionError):                                   # without the attribution: a failure
            attribution.assert_explained(ga[what], gb[what], None, 1e-5, 1e-6, what)
    # an error of the same size at an element no marked pair feeds is NOT excused
    wrong = ga["grad_x"].clone()
    other = (s + 1) % n
    wrong[other, c] += moved
    with pytest.raises(AssertionError):
        attribution.assert_explained(wrong, gb["grad_x"], bounds["grad_x"], 1e-5, 1e-6, "grad_x")


def test_max_aggregation_near_tie_is_marked():
    n, C, K = 64, 8, 16
    ei = synth.tricky_graph(n=64, e=700, hub_deg=300, seed=7)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, C, generator=g)
    feat = torch.randn(ei.size(1), K, generator=g)
    W = torch.randn(C, K, generator=g) / K ** 0.5
    b = torch.randn(C, generator=g)
    probe = torch.randn(n, C, generator=g)
    # duplicate the edge that wins row i, channel c: an exact tie (two candidates, both passing their gradient)
    emb = torch.nn.functional.linear(feat, W, b)
    m = torch.relu(x[ei[0]] + emb)
    i = int(ei[1, 0])
    rows = (ei[1] == i).nonzero().squeeze(1)
    c = 2
    win = int(rows[m[rows, c].argmax()])
    assert m[win, c] > 0
    ei2 = torch.cat([ei, ei[:, win:win + 1]], dim=1)
    feat2 = torch.cat([feat, feat[win:win + 1]], dim=0)
    bounds = attribution.sparse_flip_bounds(x, ei2, feat2, W, b, n, "max", probe)
    assert bounds["n_tied"] >= 2
    assert float(bounds["grad_x"][int(ei[0, win]), c]) >= abs(float(probe[i, c])) * 0.999
    assert attribution.sparse_flip_bounds(x, ei, feat, W, b, n, "max", probe)["n_tied"] == 0


@pytest.mark.parametrize("conv", ["edge", "mr"])
def test_dense_ties_are_masked_and_the_rest_compares_strictly(conv):
    B, C, N, k = 2, 8, 96, 6
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, C, N, 1, generator=g)
    x[:, :, 11] = x[:, :, 10]                                 # two identical points: every neighbourhood that holds both ties
    ei = dense_ref.dense_knn_matrix(x, k)
    torch.manual_seed(0)
    nn = torch.nn.Sequential(torch.nn.Conv2d(2 * C, C, 1), torch.nn.ReLU(), torch.nn.BatchNorm2d(C)).train()
    with torch.no_grad():
        nn[2].weight.copy_(torch.randn(C, generator=g))        # both signs: max and min branch
    probe = torch.randn(B, C, N, 1, generator=g)
    attr = attribution.dense_edgeconv_attribution if conv == "edge" else attribution.dense_mrconv_attribution
    fn = dense_ref.edgeconv2d if conv == "edge" else dense_ref.mrconv2d
    probe_m, extra, info = attr(x, ei, nn[0].weight, nn[0].bias, nn[2].weight, nn[2].bias, probe, nn[2].eps)
    assert info["n_masked_outputs"] > 0 and bool((probe_m == 0).sum() == info["n_masked_outputs"])

    def grads(xin, pr):
        xs = xin.clone().requires_grad_(True)
        return torch.autograd.grad((fn(xs, ei, nn) * pr).sum(), [xs, nn[0].weight, nn[0].bias])

    # second evaluation: the tie resolved the other way (point 10 nudged below / above its twin by less than the
    # rounding the marking allows for)
    x2 = x.clone()
    x2[:, :, 10] = x[:, :, 10] * (1 + 2e-7)
    a = grads(x, probe_m)
    b = grads(x2, probe_m)
    scale = float(a[0].abs().max())
    for what, u, v, ex in zip(("grad_x", "grad_W", "grad_b"), a, b,
                              (extra["grad_x"], extra["grad_W"].view_as(nn[0].weight), extra["grad_b"])):
        attribution.assert_explained(u, v, ex, 1e-4, 1e-5 * max(scale, float(u.abs().max())), what)
    # with the unmasked probe the two evaluations differ by whole terms
    a, b = grads(x, probe), grads(x2, probe)
    assert float((a[0] - b[0]).abs().max()) > 1e-3 * scale
