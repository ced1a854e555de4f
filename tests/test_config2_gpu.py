"""BASELINE config 2 AS ONE MODEL on the device: sem_seg_dense ResGCN-28 (examples/sem_seg_dense/architecture.py:7-56), B = 8
clouds x N = 4096 points, k = 16, dilations 1..27, train mode -- forward + backward of the cross-entropy loss with the HIP kNN
(gcn_lib/dense/torch_edge.py:32-76 -> dgcn_knn_dense_f32) IN THE LOOP: every block builds its graph from the features the
device computed, no graph is borrowed from the oracle (VERDICT r5 missing #1).

A dynamic-graph network of this depth is chaotic in its neighbour ids -- the reference's OWN float32 run shares 6 % of its
edges with its float64 run from block 6 on (tests/golden/make_resgcn28_golden.py, which executed the reference's real
architecture.py on its real gcn_lib.dense in both precisions) -- so there are two tests, each of which states something
that is true of a correct float32 implementation:

1. `test_step_equals_the_float64_replay_along_its_own_graphs`: the whole training step (logits, loss, every parameter
   gradient, the input gradient) equals a float64 evaluation of the oracle's formulation (oracle/dense_ref.py, torch
   float64 ops on the same device: checker, not product) that is handed the 28 graphs the device run built.  Everything
   but the discrete graph choice is compared at full depth; the graph choice is compared per block: the ids the HIP kNN
   emitted against the float64 ranking of the block's own input features -- counted, recorded, gated.
2. `test_divergence_from_the_reference_float64_run_is_the_reference_float32_runs`: against the fixture.  Per block, the
   relative distance of the device's features from the reference's float64 trajectory next to the same number for the
   reference's float32 run, and the count of kNN ids that differ from the float64 ranking on both sides.
"""
import os

import pytest
import torch

import arch_restated
import config_replays as cr
from conftest import gate

pytestmark = pytest.mark.gpu

_RUN = {}


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _knn_modules(model):
    return [model.knn] + [blk.body.dilated_knn_graph for blk in model.backbone]


def _rank64_ids(feats, k, d):
    """float64 ranking of (B,C,N,1) features on their own device: (B,N,k) ids of rank 0, d, 2d, ..."""
    p = feats.detach().squeeze(-1).transpose(1, 2).double()
    out = []
    for b in range(p.size(0)):
        q = p[b]
        sq = (q * q).sum(-1)
        d64 = sq.unsqueeze(1) - 2 * q @ q.t() + sq.unsqueeze(0)
        out.append(torch.topk(d64, k * d, dim=1, largest=False, sorted=True).indices[:, ::d])
    return torch.stack(out)


def _device_run():
    """One training step of the device model; cached for the two tests of this file."""
    if _RUN:
        return _RUN
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()
    dev = _dev()
    inp = cr.resgcn_inputs(8, 4096)
    m = arch_restated.DenseDeepGCN(n_blocks=28, channels=64, k=16)
    cr.dense_formula_init(m, seed=2)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m.to(dev).train()
    x = inp["inputs"].to(dev).requires_grad_(True)
    graphs, knn_in, feats = [], [], []
    def knn_hook(mod, a, out):                      # (a hook that returns something replaces the output: return None)
        graphs.append(out)
        knn_in.append((a[0].detach(), mod.k, mod.dilation))

    def feat_hook(mod, a, out):
        feats.append(out.detach())

    handles = [km.register_forward_hook(knn_hook) for km in _knn_modules(m)]
    handles += [blk.register_forward_hook(feat_hook) for blk in [m.head] + list(m.backbone)]
    # the step's DECISIONS, for the float64 evaluation along them (third test of this file): per edge convolution the
    # P | Q rows, neighbour ids and arg-max / arg-min slots its kernels keep for their backward (the per-edge ReLU's input is
    # the single fp32 addition P_i + Q_j: its sign is reproduced exactly from these); the masks of the head's three ReLU
    # modules; the arg-max of the global max-pool
    import attribution
    from deep_gcns_torch_amd import dense_ops
    conv_dec, pool = [], {}
    orig_fwd = dense_ops._EdgeConv2dFused.forward

    def fwd(ctx, *a, **k):
        out = orig_fwd(ctx, *a, **k)
        x3, W2, pq, idx, amax, amin, vmax, vmin, bnbuf, gamma = ctx.to_save
        conv_dec.append(dict(pq=pq, idx=idx, amax=amax, amin=amin, gamma=gamma.detach().clone()))
        return out
    dense_ops._EdgeConv2dFused.forward = staticmethod(fwd)
    handles.append(m.fusion_block.register_forward_hook(
        lambda mod, a, out: pool.__setitem__("argmax", out.detach().flatten(2).argmax(2))))
    head_relus = attribution.ReluDecisions()
    try:
        with head_relus.recording():
            logits = m(x)
            loss = torch.nn.functional.cross_entropy(logits, inp["target"].to(dev))
            loss.backward()
    finally:
        dense_ops._EdgeConv2dFused.forward = orig_fwd
    for h in handles:
        h.remove()
    assert len(conv_dec) == 28 and len(head_relus.masks) == 3, (len(conv_dec), len(head_relus.masks))
    assert len(graphs) == 28 and len(feats) == 28
    vs_rank64 = []
    for ei, (xin, k, d) in zip(graphs, knn_in):
        assert ei.shape == (2, 8, 4096, 16) and ei.dtype == torch.int64
        vs_rank64.append(int((_rank64_ids(xin, k, d) != ei[0]).sum()))
    _RUN.update(model=m, sd=sd, inp=inp, x=x, logits=logits.detach(), loss=float(loss.detach()), graphs=graphs, feats=feats,
                grads={k: p.grad.detach().clone() for k, p in m.named_parameters()}, grad_x=x.grad.detach().clone(),
                knn_vs_rank64=vs_rank64, conv_dec=conv_dec, head_relus=head_relus, pool_argmax=pool["argmax"])
    return _RUN


def test_step_equals_the_float64_replay_along_its_own_graphs():
    run = _device_run()                                  # (install() inside: gcn_lib resolves to this package afterwards)
    from gcn_lib.dense import torch_edge, torch_vertex
    from oracle import dense_ref
    dev = _dev()
    m64 = arch_restated.DenseDeepGCN(n_blocks=28, channels=64, k=16)
    m64.load_state_dict(run["sd"])
    m64.double().to(dev).train()
    feed = iter(run["graphs"])
    saved_knn, saved_edge = torch_edge.DenseDilatedKnnGraph.forward, torch_vertex.EdgeConv2d.forward
    torch_edge.DenseDilatedKnnGraph.forward = lambda self, x: next(feed)
    torch_vertex.EdgeConv2d.forward = lambda self, x, edge_index, res_scale=None: torch_vertex._with_skip(
        dense_ref.edgeconv2d(x, edge_index, self.nn), x, res_scale)
    try:
        x64 = run["inp"]["inputs"].double().to(dev).requires_grad_(True)
        ref = m64(x64)
        loss64 = torch.nn.functional.cross_entropy(ref, run["inp"]["target"].to(dev))
        loss64.backward()
    finally:
        torch_edge.DenseDilatedKnnGraph.forward, torch_vertex.EdgeConv2d.forward = saved_knn, saved_edge

    def rel_max(a, b):
        return float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-300))

    # measure everything first (a failed gate must not hide the other numbers), then gate
    e_logits = rel_max(run["logits"], ref.detach())
    e_loss = abs(run["loss"] - float(loss64.detach()))
    e_gx = rel_max(run["grad_x"], x64.grad)
    per_param = {name: rel_max(run["grads"][name], p.grad) for name, p in m64.named_parameters()}
    worst_name = max(per_param, key=per_param.get)
    worst = per_param[worst_name]
    counts = run["knn_vs_rank64"]
    total = 8 * 4096 * 16
    print(f"[config 2] float64 replay along the device's graphs: logits {e_logits:.2e} of max |logit|, loss {e_loss:.2e}, "
          f"input gradient {e_gx:.2e} of its max, worst of {len(per_param)} parameter gradients {worst:.2e} ({worst_name}), "
          f"median {sorted(per_param.values())[len(per_param) // 2]:.2e}")
    print(f"[config 2] kNN ids that differ from the float64 ranking of the block's own features, per block, of {total}: {counts}")
    # The yardstick ("gated by the reference's own fp32-vs-fp64 error"): the fixture's third run -- the
    # reference's REAL architecture.py in float32 with its float64 run's graphs forced on every block -- against its float64
    # run: logits 7.4e-4, loss 2.7e-7, input gradient 3.1e-3, parameter gradients worst 1.3e-2 / median 4.4e-3 (max error /
    # max).  That is what rounding alone does to this step at full size: 33.5 M edge activations per block sit under a max
    # whose near-ties resolve either way and a ReLU whose kinks flip, and every flipped choice re-routes a gradient term
    # (tests/attribution.py shows that per layer; at 4 blocks x 1,024 points, where nothing flips, the same comparison
    # gives 2e-5).  The device, against the float64 evaluation of ITS graphs, may be off by at most twice that.
    path = cr.resgcn_fixture_path(28, 8, 4096)
    assert os.path.exists(path), "tests/golden/config_resgcn28_b8.pt missing (tests/golden/make_resgcn28_golden.py)"
    yard = torch.load(path, map_location="cpu", weights_only=False)["forced32_vs_64"]
    yg = sorted(yard["grads"].values())
    print(f"[config 2] the reference's float32 run along ITS float64 graphs vs float64: logits {yard['logits']:.2e}, loss "
          f"{yard['loss']:.2e}, input gradient {yard['grad_x']:.2e}, parameter gradients worst {yg[-1]:.2e} / median "
          f"{yg[len(yg) // 2]:.2e}")
    F = 2.0
    gate("config 2 step along its own graphs: logits, max error / max |logit| (float64 replay) / the reference float32's own",
         e_logits / yard["logits"], F)
    gate("config 2 step along its own graphs: |loss - float64 loss| / the reference float32's own", e_loss / yard["loss"], F)
    gate("config 2 step along its own graphs: input gradient, max error / max, / the reference float32's own",
         e_gx / yard["grad_x"], F)
    gate("config 2 step along its own graphs: worst parameter gradient (max error / max) / the reference float32's worst",
         worst / yg[-1], F, worst_name)
    gate("config 2 step along its own graphs: median parameter gradient error / the reference float32's median",
         sorted(per_param.values())[len(per_param) // 2] / yg[len(yg) // 2], F)
    # the discrete part: ids vs the float64 ranking of each block's own input
    gate("config 2 kNN in the loop: worst per-block fraction of ids that differ from the float64 ranking of the same features",
         max(counts) / total, 6e-3)


def test_step_equals_float64_along_its_own_graphs_and_decisions():
    """The strict form of the test above: the float64 evaluation is forced through the device step's graphs AND through every
    discontinuity behind them -- the per-edge ReLU of all 28 edge convolutions (33.5 M decisions per block: the sign of the
    single fp32 addition P_i + Q_j the kernel takes, reproduced bit for bit from the rows it keeps), the neighbour that wins
    each (point, channel) maximum (arg-max, or arg-min where the BatchNorm scale is negative), the head's three ReLU layers
    and the arg-max of the global max-pool (gcn_lib/dense/torch_vertex.py:31-35, torch_nn.py:48-60; examples/sem_seg_dense/
    architecture.py:40-55).  What is left between the device's float32 step and this evaluation is rounding, and it has to
    be small for the logits, the input gradient and EVERY parameter gradient -- no yardstick from the reference's own noise."""
    run = _device_run()
    import attribution
    from gcn_lib.dense import torch_edge, torch_vertex
    from oracle import dense_ref
    dev = _dev()
    def replay(dtype):
        mm = arch_restated.DenseDeepGCN(n_blocks=28, channels=64, k=16)
        mm.load_state_dict(run["sd"])
        mm.to(dtype).to(dev).train()
        feed = iter(run["graphs"])
        decs = iter(run["conv_dec"])

        def forced_edgeconv(self, x, edge_index, res_scale=None):
            d = next(decs)
            conv, bn = self.nn[0], self.nn[2]
            assert isinstance(bn, torch.nn.BatchNorm2d) and isinstance(self.nn[1], torch.nn.ReLU)
            nbr = edge_index[0]
            B, N, k = nbr.shape
            Cout = conv.out_channels
            # the device's per-edge decisions: sign of P_i + Q_j, one fp32 addition
            P, Q = d["pq"][..., :Cout], d["pq"][..., Cout:]
            Qj = Q[torch.arange(B, device=dev).view(B, 1, 1), nbr]                     # (B, N, k, Cout)
            on = ((P.unsqueeze(2) + Qj) > 0).permute(0, 3, 1, 2)                       # (B, Cout, N, k)
            x_i = dense_ref.batched_index_select(x, edge_index[1])
            x_j = dense_ref.batched_index_select(x, nbr)
            u = conv(torch.cat([x_i, x_j - x_i], dim=1))
            r = on.to(u.dtype) * attribution.passed_value(u)
            y = bn(r)
            slot = torch.where((d["gamma"] >= 0).view(1, 1, Cout), d["amax"].long(), d["amin"].long())     # (B, N, Cout)
            out = torch.gather(y, 3, slot.permute(0, 2, 1).unsqueeze(-1))
            return torch_vertex._with_skip(out, x, res_scale)

        pool_idx = run["pool_argmax"]

        def forced_pool(t, kernel_size, *a, **k):
            B, C = t.shape[:2]
            assert tuple(kernel_size) == tuple(t.shape[2:]) and pool_idx.shape == (B, C)
            return torch.gather(t.flatten(2), 2, pool_idx.unsqueeze(-1)).view(B, C, 1, 1)

        saved = (torch_edge.DenseDilatedKnnGraph.forward, torch_vertex.EdgeConv2d.forward, torch.max_pool2d)
        torch_edge.DenseDilatedKnnGraph.forward = lambda self, x: next(feed)
        torch_vertex.EdgeConv2d.forward = forced_edgeconv
        torch.max_pool2d = forced_pool
        try:
            with run["head_relus"].replaying():
                xin = run["inp"]["inputs"].to(dtype).to(dev).requires_grad_(True)
                out = mm(xin)
                loss = torch.nn.functional.cross_entropy(out, run["inp"]["target"].to(dev))
                loss.backward()
        finally:
            torch_edge.DenseDilatedKnnGraph.forward, torch_vertex.EdgeConv2d.forward, torch.max_pool2d = saved
        return dict(logits=out.detach(), loss=float(loss.detach()), grad_x=xin.grad,
                    grads={name: p.grad for name, p in mm.named_parameters()})

    r64 = replay(torch.float64)
    r32 = replay(torch.float32)          # the yardstick: torch's float32 kernels through the SAME branches

    def rel_max(a, b):
        return float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-300))

    def errors(got):
        pp = {name: rel_max(got["grads"][name], g) for name, g in r64["grads"].items()}
        return dict(logits=rel_max(got["logits"], r64["logits"]), loss=abs(got["loss"] - r64["loss"]),
                    gx=rel_max(got["grad_x"], r64["grad_x"]), worst=max(pp.values()), worst_name=max(pp, key=pp.get),
                    median=sorted(pp.values())[len(pp) // 2])

    e_dev = errors(dict(logits=run["logits"], loss=run["loss"], grad_x=run["grad_x"], grads=run["grads"]))
    e_32 = errors(r32)
    n_dec = 28 * 8 * 64 * 4096 * 16 + 28 * 8 * 64 * 4096 + sum(int(t.numel()) for t in run["head_relus"].masks) + 8 * 1024
    for who, e in (("the device step", e_dev), ("torch float32 through the same branches", e_32)):
        print(f"[config 2] float64 along the device's graphs AND its {n_dec} decisions vs {who}: logits {e['logits']:.2e} of max "
              f"|logit|, loss {e['loss']:.2e}, input gradient {e['gx']:.2e} of its max, parameter gradients worst {e['worst']:.2e} "
              f"({e['worst_name']}) / median {e['median']:.2e}")
    # With every branch fixed what remains is arithmetic: 28 blocks of training-mode BatchNorm over 33.5 M edge activations
    # each.  torch's own float32 kernels through the same branches are the measure of that; the device has to be as good.
    # measured (round 6): device 2.7e-4 / 1.1e-3 / 1.8e-3 / 6.4e-4, torch float32 1.9e-4 / 6.2e-4 / 3.7e-3 / 5.9e-4
    # (logits / input gradient / worst / median parameter gradient); without the decisions forced the device's gradients
    # are at 4.6e-3 / 1.9e-2 / 4.4e-3 (first test of this file)
    F = 2.5
    gate("config 2 along graphs and decisions: logits error / torch float32's through the same branches", e_dev["logits"] / e_32["logits"], F)
    gate("config 2 along graphs and decisions: input-gradient error / torch float32's", e_dev["gx"] / e_32["gx"], F)
    gate("config 2 along graphs and decisions: worst parameter-gradient error / torch float32's worst", e_dev["worst"] / e_32["worst"], F,
         e_dev["worst_name"])
    gate("config 2 along graphs and decisions: median parameter-gradient error / torch float32's median",
         e_dev["median"] / e_32["median"], F)


def test_divergence_from_the_reference_float64_run_is_the_reference_float32_runs():
    path = cr.resgcn_fixture_path(28, 8, 4096)
    assert os.path.exists(path), "tests/golden/config_resgcn28_b8.pt missing (tests/golden/make_resgcn28_golden.py)"
    fix = torch.load(path, map_location="cpu", weights_only=False)
    run = _device_run()
    names = set(fix["param_keys"])
    sd = {k: v.float() for k, v in run["sd"].items() if k in names}                    # parameters, as the generator sums them
    mine = cr.checksums(run["inp"]["inputs"], run["inp"]["target"].view(1, -1).repeat(2, 1), sd)
    for key, want in fix["checksums"].items():
        assert abs(mine[key] - want) <= 1e-9 * max(1.0, abs(want)), f"seeded {key} differs from the generator's"
    assert list(dict(run["model"].named_parameters()).keys()) == fix["param_keys"]

    pos = fix["positions"][:, ::fix["feat_stride"]]
    pick = lambda t: t.squeeze(-1).permute(0, 2, 1)[pos[0].to(t.device), pos[1].to(t.device)].cpu()
    f64 = fix["feats64"].double()
    dev_curve = [float((pick(f).double() - f64[l]).norm() / f64[l].norm()) for l, f in enumerate(run["feats"])]
    ref_curve = [float((fix["feats32"][l].double() - f64[l]).norm() / f64[l].norm()) for l in range(28)]
    print("[config 2] block features vs the reference's float64 run, relative L2 on the sampled positions")
    print("   device          :", [f"{v:.1e}" for v in dev_curve])
    print("   reference fp32  :", [f"{v:.1e}" for v in ref_curve])
    print("   (all positions) :", [f"{v:.1e}" for v in fix["feats_rel_l2_32_vs_64"]])
    total = fix["ids_per_block"]
    print(f"[config 2] ids that differ from the float64 ranking of the same features, of {total} per block")
    print("   device          :", run["knn_vs_rank64"])
    print("   reference fp32  :", fix["knn32_vs_rank64"])
    # block 0 ranks the xyz coordinates (the same numbers on both sides): the head's features leave the float64 run only
    # where a neighbour is ranked differently; from there both float32 runs leave the float64 trajectory exponentially.
    # Gate what is comparable: the head block, and the depth at which the run has left the trajectory (first block whose
    # features are further than 10 % from the float64 run's) -- the device may not leave it earlier than two blocks
    # before the reference's own float32 run does.
    gate("config 2 vs the reference float64 run: head block features, relative L2 (reference float32: "
         f"{ref_curve[0]:.1e})", dev_curve[0], max(10 * ref_curve[0], 1e-3))
    left = lambda curve: next((l for l, v in enumerate(curve) if v > 0.1), 28)
    print(f"[config 2] first block further than 10 % from the float64 run: device {left(dev_curve)}, reference float32 {left(ref_curve)}")
    assert left(dev_curve) >= left(ref_curve) - 2
    # ids: the device's rounding of a distance (six-product bf16 sum) against the reference's (float32 BLAS + adds)
    worst_dev, worst_ref = max(run["knn_vs_rank64"]), max(fix["knn32_vs_rank64"])
    gate("config 2 kNN: device's worst per-block count of ids off the float64 ranking / the reference float32's worst",
         worst_dev / max(worst_ref, 1), 4.0)
    # the float64 run's graph of cloud 0 in the first blocks (before either run has left the trajectory)
    g64 = fix["graphs64_cloud0"].long()
    mine0 = [int((run["graphs"][l][0, 0].cpu() != g64[l]).sum()) for l in range(g64.size(0))]
    print(f"[config 2] cloud 0: device ids that differ from the reference float64 RUN's graph, blocks 0..7, of {4096 * 16}: {mine0}")
    gate("config 2 kNN: block 0 (xyz) ids of cloud 0 that differ from the reference float64 run's graph, fraction",
         mine0[0] / (4096 * 16), 1e-3)
