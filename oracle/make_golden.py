"""ORACLE / TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.pt by executing the REAL
reference code (/root/reference, imported through oracle/refshim.py) on seeded inputs.

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden

Runs only in the build container (the reference tree does not exist on the GPU box); the
fixtures it writes are committed and are what every parity test compares against.
Each fixture stores inputs, parameters (state_dict) and the reference's outputs and
input/parameter gradients for the scalar loss  L = sum(out * probe).
"""
from __future__ import annotations

import os
import zlib
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from deep_gcns_torch_amd import synth  # noqa: E402
from oracle import refshim  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def _probe(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


def sparse_aggregate_cases(sparse):
    """GENConv.propagate (message + aggregate) of the reference for every aggregator."""
    cases = []
    ei = synth.tricky_graph()
    N = 257

    small = synth.tricky_graph(n=64, e=700, hub_deg=300, seed=7)
    graphs = {"tricky": ei, "small": small}

    def run(name, C, aggr, edge_attr=None, gname="tricky", **kw):
        graph = graphs[gname]
        n = 257 if gname == "tricky" else 64
        g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % (2 ** 31))
        x = torch.randn(n, C, generator=g).requires_grad_(True)
        conv = sparse.GENConv(C, C, aggr=aggr, norm="batch", **kw)
        ea = None
        if edge_attr:
            ea = torch.randn(graph.size(1), C, generator=g).requires_grad_(True)
        out = conv.propagate(graph, x=x, edge_attr=ea)
        if aggr.endswith("_sum"):
            pass  # degree scaling already applied inside aggregate (torch_message.py:60-63)
        probe = _probe(out.shape, 1234)
        params = {k: v for k, v in (("t", getattr(conv, "t", None)), ("p", getattr(conv, "p", None)),
                                    ("y", getattr(conv, "y", None))) if isinstance(v, torch.nn.Parameter)}
        wrt = [x] + ([ea] if ea is not None else []) + [v for v in params.values() if v.requires_grad]
        grads = torch.autograd.grad((out * probe).sum(), wrt, allow_unused=True)
        rec = dict(name=name, C=C, aggr=aggr, kw=kw, n=n, graph=gname, x=x.detach(),
                   edge_attr=None if ea is None else ea.detach(), probe=probe, out=out.detach(),
                   grad_x=grads[0])
        gi = 1
        if ea is not None:
            rec["grad_edge_attr"] = grads[gi]; gi += 1
        for k, v in params.items():
            if v.requires_grad:
                rec["grad_" + k] = grads[gi]; gi += 1
        cases.append(rec)

    for aggr in ["add", "mean", "max", "softmax", "softmax_sg", "power"]:
        run(f"tricky64_{aggr}", 64, aggr)
    run("tricky64_softmax_t0.1", 64, "softmax_sg", t=0.1)
    run("tricky64_softmax_learn_t", 64, "softmax", t=0.7, learn_t=True)
    run("tricky64_softmax_sum_learn", 64, "softmax_sum", t=0.5, learn_t=True, y=0.3, learn_y=True)
    run("tricky64_power_learn_p", 64, "power", p=2.0, learn_p=True)
    run("tricky64_power_p3", 64, "power", p=3.0)
    run("tricky64_power_sum", 64, "power_sum", p=1.5, learn_p=True, y=-0.2, learn_y=True)
    run("small128_softmax_sg", 128, "softmax_sg", gname="small", t=0.1)
    run("small20_softmax", 20, "softmax", gname="small")          # 5 of 8 lanes active
    run("small112_power", 112, "power", gname="small", p=2.0, learn_p=True)   # RevGCN-wide group width
    run("small50_max", 50, "max", gname="small")                  # C % 4 != 0 -> scalar path
    run("small50_softmax", 50, "softmax_sg", gname="small", t=2.0)
    run("small256_mean", 256, "mean", gname="small")
    run("small264_softmax", 264, "softmax_sg", gname="small")     # more than one channel block
    for aggr, kw in [("softmax", dict(t=0.9, learn_t=True)), ("softmax_sg", dict(t=0.1)),
                     ("power", dict(p=2.0, learn_p=True)), ("max", {}), ("add", {}), ("mean", {})]:
        run(f"small16_ea_{aggr}", 16, aggr, edge_attr=True, gname="small", **kw)
    return dict(graphs=graphs, cases=cases)


def sparse_module_cases(sparse):
    cases = []
    ei = synth.tricky_graph()
    N = 257

    def finish(name, mod, x, out, extra=None):
        probe = _probe(out.shape, 99)
        ps = [p for p in mod.parameters() if p.requires_grad]
        grads = torch.autograd.grad((out * probe).sum(), [x] + ps, allow_unused=True)
        rec = dict(name=name, edge_index=ei, x=x.detach(), probe=probe, out=out.detach(),
                   grad_x=grads[0], state_dict={k: v.clone() for k, v in mod.state_dict().items()},
                   param_grads={n: g for (n, p), g in zip([(n, p) for n, p in mod.named_parameters()
                                                           if p.requires_grad], grads[1:])})
        rec.update(extra or {})
        cases.append(rec)

    torch.manual_seed(11)
    x = torch.randn(N, 64, requires_grad=True)
    kw = dict(aggr="softmax", t=1.0, learn_t=True, msg_norm=True, learn_msg_scale=True,
              norm="batch", mlp_layers=2)
    conv = sparse.GENConv(64, 64, **kw)
    sd0 = {k: v.clone() for k, v in conv.state_dict().items()}
    finish("genconv_softmax_msgnorm", conv, x, conv(x, ei), dict(ctor=dict(in_dim=64, emb_dim=64, **kw), state_dict_before=sd0))

    torch.manual_seed(12)
    x = torch.randn(N, 32, requires_grad=True)
    kw = dict(aggr="power", p=2.0, learn_p=True, norm="layer", mlp_layers=1, encode_edge=True, edge_feat_dim=8)
    conv = sparse.GENConv(32, 48, **kw)
    ea = torch.rand(ei.size(1), 8)
    sd0 = {k: v.clone() for k, v in conv.state_dict().items()}
    finish("genconv_power_edge_encoder", conv, x, conv(x, ei, ea),
           dict(ctor=dict(in_dim=32, emb_dim=48, **kw), edge_attr=ea, state_dict_before=sd0))

    for convname in ["mr", "edge"]:
        torch.manual_seed(13)
        x = torch.randn(N, 24, requires_grad=True)
        m = sparse.GraphConv(24, 40, convname, "relu", "batch", True)
        sd0 = {k: v.clone() for k, v in m.state_dict().items()}
        finish(f"graphconv_{convname}", m, x, m(x, ei),
               dict(ctor=dict(in_channels=24, out_channels=40, conv=convname, act="relu", norm="batch", bias=True),
                    state_dict_before=sd0))
    torch.manual_seed(14)
    x = torch.randn(N, 50, requires_grad=True)   # PPI width
    m = sparse.ResGraphBlock(50, "mr", "relu", "batch", True, 8, 1)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    finish("resgraphblock_mr50", m, x, m(x, ei)[0],
           dict(ctor=dict(channels=50, conv="mr", act="relu", norm="batch", bias=True, heads=8, res_scale=1),
                state_dict_before=sd0))
    return cases


def dense_cases(dense):
    cases = []
    # kNN on lattice clouds (exact fp32 distance arithmetic)
    for name, B, C, N, k, d in [("knn_xyz", 2, 3, 256, 16, 1), ("knn_xyz_d4", 2, 3, 256, 8, 4),
                                ("knn_feat64", 2, 64, 192, 16, 3), ("knn_feat64_big", 1, 64, 512, 16, 27)]:
        x = synth.lattice_cloud(B, C, N, seed=len(cases) + 1)
        full = dense.dense_knn_matrix(x, k * d)
        dil = dense.DenseDilatedKnnGraph(k, d)(x)
        xt = x.transpose(2, 1).squeeze(-1)
        dist = dense.pairwise_distance(xt)
        cases.append(dict(kind="knn", name=name, x=x, k=k, dilation=d,
                          edge_index_full=None if N > 256 else full.to(torch.int32),
                          edge_index=dil.contiguous().to(torch.int32), dist=dist))
    # random (non-lattice) cloud: only distance-rank consistency is checked
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 16, 200, 1, generator=g)
    cases.append(dict(kind="knn", name="knn_randn16", x=x, k=9, dilation=2,
                      edge_index_full=dense.dense_knn_matrix(x, 18).to(torch.int32),
                      edge_index=dense.DenseDilatedKnnGraph(9, 2)(x).contiguous().to(torch.int32),
                      dist=dense.pairwise_distance(x.transpose(2, 1).squeeze(-1))))

    # sparse-layout kNN (gcn_lib/sparse/torch_edge.py) and the self-excluding torch_cluster variants, on lattice clouds
    import importlib
    sp_edge = importlib.import_module("gcn_lib.sparse.torch_edge")
    x = synth.lattice_cloud(3, 6, 160, seed=77)
    flat = x.squeeze(-1).transpose(1, 2).reshape(3 * 160, 6).contiguous()
    batch = torch.arange(3).repeat_interleave(160)
    cases.append(dict(kind="knn_sparse", name="sparse_knn_matrix", x=x, flat=flat, batch=batch, k=6, dilation=2,
                      edge_index=sp_edge.DilatedKnnGraph(6, 2, knn="matrix")(flat, batch).to(torch.int32),
                      edge_index_tree=sp_edge.DilatedKnnGraph(6, 2, knn="tree")(flat, batch).to(torch.int32),
                      dense_tree=dense.DilatedKnnGraph(6, 2)(x).contiguous().to(torch.int32),
                      dist=dense.pairwise_distance(x.transpose(2, 1).squeeze(-1))))

    def conv_case(name, cls, Cin, Cout, B, N, k, norm, act="relu", seed=0):
        torch.manual_seed(seed)
        x = torch.randn(B, Cin, N, 1, requires_grad=True)
        ei = dense.dense_knn_matrix(x.detach(), k)
        m = cls(Cin, Cout, act, norm, True)
        if norm == "batch":  # mixed-sign BN weights exercise the max/min trick
            bn = [mm for mm in m.modules() if isinstance(mm, torch.nn.BatchNorm2d)][0]
            with torch.no_grad():
                bn.weight.copy_(torch.randn(Cout))
                bn.bias.copy_(torch.randn(Cout) * 0.1)
        sd0 = {kk: v.clone() for kk, v in m.state_dict().items()}
        m.train()
        out = m(x, ei)
        probe = _probe(out.shape, 77)
        ps = [p for p in m.parameters()]
        grads = torch.autograd.grad((out * probe).sum(), [x] + ps)
        sd1 = {kk: v.clone() for kk, v in m.state_dict().items()}
        m.eval()
        with torch.no_grad():
            out_eval = m(x, ei)
        cases.append(dict(kind="conv", name=name, cls=cls.__name__, Cin=Cin, Cout=Cout, norm=norm, act=act,
                          x=x.detach(), edge_index=ei, probe=probe, out=out.detach(), out_eval=out_eval,
                          grad_x=grads[0], state_dict_before=sd0, state_dict_after=sd1,
                          param_grads={n: g for (n, _), g in zip(m.named_parameters(), grads[1:])}))

    conv_case("edgeconv_bn", dense.EdgeConv2d, 16, 32, 2, 128, 8, "batch", seed=1)
    conv_case("edgeconv_nonorm", dense.EdgeConv2d, 9, 64, 2, 96, 16, None, seed=2)
    conv_case("edgeconv_leaky_bn", dense.EdgeConv2d, 16, 16, 1, 64, 4, "batch", act="leakyrelu", seed=3)
    conv_case("mrconv_bn", dense.MRConv2d, 16, 32, 2, 128, 8, "batch", seed=4)
    conv_case("mrconv_nonorm", dense.MRConv2d, 12, 20, 2, 80, 6, None, seed=5)

    # a residual dynamic block end to end (kNN on features + EdgeConv + residual)
    torch.manual_seed(6)
    x = synth.lattice_cloud(2, 16, 128, seed=21).requires_grad_(True)
    blk = dense.ResDynBlock2d(16, 8, 2, "edge", "relu", "batch", True, False, 0.0, "matrix", 1)
    sd0 = {kk: v.clone() for kk, v in blk.state_dict().items()}
    blk.train()
    out = blk(x)
    probe = _probe(out.shape, 78)
    grads = torch.autograd.grad((out * probe).sum(), [x] + list(blk.parameters()))
    cases.append(dict(kind="block", name="resdynblock_edge", x=x.detach(), probe=probe, out=out.detach(),
                      grad_x=grads[0], state_dict_before=sd0,
                      param_grads={n: g for (n, _), g in zip(blk.named_parameters(), grads[1:])},
                      ctor=dict(in_channels=16, kernel_size=8, dilation=2, conv="edge", act="relu",
                                norm="batch", bias=True, stochastic=False, epsilon=0.0, knn="matrix",
                                res_scale=1)))
    return cases


def model_key_tables():
    """state_dict key -> shape of the reference's example architectures built on the REFERENCE gcn_lib."""
    import io
    import tempfile
    from contextlib import redirect_stdout
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ref_models
    out = {}
    with redirect_stdout(io.StringIO()), tempfile.TemporaryDirectory() as tmp:
        models = dict(sem_seg_dense_resgcn28=ref_models.dense_deepgcn(28),
                      ogbn_arxiv_deepergcn28=ref_models.arxiv_deepergcn(28),
                      ogbn_proteins_style_learn=ref_models.arxiv_deepergcn(3, gcn_aggr="softmax", learn_t=True,
                                                                       msg_norm=True, learn_msg_scale=True, mlp_layers=2, norm="layer"),
                      ppi_deepgcn_mr=ref_models.ppi_deepgcn("mr"), ppi_deepgcn_edge=ref_models.ppi_deepgcn("edge"),
                      proteins_revgcn=ref_models.proteins_revgcn(tmp))
    for name, m in models.items():
        out[name] = dict(keys={k: tuple(v.shape) for k, v in m.state_dict().items()},
                         n_params=sum(p.numel() for p in m.parameters()))
    return out


def full_model_cases():
    """Small instances of the reference's example models, REAL architecture files on the REFERENCE gcn_lib."""
    import argparse
    import io
    from contextlib import redirect_stdout
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ref_models
    cases = []

    def record(name, model, inputs, ctor):
        model.train()
        sd0 = {k: v.clone() for k, v in model.state_dict().items()}
        out = model(*inputs)
        probe = _probe(out.shape, 5)
        wrt = [t for t in inputs if t.is_floating_point() and t.requires_grad]
        (out * probe).sum().backward()          # (.backward, not autograd.grad: re-entrant checkpointing)
        grads = [t.grad.clone() for t in wrt]
        cases.append(dict(name=name, ctor=ctor, inputs=[t.detach() for t in inputs], probe=probe, out=out.detach(),
                          grads=[g for g in grads], state_dict_before=sd0))

    with redirect_stdout(io.StringIO()):
        torch.manual_seed(31)
        m = ref_models.dense_deepgcn(4, n_filters=32, k=8, stochastic=False, epsilon=0.0, dropout=0.0)
        pos = synth.lattice_cloud(2, 3, 256, seed=9)
        x = torch.cat([pos, torch.rand(2, 6, 256, 1)], dim=1).requires_grad_(True)
        record("sem_seg_dense_resgcn4", m, [x], dict(n_blocks=4, channels=32, k=8))

        torch.manual_seed(32)
        m = ref_models.arxiv_deepergcn(8, in_channels=32, hidden_channels=64, num_tasks=10, dropout=0.0)
        ei = synth.tricky_graph()
        x = torch.randn(257, 32, requires_grad=True)
        record("ogbn_arxiv_deepergcn8_ckpt", m, [x, ei], dict(num_layers=8, in_channels=32, hidden=64, num_tasks=10))

        for conv in ("mr", "edge"):
            torch.manual_seed(33)
            m = ref_models.ppi_deepgcn(conv, dropout=0.0)
            x = torch.randn(257, 50, requires_grad=True)
            data = argparse.Namespace(x=x, edge_index=ei, batch=None)

            class _Wrap(torch.nn.Module):
                def __init__(self, inner):
                    super().__init__()
                    self.inner = inner

                def forward(self, x, edge_index):
                    return self.inner(argparse.Namespace(x=x, edge_index=edge_index, batch=None))

                def state_dict(self, *a, **k):
                    return self.inner.state_dict(*a, **k)
            record(f"ppi_deepgcn_{conv}", _Wrap(m), [x, ei], dict(conv=conv))
    return cases


def revgcn_cases():
    """BASELINE config 5 as the reference runs it: the REAL examples/ogb_eff/ogbn_proteins/model_rev.RevGCN on the
    REAL eff_gcn_modules/rev/{gcn_revop,memgcn,rev_layer}.py and the reference gcn_lib.sparse (third-party
    primitives from oracle/thirdparty.py).  One model-level Linear(8 -> hidden) edge embedding, repeated per group,
    every GENConv owning a Linear(hidden -> hidden/group) edge encoder (model_rev.py:45-55,98-107).
    Recorded: inputs, the shared dropout mask (drawn by the model from the seeded CPU RNG, re-drawn here with the
    same seed), the output of last_norm (the last deterministic tensor: F.dropout follows) and the gradients of
    L = sum(last_norm_out * probe) w.r.t. every parameter."""
    import io
    import tempfile
    from contextlib import redirect_stdout
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ref_models
    cases = []

    def run(name, graph, n, num_layers, hidden, aggr, seed, **over):
        with redirect_stdout(io.StringIO()), tempfile.TemporaryDirectory() as tmp:
            torch.manual_seed(seed)
            m = ref_models.proteins_revgcn(tmp, num_layers=num_layers, hidden=hidden, aggr=aggr, n_table=n, **over)
        m.train()
        g = torch.Generator().manual_seed(seed + 1)
        x = torch.rand(n, 8, generator=g)
        node_index = torch.randperm(n, generator=g)
        edge_attr = torch.rand(graph.size(1), 8, generator=g)
        sd0 = {k: v.clone() for k, v in m.state_dict().items()}
        keep = {}
        hook = m.last_norm.register_forward_hook(lambda mod, i, o: keep.__setitem__("hn", o))
        torch.manual_seed(seed + 2)
        pred = m(x, node_index, graph, edge_attr)
        hook.remove()
        torch.manual_seed(seed + 2)          # first RNG consumer of forward() is the shared mask (model_rev.py:101)
        mask = torch.zeros(n, hidden).bernoulli_(1 - m.dropout) / (1 - m.dropout)
        hn = keep["hn"]
        probe = _probe(hn.shape, 6)
        (hn * probe).sum().backward()
        grads = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
        cases.append(dict(name=name, ctor=dict(num_layers=num_layers, hidden=hidden, aggr=aggr, dropout=m.dropout,
                                               learn_p=m.learn_p, **{k: v for k, v in over.items() if k in ("p", "t", "learn_t")}),
                          n=n, edge_index=graph, x=x, node_index=node_index, edge_attr=edge_attr, mask=mask,
                          node_table=m.node_features.clone(), state_dict_before=sd0, probe=probe, hn=hn.detach(),
                          pred_shape=tuple(pred.shape), grads=grads))

    tricky = synth.tricky_graph()
    small = synth.tricky_graph(n=64, e=700, hub_deg=300, seed=7)
    run("revgcn3_h64_max", tricky, 257, 3, 64, "max", 41)
    run("revgcn3_h64_power", tricky, 257, 3, 64, "power", 42, learn_p=True, p=2.0)
    run("revgcn1_h224_max", small, 64, 1, 224, "max", 43)                  # RevGNN-Wide width: C = 112 per group
    run("revgcn2_h224_power", small, 64, 2, 224, "power", 44, learn_p=True)
    run("revgcn2_h64_softmax", small, 64, 2, 64, "softmax", 45, learn_t=True, t=0.5)
    return cases


def main():
    os.makedirs(GOLD, exist_ok=True)
    dense, sparse = refshim.import_reference()
    if "--only-rev" in sys.argv:
        torch.set_num_threads(8)
        torch.save(revgcn_cases(), os.path.join(GOLD, "revgcn.pt"))
        print("revgcn.pt", os.path.getsize(os.path.join(GOLD, "revgcn.pt")) // 1024, "KiB")
        return
    torch.save(model_key_tables(), os.path.join(GOLD, "model_keys.pt"))
    torch.save(full_model_cases(), os.path.join(GOLD, "models.pt"))
    torch.set_num_threads(8)
    agg = sparse_aggregate_cases(sparse)
    torch.save(agg, os.path.join(GOLD, "sparse_aggregate.pt"))
    mods = sparse_module_cases(sparse)
    torch.save(mods, os.path.join(GOLD, "sparse_modules.pt"))
    dn = dense_cases(dense)
    torch.save(dn, os.path.join(GOLD, "dense.pt"))
    torch.save(revgcn_cases(), os.path.join(GOLD, "revgcn.pt"))
    for f in ("sparse_aggregate.pt", "sparse_modules.pt", "dense.pt", "revgcn.pt"):
        print(f, os.path.getsize(os.path.join(GOLD, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
