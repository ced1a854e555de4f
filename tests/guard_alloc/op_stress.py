#!/usr/bin/env python
"""Bit-reproducibility stress of the pieces of one GENBlock forward (fused RevGCN route, ogbn-proteins cluster shape):
each piece is launched ITER times on the same inputs and every output compared with the first bit for bit."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import deep_gcns_torch_amd  # noqa: E402

deep_gcns_torch_amd.install()
from deep_gcns_torch_amd import ops, synth  # noqa: E402
from gcn_lib.sparse.torch_nn import MLP, norm_layer  # noqa: E402

ITER = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
dev = torch.device("cuda:0")
s = synth.SHAPES["proteins_cluster"]
ei = synth.powerlaw_graph(s["n"], s["n_undirected"], s["seed"], device=dev)
n, E, C = s["n"], ei.size(1), 112
gen = torch.Generator(device=dev).manual_seed(1)
y = torch.randn(n, 2 * C, device=dev, generator=gen)
xv = y[:, :C]                                  # a strided column view, as the coupling hands it to the blocks
feat = torch.rand(E, 8, device=dev, generator=gen)
W = torch.randn(C, 8, device=dev, generator=gen) / 3
b = torch.randn(C, device=dev, generator=gen)
ln = norm_layer("layer", C).to(dev)
mlp = MLP([C, 2 * C, C], norm="layer", last_lin=True).to(dev)
other = torch.randn(n, C, device=dev, generator=gen)
outbuf = torch.empty(n, 2 * C, device=dev)


def run(name, fn):
    with torch.no_grad():
        ref = fn().clone()
        bad = 0
        rows = set()
        for it in range(ITER):
            o = fn()
            if not torch.equal(o, ref):
                bad += 1
                d = (o - ref).abs()
                rows.update((d.reshape(d.size(0), -1).max(1).values > 0).nonzero().flatten().tolist()[:4])
        torch.cuda.synchronize()
    print(f"{name}: {bad} of {ITER} launches differ from the first" + (f" (rows {sorted(rows)[:12]})" if bad else ""), flush=True)


run("LayerNorm + ReLU on a strided view", lambda: ln(xv, True) if "fuse_relu" in ln.forward.__code__.co_varnames else torch.relu(ln(xv)))
run("per-edge encoder aggregation, max, add_root", lambda: ops.gen_aggregate(other, ei, feat, aggr="max", edge_encoder=(W, b), add_root=True))
run("MLP (rows_linear -> LayerNorm+ReLU -> rows_linear)", lambda: mlp(other))
run("torch.add into a column view", lambda: torch.add(xv, other, out=outbuf[:, C:]))
run("sum of chunks", lambda: sum(torch.chunk(y, 2, dim=1)[1:]) + 0)
