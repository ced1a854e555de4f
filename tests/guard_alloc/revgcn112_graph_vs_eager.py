#!/usr/bin/env python
"""RevGCN-112 (fused route): the replayed hipGraph step against the eager step, tensor by tensor."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import deep_gcns_torch_amd  # noqa: E402

deep_gcns_torch_amd.install()
import config_replays as cr  # noqa: E402
import rev_restated  # noqa: E402
from deep_gcns_torch_amd import fuse  # noqa: E402
from deep_gcns_torch_amd.graphs import GraphedStep  # noqa: E402

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 112
aggr = sys.argv[2] if len(sys.argv) > 2 else "max"
dev = torch.device("cuda:0")
inp = cr.revgcn_inputs()
m = rev_restated.RevGCNModelFile(num_layers=layers, hidden=224, aggr=aggr, dropout=0.0, learn_p=aggr == "power", p=1.0,
                                 node_table=inp["table"].to(dev), impl="product")
cr.formula_init(m, seed=5)
m = fuse.fuse_model(m.to(dev).train())
x, nidx, ei, ea = (inp[k].to(dev) for k in ("x", "node_index", "edge_index", "edge_attr"))
probe = inp["probe"].to(dev)
keep = {}
m.last_norm.register_forward_hook(lambda mod, i, o: keep.__setitem__("hn", o))
hn_s = torch.empty(inp["n"], 224, device=dev)


def step():
    for p in m.parameters():
        p.grad = None
    m(x, nidx, ei, ea)
    hn = keep["hn"]
    (hn * probe).sum().backward()
    with torch.no_grad():
        hn_s.copy_(hn)


import copy  # noqa: E402
m_graph = copy.deepcopy(m)                 # a model whose AccumulateGrad nodes are first created on the warm-up stream
m_graph.node_features = m.node_features
m_graph.last_norm.register_forward_hook(lambda mod, i, o: keep.__setitem__("hn", o))
step()
torch.cuda.synchronize()
hn_e = hn_s.clone()
g_e = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
step()
torch.cuda.synchronize()
print("eager vs eager: hn max diff", float((hn_s - hn_e).abs().max()), flush=True)
m = m_graph
g = GraphedStep(step, warmup=1)
prev = None
for rep in range(3):
    hn_s.fill_(float("nan"))
    g()
    torch.cuda.synchronize()
    d = (hn_s - hn_e).abs()
    rows = (d.max(1).values > 1e-5).nonzero().flatten()
    line = f"replay {rep}: hn max diff {float(d.max()):.3e}, rows off by > 1e-5: {rows.numel()} (first: {rows[:12].tolist()})"
    if prev is not None:
        line += f" | identical to the previous replay: {bool(torch.equal(prev, hn_s))}"
    prev = hn_s.clone()
    print(line)
worst = max(((k, float((p.grad - g_e[k]).abs().max() / (g_e[k].abs().max() + 1e-30))) for k, p in m.named_parameters() if k in g_e),
            key=lambda kv: kv[1])
print("worst parameter gradient, replay vs eager (max error / max):", worst)
