#!/usr/bin/env python
"""RevGCN train steps only (ogbn-proteins cluster shape), for rocprofv3:
    python benchmarks/revgcn_only.py [layers] [aggr] [composed 0|1] [steps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import deep_gcns_torch_amd  # noqa: E402

deep_gcns_torch_amd.install()
import rev_restated  # noqa: E402
from deep_gcns_torch_amd import synth  # noqa: E402

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
aggr = sys.argv[2] if len(sys.argv) > 2 else "max"
composed = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 7
dev = torch.device("cuda:0")
torch.manual_seed(0)
s = synth.SHAPES["proteins_cluster"]
ei = synth.powerlaw_graph(s["n"], s["n_undirected"], s["seed"], device=dev)
N, E = s["n"], ei.size(1)
table = torch.rand(N, 8, device=dev)
xin, nidx = torch.rand(N, 8, device=dev), torch.arange(N, device=dev)
ea = torch.rand(E, 8, device=dev)
y = (torch.rand(N, 112, device=dev) > 0.5).float()
m = rev_restated.RevGCN(num_layers=layers, hidden=224, aggr=aggr, dropout=0.2, node_table=table, impl="product",
                        composed_edges=composed).to(dev).train()
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
for _ in range(steps):
    opt.zero_grad(set_to_none=True)
    pred, _ = m(xin, nidx, ei, ea)
    torch.nn.functional.binary_cross_entropy_with_logits(pred, y).backward()
    opt.step()
torch.cuda.synchronize()
