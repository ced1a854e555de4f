#!/usr/bin/env python
"""One per-edge encoder kernel in a loop (profiling target): ogbn-proteins cluster shape, C = 112.

    python benchmarks/enc_only.py [--aggr max] [--bwd] [--iters 30]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--aggr", default="max")
    ap.add_argument("--bwd", action="store_true")
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    from deep_gcns_torch_amd import ops, synth
    from deep_gcns_torch_amd.graph import Graph
    dev = torch.device("cuda:0")
    s = synth.SHAPES["proteins_cluster"]
    ei = synth.powerlaw_graph(s["n"], s["n_undirected"], s["seed"], device=dev)
    n, E, C = s["n"], ei.size(1), 112
    g = Graph.from_edge_index(ei, n)
    torch.manual_seed(0)
    x = torch.randn(n, C, device=dev, requires_grad=a.bwd)
    f8 = torch.rand(E, 8, device=dev)
    W = (torch.randn(C, 8, device=dev) / 3).requires_grad_(a.bwd)
    b = torch.randn(C, device=dev).requires_grad_(a.bwd)
    go = torch.randn(n, C, device=dev)
    kw = dict(p=1.0) if a.aggr == "power" else (dict(t=1.0) if a.aggr == "softmax" else {})
    for _ in range(a.iters):
        if a.bwd:
            torch.autograd.grad(ops.gen_aggregate(x, g, f8, aggr=a.aggr, edge_encoder=(W, b), **kw), [x, W, b], go)
        else:
            with torch.no_grad():
                ops.gen_aggregate(x, g, f8, aggr=a.aggr, edge_encoder=(W, b), **kw)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
