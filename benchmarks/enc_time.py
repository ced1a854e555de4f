#!/usr/bin/env python
"""The per-edge encoder kernels (dgcn_gen_aggr_enc_{fwd,bwd}_f32) alone at the ogbn-proteins cluster shape
(N = 13,253, E = 791,225, 8 raw features -> C = 112 channels), per aggregator: forward and forward+backward ms.

    python benchmarks/enc_time.py [--iters 50]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--channels", type=int, default=112)
    a = ap.parse_args()
    from deep_gcns_torch_amd import ops, synth
    from deep_gcns_torch_amd.graph import Graph
    dev = torch.device("cuda:0")
    s = synth.SHAPES["proteins_cluster"]
    ei = synth.powerlaw_graph(s["n"], s["n_undirected"], s["seed"], device=dev)
    n, E, C = s["n"], ei.size(1), a.channels
    g = Graph.from_edge_index(ei, n)
    torch.manual_seed(0)
    x = torch.randn(n, C, device=dev)
    f8 = torch.rand(E, 8, device=dev)
    W = (torch.randn(C, 8, device=dev) / 3).requires_grad_(True)
    b = torch.randn(C, device=dev).requires_grad_(True)
    xq = x.clone().requires_grad_(True)
    go = torch.randn(n, C, device=dev)
    out = {}
    for aggr, kw in (("max", {}), ("power", dict(p=1.0)), ("softmax", dict(t=1.0)), ("mean", {})):
        with torch.no_grad():
            f = timed(lambda: ops.gen_aggregate(x, g, f8, aggr=aggr, edge_encoder=(W, b), **kw), a.iters)
        fb = timed(lambda: torch.autograd.grad(ops.gen_aggregate(xq, g, f8, aggr=aggr, edge_encoder=(W, b), **kw),
                                               [xq, W, b], go), a.iters)
        out[aggr] = dict(fwd_ms=f, fwd_bwd_ms=fb)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
