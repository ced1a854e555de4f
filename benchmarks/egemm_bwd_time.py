#!/usr/bin/env python
"""Backward of the fused edge GEMM under max aggregation at the ogbn-proteins cluster shape (N = 13,253, E = 791,225,
K = 224 features -> C = 112 channels), the two routes of ops._GenAggregate.backward:

  winners   csrc/egemm_max_bwd.hip: walk the (row, channel) arg-max winners, no (E, C) gradient
  dense     dz (E, C) written by the CSC walk, then dz @ W (rows_linear) and dz^T F (rows_tn)

    python benchmarks/egemm_bwd_time.py [--iters 30] [--graph powerlaw|uniform]

Prints forward, forward+backward and (by difference) backward ms, gradients accumulated into a running sink as the
reversible backward does, plus the share of edges that win at least one channel."""
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
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--graph", default="powerlaw")
    a = ap.parse_args()
    from deep_gcns_torch_amd import ops, synth
    from deep_gcns_torch_amd.graph import Graph
    dev = torch.device("cuda:0")
    s = synth.SHAPES["proteins_cluster"]
    if a.graph == "powerlaw":
        ei = synth.powerlaw_graph(s["n"], s["n_undirected"], s["seed"], device=dev)
    else:
        ei = synth.undirected_random_graph(s["n"], s["n_undirected"], s["seed"], device=dev)
    n, E, C, K = s["n"], ei.size(1), 112, 224
    g = Graph.from_edge_index(ei, n)
    torch.manual_seed(0)
    x = torch.randn(n, C, device=dev, requires_grad=True)
    feat = torch.randn(E, 2 * K, device=dev)[:, :K].requires_grad_(True)
    W = (torch.randn(C, K, device=dev) / 15).requires_grad_(True)
    b = torch.randn(C, device=dev, requires_grad=True)
    go = torch.randn(n, C, device=dev)
    sink = torch.zeros(E, K, device=dev)
    out = {"graph": a.graph, "E": E}
    with torch.no_grad():
        fwd = timed(lambda: ops.gen_aggregate(x, g, feat, aggr="max", edge_encoder=(W, b), add_root=True), a.iters)
    out["fwd_no_grad_ms"] = fwd

    def step():
        with ops.edge_grad_sink(feat, sink):
            o = ops.gen_aggregate(x, g, feat, aggr="max", edge_encoder=(W, b), add_root=True)
            torch.autograd.grad(o, [x, W, b, feat], go, allow_unused=True)
    for name, flag in (("winners", True), ("dense", False)):
        ops.EGEMM_MAX_WINNER_BWD = flag
        fb = timed(step, a.iters)
        out[name] = dict(fwd_bwd_ms=fb, bwd_ms_by_difference=fb - fwd)
    ops.EGEMM_MAX_WINNER_BWD = True
    sink.zero_()
    step()
    out["edges_with_a_winning_channel"] = float((sink != 0).any(1).float().mean())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
