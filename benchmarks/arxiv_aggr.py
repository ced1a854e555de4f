#!/usr/bin/env python
"""Aggregation op alone on the ogbn-arxiv shape (N=169,343, E=2,484,941 with self loops, C=128): forward, forward+backward.
    python benchmarks/arxiv_aggr.py [aggr] [uniform|local]
'local' = a locality-ordered graph of the same size (neighbours within n/256 ids): the gathered rows then hit the XCD's L2
instead of the Infinity Cache, which separates "bound by the gather path" from "bound by instruction issue"."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep_gcns_torch_amd import ops, synth          # noqa: E402
from deep_gcns_torch_amd.graph import graph_of      # noqa: E402


def timed(fn, iters=50, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    aggr = sys.argv[1] if len(sys.argv) > 1 else "softmax_sg"
    if os.environ.get("DGCN_MAXTHR"):                       # A/B of the max backward: edges from which the bit-mask path runs
        ops.MAX_MASK_MIN_TABLE_BYTES = int(os.environ["DGCN_MAXTHR"])
    dev = torch.device("cuda:0")
    s = synth.SHAPES["arxiv"]
    kind = sys.argv[2] if len(sys.argv) > 2 else "uniform"
    gen = synth.local_graph if kind == "local" else synth.undirected_random_graph
    ei = gen(s["n"], s["n_undirected"], s["seed"], device=dev)
    g = graph_of(ei, s["n"])
    x = torch.randn(s["n"], 128, device=dev)
    xg = x.clone().requires_grad_(True)
    probe = torch.randn(s["n"], 128, device=dev)
    kw = dict(t=0.1) if aggr.startswith("softmax") else {}
    with torch.no_grad():
        f = timed(lambda: ops.gen_aggregate(x, g, aggr=aggr, **kw))

    def fb():
        xg.grad = None
        ops.gen_aggregate(xg, g, aggr=aggr, **kw).backward(probe)
    print(json.dumps(dict(aggr=aggr, graph=kind, N=s["n"], E=int(ei.size(1)), ms_fwd=f, ms_fwd_bwd=timed(fb))))


if __name__ == "__main__":
    main()
