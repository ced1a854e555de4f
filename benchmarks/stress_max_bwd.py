#!/usr/bin/env python
"""Bit-equality of the two max-backward paths (arg-max rows vs per-edge bit masks) on graphs of 2-3e5 nodes and 1e7 edges:
    python benchmarks/stress_max_bwd.py"""
import torch, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep_gcns_torch_amd import ops, synth
dev = torch.device("cuda:0")
for name, gen, n, m in (("uniform", synth.undirected_random_graph, 300_000, 5_000_000), ("powerlaw", synth.powerlaw_graph, 200_000, 4_000_000),
                        ("local", synth.local_graph, 300_000, 5_000_000)):
    ei = gen(n, m, seed=3, device=dev)
    for C in (128, 64, 100):
        x = torch.randn(n, C, device=dev)
        probe = torch.randn(n, C, device=dev)
        res = {}
        for tag, thr in (("rows", 1 << 60), ("mask", 0)):
            ops.MAX_MASK_MIN_TABLE_BYTES = thr
            xd = x.clone().requires_grad_(True)
            out = ops.gen_aggregate(xd, ei, aggr="max", add_root=True)
            (out * probe).sum().backward()
            res[tag] = xd.grad
        print(name, C, "equal" if torch.equal(res["rows"], res["mask"]) else "DIFFERENT", float(res["rows"].abs().sum()))
