#!/usr/bin/env python
"""kNN graph build at the config-2 layer shape, timed per dilation with events: B=8, N=4096, C=64, k=16.

    python benchmarks/knn_time.py [--iters 50]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--lds-lists", action="store_true", help="rounds 4 - 5's 16-row kernel with LDS lists")
    a = ap.parse_args()
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()
    from gcn_lib.dense import DenseDilatedKnnGraph
    from deep_gcns_torch_amd import dense_ops
    dense_ops.KNN_GLOBAL_LISTS = not a.lds_lists
    torch.manual_seed(0)
    x = torch.randn(8, 64, 4096, 1, device="cuda:0")
    out = {}
    for d in (1, 2, 7, 14, 20, 27):
        g = DenseDilatedKnnGraph(16, d)
        for _ in range(10):
            g(x)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(a.iters):
            g(x)
        e.record()
        torch.cuda.synchronize()
        out[f"d{d}_K{16 * d}_ms"] = s.elapsed_time(e) / a.iters
    print(json.dumps(out))


if __name__ == "__main__":
    main()
