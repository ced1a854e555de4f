#!/usr/bin/env python
"""kNN graph build alone at the config-2 layer shape (profiling target): B=8, N=4096, C=64, k=16, dilation d.

    python benchmarks/knn_only.py [--d 14] [--iters 30] [--fp32]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--d", type=int, default=14)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--fp32", action="store_true", help="distance pass on the fp32-MFMA filter kernel")
    ap.add_argument("--lds-lists", action="store_true", help="rounds 4 - 5's 16-row kernel with LDS lists")
    a = ap.parse_args()
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()
    from deep_gcns_torch_amd import dense_ops
    from gcn_lib.dense import DenseDilatedKnnGraph
    dense_ops.KNN_BF16_PIPE = not a.fp32
    dense_ops.KNN_GLOBAL_LISTS = not a.lds_lists
    torch.manual_seed(0)
    x = torch.randn(8, 64, 4096, 1, device="cuda:0")
    g = DenseDilatedKnnGraph(16, a.d)
    for _ in range(a.iters):
        g(x)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
