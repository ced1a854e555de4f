#!/usr/bin/env python
"""CPU baseline of the dense layer (the reference path restated in oracle/dense_ref.py) on this host's cores, same
synthetic shape as benchmarks/bench_dense.py.  Lives under tests/ because only tests, smoke() and bench.py's
cpu_baseline leg may execute the oracle.   python tests/cpu_baseline_dense.py [--B 8 --N 4096 --C 64 --k 16]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--C", type=int, default=64)
    ap.add_argument("--k", type=int, default=16)
    a = ap.parse_args()
    from oracle import dense_ref
    B, N, C, k = a.B, a.N, a.C, a.k
    torch.set_num_threads(os.cpu_count())
    torch.manual_seed(0)
    x = torch.randn(B, C, N, 1)
    go = torch.randn(B, C, N, 1)
    t0 = time.perf_counter()
    ei_full = dense_ref.dense_knn_matrix(x, k * 14)          # (2, B, N, k*14): neighbours, centres
    t_knn = time.perf_counter() - t0
    ei = dense_ref.dilate(ei_full, 14)
    conv = torch.nn.Sequential(torch.nn.Conv2d(2 * C, C, 1), torch.nn.ReLU(), torch.nn.BatchNorm2d(C)).train()
    xr = x.clone().requires_grad_(True)
    t0 = time.perf_counter()
    y = dense_ref.edgeconv2d(xr, ei, conv)
    t_f = time.perf_counter() - t0
    y.backward(go)
    t_fb = time.perf_counter() - t0
    edges = B * N * k
    print(json.dumps(dict(op="cpu_oracle", cores=os.cpu_count(), knn_K224_ms=t_knn * 1e3, edgeconv_fwd_ms=t_f * 1e3,
                          edgeconv_fwd_bwd_ms=t_fb * 1e3, edgeconv_fwd_bwd_edges_per_s=edges / t_fb, B=B, N=N, C=C, k=k)))


if __name__ == "__main__":
    main()
