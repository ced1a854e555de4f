#!/usr/bin/env python
"""Dense point-cloud hot path micro-benchmark (BASELINE.md config 2 shapes: B=8, N=4096, k=16, C=64).

Prints one JSON line per measurement: kNN graph build at dilation 1/14/27, EdgeConv2d (relu, batch-norm)
forward and forward+backward and a full ResDynBlock2d step (the CPU oracle timed beside them: tests/cpu_baseline_dense.py).
    python benchmarks/bench_dense.py [--iters 20]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, iters, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return sum(ts) / len(ts), ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--C", type=int, default=64)
    ap.add_argument("--k", type=int, default=16)
    a = ap.parse_args()
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()
    from gcn_lib.dense import DenseDilatedKnnGraph, EdgeConv2d, ResDynBlock2d
    dev = torch.device("cuda:0")
    B, N, C, k = a.B, a.N, a.C, a.k
    torch.manual_seed(0)
    x = torch.randn(B, C, N, 1, device=dev)
    edges = B * N * k
    out = []

    from deep_gcns_torch_amd import dense_ops
    for pipe in (True, False):
        dense_ops.KNN_BF16_PIPE = pipe
        for d in (1, 7, 14, 27):
            g = DenseDilatedKnnGraph(k, d)
            avg, mn = timed(lambda: g(x), a.iters)
            flops = 2.0 * B * N * N * C
            out.append(dict(op="knn_dense", distance_pass="bf16x6 matrix pipe" if pipe else "fp32 MFMA", dilation=d, K=k * d,
                            ms_avg=avg, ms_min=mn, distance_tflops=flops / (avg * 1e-3) / 1e12,
                            rows_per_s=B * N / (avg * 1e-3)))
    dense_ops.KNN_BF16_PIPE = True
    ei = DenseDilatedKnnGraph(k, 1)(x)
    conv = EdgeConv2d(C, C, "relu", "batch", True).to(dev).train()
    xg = x.clone().requires_grad_(True)
    go = torch.randn(B, C, N, 1, device=dev)
    with torch.no_grad():
        avg, mn = timed(lambda: conv(x, ei), a.iters)
    out.append(dict(op="edgeconv2d_fwd", ms_avg=avg, ms_min=mn, edges_per_s=edges / (avg * 1e-3)))

    def fb():
        y = conv(xg, ei)
        torch.autograd.grad(y, [xg] + list(conv.parameters()), go)
    avg, mn = timed(fb, a.iters)
    out.append(dict(op="edgeconv2d_fwd_bwd", ms_avg=avg, ms_min=mn, edges_per_s=edges / (avg * 1e-3)))

    blk = ResDynBlock2d(C, k, 14, "edge", "relu", "batch", True).to(dev).train()

    def blk_fb():
        y = blk(xg)
        torch.autograd.grad(y, [xg] + list(blk.parameters()), go)
    avg, mn = timed(blk_fb, a.iters)
    out.append(dict(op="resdynblock2d_d14_fwd_bwd (knn + edgeconv + residual)", ms_avg=avg, ms_min=mn,
                    edges_per_s=edges / (avg * 1e-3)))

    # the same block step captured in a HIP graph (no host launch overhead): what the kernels alone cost
    params = [xg] + list(blk.parameters())
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            torch.autograd.grad(blk(xg), params, go)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        static = torch.autograd.grad(blk(xg), params, go)  # noqa: F841
    avg, mn = timed(graph.replay, a.iters)
    out.append(dict(op="resdynblock2d_d14_fwd_bwd, HIP-graph replay", ms_avg=avg, ms_min=mn,
                    edges_per_s=edges / (avg * 1e-3)))

    for r in out:
        r.update(B=B, N=N, C=C, k=k)
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
