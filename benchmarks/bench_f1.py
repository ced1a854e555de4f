#!/usr/bin/env python
"""A/B timings of the fused node-wise layer pieces on one MI355X, one JSON line per entry.

    python benchmarks/bench_f1.py [--iters 20]

* rows_linear (csrc/rows_linear.hip) against the library GEMM (+ separate residual add / statistics pass);
* pre_activation (norm -> ReLU -> dropout in one pass) against the three stock steps;
* DeeperGCN-28 (arxiv shape) and DeeperGCN-14 (one products cluster) training steps: the reference-shaped layer loop
  against blocks.res_plus_layer, dropout as in the reference's defaults;
* RevGCN-8 step (BasicBlock with the fused pre-activation).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def timed(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--skip-models", action="store_true")
    ap.add_argument("--only", default="", help="run only the DeeperGCN-28 variant with this name (profiling)")
    args = ap.parse_args()
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()
    import arch_restated
    from deep_gcns_torch_amd import nn_util, node_ops, synth
    dev = torch.device("cuda:0")
    torch.manual_seed(0)

    def emit(**kw):
        print(json.dumps(kw), flush=True)

    # ---- the GEMM alone ------------------------------------------------------------------------------------------
    for rows, K, C in () if args.only else ((169343, 128, 128), (244902, 128, 128), (13253, 112, 224), (13253, 224, 112),
                                            (791225, 224, 112), (2449029, 128, 128)):
        x = torch.randn(rows, K, device=dev)
        w = torch.randn(C, K, device=dev) / K ** 0.5
        b = torch.randn(C, device=dev)
        r = torch.randn(rows, C, device=dev)
        g = torch.randn(rows, C, device=dev)
        with torch.no_grad():
            t_lib = timed(lambda: torch.nn.functional.linear(x, w, b), args.iters)
            t_lib_res = timed(lambda: torch.nn.functional.linear(x, w, b) + r, args.iters)
            t_k = timed(lambda: node_ops.rows_linear(x, w, b), args.iters)
            t_k_res = timed(lambda: node_ops.rows_linear(x, w, b, r), args.iters)
            t_k_res_st = timed(lambda: node_ops.rows_linear(x, w, b, r, want_stats=True), args.iters)
            t_dx_lib = timed(lambda: g @ w, args.iters)
            t_dx_k = timed(lambda: node_ops._rl_launch(g, w, True, None, None, False, False, C <= 128), args.iters)
            t_dw_k = timed(lambda: node_ops.rows_tn(g, x), args.iters)
            node_ops.ROWS_TN_KERNEL = False
            t_dw_lib = timed(lambda: node_ops.rows_tn(g, x), args.iters)
            node_ops.ROWS_TN_KERNEL = True
        byt = rows * (K + 2 * C) * 4
        emit(what="rows_linear", rows=rows, K=K, C=C, lib_ms=t_lib, lib_plus_residual_ms=t_lib_res, kernel_ms=t_k,
             kernel_residual_ms=t_k_res, kernel_residual_stats_ms=t_k_res_st, dx_lib_ms=t_dx_lib, dx_kernel_with_bias_grad_ms=t_dx_k,
             dw_lib_splitk_ms=t_dw_lib, dw_kernel_ms=t_dw_k,
             kernel_residual_GBs=byt / (t_k_res * 1e-3) / 1e9, kernel_TF=2.0 * rows * K * C / (t_k * 1e-3) / 1e12)
        del x, r, g

    # ---- norm -> relu -> dropout -----------------------------------------------------------------------------------
    for rows, C, kind in () if args.only else ((169343, 128, "batch"), (169343, 128, "layer"), (13253, 112, "layer")):
        x = torch.randn(rows, C, device=dev, requires_grad=True)
        go = torch.randn(rows, C, device=dev)
        norm = (node_ops.BatchNorm1d(C) if kind == "batch" else node_ops.LayerNorm(C)).to(dev).train()

        def stock():
            y = torch.nn.functional.dropout(torch.relu(norm(x)), p=0.5, training=True)
            torch.autograd.grad(y, [x] + list(norm.parameters()), go)

        def fused():
            y = node_ops.pre_activation(norm, x, p=0.5, training=True)
            torch.autograd.grad(y, [x] + list(norm.parameters()), go)
        emit(what="pre_activation fwd+bwd", rows=rows, C=C, norm=kind, three_steps_ms=timed(stock, args.iters),
             fused_ms=timed(fused, args.iters))
    if args.skip_models:
        return

    # ---- whole models ----------------------------------------------------------------------------------------------
    s = synth.SHAPES["arxiv"]
    ei = synth.undirected_random_graph(s["n"], s["n_undirected"], s["seed"], device=dev)
    xa = torch.randn(s["n"], 128, device=dev)
    ya = torch.randint(0, 40, (s["n"],), device=dev)
    for name, kw in (("plain_loop_no_dropout", dict(dropout=0.0)), ("plain_loop", dict(dropout=0.5)),
                     ("res_plus_layer", dict(dropout=0.5, fused_layers=True)),
                     ("res_plus_layer_full_recompute", dict(dropout=0.5, fused_layers=True, checkpoint="reference_full")),
                     ("res_plus_layer_no_checkpoint", dict(dropout=0.5, fused_layers=True, checkpoint="never")),
                     ("plain_loop_no_checkpoint", dict(dropout=0.5, checkpoint="never")),
                     ("res_plus_layer_library_gemm", dict(dropout=0.5, fused_layers=True))):
        if args.only and name != args.only:
            continue
        nn_util.ROWS_KERNEL = name != "res_plus_layer_library_gemm"
        m = arch_restated.DeeperGCN(num_layers=28, in_channels=128, hidden=128, num_tasks=40, **kw).to(dev).train()
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)

        def step():
            opt.zero_grad(set_to_none=True)
            torch.nn.functional.nll_loss(m(xa, ei), ya).backward()
            opt.step()
        emit(what="DeeperGCN-28 arxiv train step", variant=name, ms=timed(step, 5, 2))
        del m, opt
    nn_util.ROWS_KERNEL = True
    if args.only:
        return
    sp = synth.SHAPES["products"]
    n_c = sp["n"] // 10
    ei_c = synth.undirected_random_graph(n_c, sp["n_undirected"] // 100, sp["seed"] + 1, device=dev)
    xc = torch.randn(n_c, 100, device=dev)
    yc = torch.randint(0, 47, (n_c,), device=dev)
    for name, kw in (("plain_loop", dict(dropout=0.5)), ("res_plus_layer", dict(dropout=0.5, fused_layers=True)),
                     ("res_plus_layer_no_checkpoint", dict(dropout=0.5, fused_layers=True, checkpoint="never"))):
        m = arch_restated.DeeperGCN(num_layers=14, in_channels=100, hidden=128, num_tasks=47, **kw).to(dev).train()
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)

        def step():
            opt.zero_grad(set_to_none=True)
            torch.nn.functional.nll_loss(m(xc, ei_c), yc).backward()
            opt.step()
        emit(what="DeeperGCN-14 products-cluster train step", variant=name, ms=timed(step, 5, 2))
        del m, opt


if __name__ == "__main__":
    main()
