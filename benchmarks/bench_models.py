#!/usr/bin/env python
"""End-to-end step times of the BASELINE configs' model shapes on this package's gcn_lib (synthetic data,
random-init weights): ResGCN-28 dense (config 2) and DeeperGCN-28 on the arxiv shape (config 3); config 5 (RevGCN, per-layer edge encoder
on the model-level edge embedding) lives in benchmarks/bench_revgcn.py.  One JSON line each.
    python benchmarks/bench_models.py [--iters 10]
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
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()
    import arch_restated
    from deep_gcns_torch_amd import synth
    from gcn_lib.sparse.torch_vertex import GENConv
    dev = torch.device("cuda:0")
    torch.manual_seed(0)

    # config 2: ResGCN-28 (examples/sem_seg_dense/config.py defaults), B=8, N=4096, k=16
    m = arch_restated.DenseDeepGCN(n_blocks=28, channels=64, k=16, in_channels=9, n_classes=13).to(dev).train()
    x = torch.cat([torch.rand(8, 3, 4096, 1), torch.rand(8, 6, 4096, 1)], 1).to(dev)
    y = torch.randint(0, 13, (8, 4096), device=dev)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)

    def step_dense():
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(m(x), y)
        loss.backward()
        opt.step()
    ms = timed(step_dense, a.iters)
    with torch.no_grad():
        ms_f = timed(lambda: m(x), a.iters)
    edges = 8 * 4096 * 16 * 28
    print(json.dumps(dict(model="ResGCN-28 dense (B=8,N=4096,k=16, dilation 1..27), train step fwd+bwd+Adam",
                          ms_per_step=ms, ms_forward=ms_f, edges_per_s=edges / (ms * 1e-3),
                          params=sum(p.numel() for p in m.parameters()))), flush=True)
    # the same step replayed as ONE HIP graph: what the ~890 launches per step cost on the host side
    try:
        opt_g = torch.optim.Adam(m.parameters(), lr=1e-3, capturable=True)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                opt_g.zero_grad(set_to_none=True)
                torch.nn.functional.cross_entropy(m(x), y).backward()
                opt_g.step()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        opt_g.zero_grad(set_to_none=True)
        with torch.cuda.graph(graph):
            loss_g = torch.nn.functional.cross_entropy(m(x), y)
            loss_g.backward()
            opt_g.step()
        ms_g = timed(graph.replay, a.iters)
        print(json.dumps(dict(model="ResGCN-28 dense train step, whole step replayed as ONE HIP graph",
                              ms_per_step=ms_g, edges_per_s=edges / (ms_g * 1e-3),
                              loss_finite=bool(torch.isfinite(loss_g).item()))), flush=True)
    except Exception as exc:   # report, do not hide
        print(json.dumps(dict(model="ResGCN-28 HIP-graph replay", error=repr(exc)[:300])), flush=True)
    del m, opt

    # config 3: DeeperGCN-28 softmax_sg t=0.1 on the arxiv shape (full graph)
    s = synth.SHAPES["arxiv"]
    ei = synth.undirected_random_graph(s["n"], s["n_undirected"], s["seed"], device=dev)
    m = arch_restated.DeeperGCN(num_layers=28, in_channels=128, hidden=128, num_tasks=40).to(dev).train()
    xa = torch.randn(s["n"], 128, device=dev)
    ya = torch.randint(0, 40, (s["n"],), device=dev)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)

    def step_arxiv():
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.nll_loss(m(xa, ei), ya)
        loss.backward()
        opt.step()
    ms = timed(step_arxiv, a.iters)
    print(json.dumps(dict(model="DeeperGCN-28 GENConv softmax_sg (arxiv shape N=169343 E=2484941 C=128), train step, "
                                "re-entrant checkpointing as in the reference", ms_per_step=ms,
                          edges_per_s=ei.size(1) * 28 / (ms * 1e-3))), flush=True)

    # the same step captured once into a HIP graph and replayed: every op of the path is capture-safe (no host
    # synchronisation, no allocation outside torch's allocator), so the ~1000 launches of a step cost one
    try:
        opt_g = torch.optim.Adam(m.parameters(), lr=1e-3, capturable=True)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                opt_g.zero_grad(set_to_none=True)
                torch.nn.functional.nll_loss(m(xa, ei), ya).backward()
                opt_g.step()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        opt_g.zero_grad(set_to_none=True)
        with torch.cuda.graph(graph):
            loss_g = torch.nn.functional.nll_loss(m(xa, ei), ya)
            loss_g.backward()
            opt_g.step()
        ms_g = timed(graph.replay, a.iters)
        print(json.dumps(dict(model="DeeperGCN-28 (arxiv shape) train step, whole step replayed as ONE HIP graph",
                              ms_per_step=ms_g, edges_per_s=ei.size(1) * 28 / (ms_g * 1e-3),
                              loss_finite=bool(torch.isfinite(loss_g).item()))), flush=True)
    except Exception as exc:   # report, do not hide
        print(json.dumps(dict(model="DeeperGCN-28 HIP-graph replay", error=repr(exc)[:300])), flush=True)
    del m, opt

    # config 5 (RevGCN on ogbn-proteins, Linear(hidden -> hidden/group) edge encoder per layer): benchmarks/bench_revgcn.py


if __name__ == "__main__":
    main()
