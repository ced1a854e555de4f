#!/usr/bin/env python
"""BASELINE config 5 as the reference runs it (examples/ogb_eff/ogbn_proteins/model_rev.py): RevGCN with ONE
model-level Linear(8 -> hidden) edge embedding repeated per group and a Linear(hidden -> hidden/group) edge encoder
inside every GENConv, reversible additive coupling (forward under no_grad, inverse + recompute in the backward).
Synthetic ogbn-proteins-cluster-shaped graph (N=13,253, E=791,225, power law), random-init weights.

    python benchmarks/bench_revgcn.py [--layers 8] [--hidden 224] [--aggr max] [--iters 5] [--fused 0|1]

Prints one JSON line per measurement:
  * one GENConv (C = hidden/group, edge_feat_dim = hidden) forward and forward+backward on the strided group view
    of the edge embedding -- the unit the model repeats 2 x layers x (forward + inverse + recompute) times;
  * the whole train step (forward + backward) of an L-layer RevGCN, and the per-layer time it implies.
--rev product (default): this package's eff_gcn_modules.rev drop-in; --rev restated: the reference's reversible
algorithm restated in tests/rev_restated.py (the reference's own files do not travel to the GPU box).
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


def timed(fn, iters, warmup=2):
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
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--hidden", type=int, default=224)
    ap.add_argument("--aggr", default="max")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--fused", type=int, default=1, help="0: stock GEMM + (E,C) edge embedding; 1: fused edge GEMM kernels")
    ap.add_argument("--skip-model", action="store_true")
    ap.add_argument("--rev", default="product", choices=["product", "restated"],
                    help="product: this package's eff_gcn_modules.rev (fused reversible step); restated: the reference's "
                         "algorithm (tests/rev_restated.py: inverse + separate recompute, autograd-summed edge gradients)")
    a = ap.parse_args()
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()
    from deep_gcns_torch_amd import ops, synth
    from gcn_lib.sparse.torch_vertex import GENConv
    if hasattr(ops, "FUSED_EDGE_GEMM"):
        ops.FUSED_EDGE_GEMM = bool(a.fused)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    s = synth.SHAPES["proteins_cluster"]
    ei = synth.powerlaw_graph(s["n"], s["n_undirected"], s["seed"], device=dev)
    N, E = s["n"], ei.size(1)
    hidden, group = a.hidden, 2
    C = hidden // group
    kw = dict(p=1.0, learn_p=True) if a.aggr == "power" else {}
    base = dict(aggr=a.aggr, hidden=hidden, C=C, N=N, E=E, fused=bool(a.fused) and hasattr(ops, "FUSED_EDGE_GEMM"))

    # ---- one GENConv of the model: C channels, edge encoder Linear(hidden -> C) on a strided (E, hidden) view ----
    conv = GENConv(C, C, aggr=a.aggr, norm="layer", mlp_layers=2, encode_edge=True, edge_feat_dim=hidden, **kw).to(dev)
    edge_emb_full = torch.randn(E, hidden * group, device=dev)
    edge_emb = edge_emb_full[:, :hidden]                      # what torch.chunk hands to group 0 (row stride 2*hidden)
    x = torch.randn(N, C, device=dev)
    with torch.no_grad():
        ms_f = timed(lambda: conv(x, ei, edge_emb), a.iters)
    xr = x.clone().requires_grad_(True)
    er = edge_emb_full.clone().requires_grad_(True)

    def conv_step():
        out = conv(xr, ei, er[:, :hidden])
        torch.autograd.grad(out.sum(), [xr, er] + list(conv.parameters()))
    ms_fb = timed(conv_step, a.iters)
    flop = 2.0 * E * hidden * C
    # the aggregation op alone (edge encoder + gather + aggregate), as GENConv calls it
    W, bvec = conv.edge_encoder.weight.detach(), conv.edge_encoder.bias.detach()
    okw = dict(p=1.0) if a.aggr == "power" else {}

    def op_fwd():
        if ops.encoder_fusable(x, edge_emb, W):
            return ops.gen_aggregate(x, ei, edge_emb, aggr=a.aggr, edge_encoder=(W, bvec), add_root=True, **okw)
        return ops.gen_aggregate(x, ei, torch.nn.functional.linear(edge_emb, W, bvec), aggr=a.aggr, add_root=True, **okw)
    with torch.no_grad():
        ms_op = timed(op_fwd, a.iters)
    print(json.dumps(dict(base, what="aggregation op alone: edge encoder GEMM + gather + aggregate (+x), forward",
                          ms_forward=ms_op, edge_gemm_tflops=flop / (ms_op * 1e-3) / 1e12,
                          feature_gbps=E * hidden * 4 / (ms_op * 1e-3) / 1e9)), flush=True)
    print(json.dumps(dict(base, what="GENConv layer (edge encoder hidden->C + aggregation + MLP)", ms_forward=ms_f,
                          ms_fwd_bwd=ms_fb, edge_gemm_gflop=flop / 1e9,
                          fwd_edge_gemm_tflops_if_all_time=flop / (ms_f * 1e-3) / 1e12)), flush=True)
    del conv, xr, er, edge_emb_full
    if a.skip_model:
        return

    # ---- the whole model ----------------------------------------------------------------------------------------
    import rev_restated
    table = torch.rand(N, 8, device=dev)
    m = rev_restated.RevGCN(num_layers=a.layers, hidden=hidden, aggr=a.aggr, dropout=0.2, node_table=table,
                            learn_p=(a.aggr == "power"), impl=a.rev).to(dev).train()
    xin = torch.rand(N, 8, device=dev)
    node_index = torch.arange(N, device=dev)
    edge_attr = torch.rand(E, 8, device=dev)
    y = (torch.rand(N, 112, device=dev) > 0.5).float()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)

    def step():
        opt.zero_grad(set_to_none=True)
        pred, _ = m(xin, node_index, ei, edge_attr)
        loss = torch.nn.functional.binary_cross_entropy_with_logits(pred, y)
        loss.backward()
        opt.step()
    ms = timed(step, a.iters, warmup=2)
    print(json.dumps(dict(base, rev=a.rev, what=f"RevGCN-{a.layers} train step (fwd + inverse + recompute + bwd + Adam)",
                          ms_per_step=ms, ms_per_layer=ms / a.layers, layers=a.layers,
                          peak_mem_gb=torch.cuda.max_memory_allocated() / 2 ** 30,
                          edges_per_s=E * a.layers * group / (ms * 1e-3))), flush=True)


if __name__ == "__main__":
    main()
