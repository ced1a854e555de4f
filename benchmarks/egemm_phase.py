#!/usr/bin/env python
"""Phase attribution of the fused edge-GEMM forward kernel (profiling builds: the DGCN_EG_DEBUG / DGCN_EG_WAVES switches
of csrc/gen_aggr_egemm.hip are read per call).  Prints the op time for every combination."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deep_gcns_torch_amd import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
s = synth.SHAPES["proteins_cluster"]
ei = synth.powerlaw_graph(s["n"], s["n_undirected"], s["seed"], device=dev)
E = ei.size(1)
x = torch.randn(s["n"], 112, device=dev)
feat = torch.randn(E, 448, device=dev)[:, :224]
W = torch.randn(112, 224, device=dev) / 15
b = torch.randn(112, device=dev)
aggr = sys.argv[1] if len(sys.argv) > 1 else "max"


def run():
    with torch.no_grad():
        return ops.gen_aggregate(x, ei, feat, aggr=aggr, edge_encoder=(W, b))


def timed(iters=20):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


for waves in ((sys.argv[2],) if len(sys.argv) > 2 else ("0",)):
    for dbg, name in ((0, "full"), (1, "no walk"), (2, "no MFMA chain"), (3, "no walk, no MFMA"), (4, "no feature loads"),
                      (5, "no feature loads, no walk"), (16, "full + stagger")):
        os.environ["DGCN_EG_DEBUG"] = str(dbg)
        if waves != "0":
            os.environ["DGCN_EG_WAVES"] = waves
        print(json.dumps(dict(waves_per_wg=int(waves), variant=name, ms=round(timed(), 4))), flush=True)
