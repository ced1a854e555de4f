#!/usr/bin/env python
"""BatchNorm1d [+ReLU] on (rows, C) node features: HIP row kernels vs stock torch, fwd and fwd+bwd (arxiv / products rows)."""
import json
import os
import sys
import time

import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, iters=30, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    from deep_gcns_torch_amd.node_ops import BatchNorm1d
    dev = torch.device("cuda:0")
    for rows, C in ((169343, 128), (169343, 256), (2449029, 128)):
        x = torch.randn(rows, C, device=dev, requires_grad=True)
        g = torch.randn(rows, C, device=dev)
        ours, stock = BatchNorm1d(C).to(dev), nn.BatchNorm1d(C).to(dev)
        mb = rows * C * 4 / 1e6
        res = dict(rows=rows, C=C, tensor_MB=round(mb, 1))
        for name, f in (("hip_bn", lambda: ours(x)), ("hip_bn_relu", lambda: ours(x, fuse_relu=True)),
                        ("torch_bn", lambda: stock(x)), ("torch_bn_relu", lambda: torch.relu(stock(x)))):
            with torch.no_grad():
                res[name + "_fwd_ms"] = round(timed(f), 4)
            res[name + "_fwdbwd_ms"] = round(timed(lambda: torch.autograd.grad(f(), x, g)), 4)
        # streaming floor: fwd = 2 reads + 1 write, bwd = 4 reads + 1 write (+1 with the relu mask) of the tensor
        res["hip_fwd_GBs"] = round(3 * mb / res["hip_bn_fwd_ms"], 1)
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
