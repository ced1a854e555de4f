#!/usr/bin/env python
"""One shape of the transposed row GEMM (csrc/rows_tn.hip), for rocprofv3:  python benchmarks/rows_tn_only.py ROWS C K"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep_gcns_torch_amd import node_ops  # noqa: E402

rows, C, K = (int(a) for a in sys.argv[1:4])
dev = torch.device("cuda:0")
g = torch.randn(rows, C, device=dev)
x = torch.randn(rows, K, device=dev)
for _ in range(30):
    node_ops.rows_tn(g, x)
torch.cuda.synchronize()
