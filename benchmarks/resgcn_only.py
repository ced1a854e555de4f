#!/usr/bin/env python
"""ResGCN-28 (sem_seg_dense, B = 8 x 4096 points) training steps only, for rocprofv3:  python benchmarks/resgcn_only.py [steps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import deep_gcns_torch_amd  # noqa: E402

deep_gcns_torch_amd.install()
import arch_restated  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = arch_restated.DenseDeepGCN(n_blocks=28, channels=64, k=16, in_channels=9, n_classes=13).to(dev).train()
x = torch.cat([torch.rand(8, 3, 4096, 1), torch.rand(8, 6, 4096, 1)], 1).to(dev)
y = torch.randint(0, 13, (8, 4096), device=dev)
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
for _ in range(steps):
    opt.zero_grad(set_to_none=True)
    torch.nn.functional.cross_entropy(m(x), y).backward()
    opt.step()
torch.cuda.synchronize()
