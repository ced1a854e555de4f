#!/usr/bin/env python
"""The guard allocator checked against itself: torch ops and the device graph builder give the host's answers in a
process whose every allocation is guarded (run through tests/guard_alloc/run.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import conftest  # noqa: E402

assert conftest.install_guard_allocator(), "run with DGCN_GUARD_ALLOC=1"
from deep_gcns_torch_amd import synth  # noqa: E402
from deep_gcns_torch_amd.graph import Graph  # noqa: E402

dev = torch.device("cuda:0")
a = torch.arange(1000, device=dev, dtype=torch.float32)
assert float((a * 2).sum()) == 999000.0
z = torch.zeros(4097, device=dev, dtype=torch.int32)
assert int(z.sum()) == 0
z.fill_(3)
assert int(z.sum()) == 3 * 4097
print("torch ops ok", flush=True)
for rep in range(3):
    for n, e, hub, seed in ((257, 4099, 2100, 0), (64, 700, 300, 7), (1000, 20000, 900, 3)):
        ei = synth.tricky_graph(n=n, e=e, hub_deg=hub, seed=seed)
        gc = Graph.from_edge_index(ei, n)
        gd = Graph.from_edge_index(ei.to(dev), n)
        for name in ("rowptr", "col", "t_rowptr", "t_col", "t_eperm"):
            assert torch.equal(getattr(gc, name).cpu(), getattr(gd, name).cpu()), (rep, n, name)
        print("graph", rep, n, "ok", flush=True)
print("selftest ok")
