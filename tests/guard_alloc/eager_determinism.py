#!/usr/bin/env python
"""Run-to-run determinism of the eager fused RevGCN step (forward output of last_norm), optionally under the guard
allocator with poisoned allocations (run.py --poison): a kernel that reads memory nobody wrote shows up as NaN / garbage."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import conftest  # noqa: E402

conftest.install_guard_allocator()
import deep_gcns_torch_amd  # noqa: E402

deep_gcns_torch_amd.install()
import config_replays as cr  # noqa: E402
import rev_restated  # noqa: E402
from deep_gcns_torch_amd import fuse, ops  # noqa: E402

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 112
aggr = sys.argv[2] if len(sys.argv) > 2 else "max"
fwd_only = len(sys.argv) > 3 and sys.argv[3] == "fwd"
if os.environ.get("DGCN_STATIC_ITEMS"):
    ops.ENC_STATIC_ITEMS = True
dev = torch.device("cuda:0")
inp = cr.revgcn_inputs()
m = rev_restated.RevGCNModelFile(num_layers=layers, hidden=224, aggr=aggr, dropout=0.0, learn_p=aggr == "power", p=1.0,
                                 node_table=inp["table"].to(dev), impl="product")
cr.formula_init(m, seed=5)
m = fuse.fuse_model(m.to(dev).train())
x, nidx, ei, ea = (inp[k].to(dev) for k in ("x", "node_index", "edge_index", "edge_attr"))
probe = inp["probe"].to(dev)
keep = {}
m.last_norm.register_forward_hook(lambda mod, i, o: keep.__setitem__("hn", o.detach().clone()))
outs = []
for it in range(4):
    for p in m.parameters():
        p.grad = None
    if fwd_only:
        with torch.no_grad():
            m(x, nidx, ei, ea)
    else:
        m(x, nidx, ei, ea)
        hn = keep["hn"]
    torch.cuda.synchronize()
    outs.append(keep["hn"].clone())
    if not fwd_only:
        # backward through a fresh forward (the hook cloned a detached copy)
        pred = m(x, nidx, ei, ea)
        pred.sum().backward()
        torch.cuda.synchronize()
for it in range(1, 4):
    d = (outs[it] - outs[0]).abs()
    rows = (d.max(1).values > 1e-6).nonzero().flatten()
    print(f"step {it} vs step 0: max diff {float(d.max()):.3e}, finite {bool(torch.isfinite(outs[it]).all())}, rows off: "
          f"{rows.numel()} {rows[:10].tolist()}", flush=True)
