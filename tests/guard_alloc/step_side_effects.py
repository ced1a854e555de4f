#!/usr/bin/env python
"""Does one eager training step (loss on last_norm's output, as tests/test_revgcn112_gpu.py) change anything it should
not: inputs, parameters, the probe; and is last_norm's output the same tensor after the backward?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import deep_gcns_torch_amd  # noqa: E402

deep_gcns_torch_amd.install()
import config_replays as cr  # noqa: E402
import rev_restated  # noqa: E402
from deep_gcns_torch_amd import fuse  # noqa: E402

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 112
dev = torch.device("cuda:0")
inp = cr.revgcn_inputs()
m = rev_restated.RevGCNModelFile(num_layers=layers, hidden=224, aggr="max", dropout=0.0, node_table=inp["table"].to(dev),
                                 impl="product")
cr.formula_init(m, seed=5)
m = fuse.fuse_model(m.to(dev).train())
x, nidx, ei, ea = (inp[k].to(dev) for k in ("x", "node_index", "edge_index", "edge_attr"))
probe = inp["probe"].to(dev)
keep = {}
m.last_norm.register_forward_hook(lambda mod, i, o: keep.update(hn=o, hn_clone=o.detach().clone()))


def sums():
    d = dict(x=x, ea=ea, probe=probe, table=m.node_features, ei=ei.float())
    d.update({k: p for k, p in m.named_parameters()})
    return {k: float(v.detach().double().abs().sum()) for k, v in d.items()}


before = sums()
prev_clone = None
for it in range(3):
    for p in m.parameters():
        p.grad = None
    m(x, nidx, ei, ea)
    hn = keep["hn"]
    (hn * probe).sum().backward()
    torch.cuda.synchronize()
    after = sums()
    changed = [k for k in before if before[k] != after[k]]
    d_after = float((hn.detach() - keep["hn_clone"]).abs().max())
    line = f"step {it}: persistent tensors changed: {changed[:6]}; hn read after the backward vs cloned in the forward: {d_after:.3e}"
    if prev_clone is not None:
        dd = (keep["hn_clone"] - prev_clone).abs()
        line += f"; forward clone vs previous step's: {float(dd.max()):.3e} in {int((dd.max(1).values > 1e-6).sum())} rows"
    prev_clone = keep["hn_clone"].clone()
    print(line, flush=True)
