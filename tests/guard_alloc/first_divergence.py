#!/usr/bin/env python
"""Which module's FORWARD output first differs between the first and the second eager step of the fused RevGCN."""
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
from deep_gcns_torch_amd import fuse, ops  # noqa: E402

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
if os.environ.get("DGCN_STATIC_ITEMS"):
    ops.ENC_STATIC_ITEMS = True
if os.environ.get("DGCN_NO_KEEP"):
    from deep_gcns_torch_amd.eff_gcn_modules.rev import gcn_revop
    gcn_revop.KEEP_AGGREGATION = False
dev = torch.device("cuda:0")
inp = cr.revgcn_inputs()
m = rev_restated.RevGCNModelFile(num_layers=layers, hidden=224, aggr="max", dropout=0.0, node_table=inp["table"].to(dev),
                                 impl="product")
cr.formula_init(m, seed=5)
m = fuse.fuse_model(m.to(dev).train())
x, nidx, ei, ea = (inp[k].to(dev) for k in ("x", "node_index", "edge_index", "edge_attr"))
probe = inp["probe"].to(dev)
rec, state = [], {"on": False}


def hook(name):
    def h(mod, i, o):
        if state["on"]:
            t = o[0] if isinstance(o, tuple) else o
            rec[-1].append((name, t.detach().clone()))
    return h


m.node_features_encoder.register_forward_hook(hook("node_features_encoder"))
for l, w in enumerate(m.gcns):
    for i, fm in enumerate(w._fn.Fms):
        fm.norm.register_forward_hook(hook(f"layer {l} group {i} norm"))
        fm.gcn.register_forward_hook(hook(f"layer {l} group {i} GENConv"))
    w.register_forward_hook(hook(f"layer {l} output"))
m.last_norm.register_forward_hook(hook("last_norm"))
keep = {}
m.last_norm.register_forward_hook(lambda mod, i, o: keep.__setitem__("hn", o))
for it in range(3):
    for p in m.parameters():
        p.grad = None
    rec.append([])
    state["on"] = True
    m(x, nidx, ei, ea)
    state["on"] = False
    (keep["hn"] * probe).sum().backward()
    torch.cuda.synchronize()
for it in (1, 2):
    first = None
    for (n0, a), (n1, b) in zip(rec[0], rec[it]):
        assert n0 == n1
        d = (a - b).abs()
        if float(d.max()) > 0:
            rows = (d.reshape(d.size(0), -1).max(1).values > 0).nonzero().flatten()
            first = f"{n0}: max diff {float(d.max()):.3e} in {rows.numel()} rows {rows[:8].tolist()}"
            break
    print(f"step {it} vs step 0: first differing forward output: {first}", flush=True)
