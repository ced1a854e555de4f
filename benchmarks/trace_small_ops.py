#!/usr/bin/env python
"""Which Python lines launch the small ATen kernels (sums, copies, fills, adds) of a model step?

    python benchmarks/trace_small_ops.py revgcn8 | deepergcn28

One step under torch.profiler with stacks; prints, per (ATen op, innermost package frame), the number of calls and the
device time of one step.  Used to find launch-bound leftovers around the HIP kernels.
"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build(which, dev):
    import deep_gcns_torch_amd
    deep_gcns_torch_amd.install()
    from deep_gcns_torch_amd import synth
    if which == "revgcn8":
        import rev_restated
        s = synth.SHAPES["proteins_cluster"]
        ei = synth.powerlaw_graph(s["n"], s["n_undirected"], s["seed"], device=dev)
        N, E = s["n"], ei.size(1)
        table = torch.rand(N, 8, device=dev)
        m = rev_restated.RevGCN(num_layers=8, hidden=224, aggr="max", dropout=0.2, node_table=table, impl="product").to(dev).train()
        xin, nidx, ea = torch.rand(N, 8, device=dev), torch.arange(N, device=dev), torch.rand(E, 8, device=dev)
        y = (torch.rand(N, 112, device=dev) > 0.5).float()
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)

        def step():
            opt.zero_grad(set_to_none=True)
            pred, _ = m(xin, nidx, ei, ea)
            torch.nn.functional.binary_cross_entropy_with_logits(pred, y).backward()
            opt.step()
        return step
    import arch_restated
    s = synth.SHAPES["arxiv"]
    ei = synth.undirected_random_graph(s["n"], s["n_undirected"], s["seed"], device=dev)
    xa = torch.randn(s["n"], 128, device=dev)
    ya = torch.randint(0, 40, (s["n"],), device=dev)
    m = arch_restated.DeeperGCN(num_layers=28, in_channels=128, hidden=128, num_tasks=40, dropout=0.5,
                                fused_layers=True).to(dev).train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)

    def step():
        opt.zero_grad(set_to_none=True)
        torch.nn.functional.nll_loss(m(xa, ei), ya).backward()
        opt.step()
    return step


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "revgcn8"
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    step = build(which, dev)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        dt = getattr(ev, "device_time_total", 0) or getattr(ev, "cuda_time_total", 0)
        if not dt or not ev.name.startswith("aten::"):
            continue
        if ev.cpu_children and any(c.name.startswith("aten::") and (getattr(c, "device_time_total", 0) or 0) for c in ev.cpu_children):
            continue                                      # count the innermost ATen op only
        frame = "?"
        for fr in ev.stack or []:
            if "deep_gcns_torch_amd" in fr or "/tests/" in fr or "benchmarks" in fr:
                frame = fr.replace(ROOT + "/", "")
                break
        a = agg[(ev.name, frame)]
        a[0] += 1
        a[1] += dt
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for _, v in rows)
    print(f"ATen ops with device time, one {which} step: {tot / 1e3:.2f} ms in {sum(v[0] for _, v in rows)} calls")
    for (name, frame), (n, t) in rows[:45]:
        print(f"{t / 1e3:8.3f} ms {n:5d} x  {name:28s} {frame[:150]}")


if __name__ == "__main__":
    main()
