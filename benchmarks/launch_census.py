#!/usr/bin/env python
"""Kernel launches of one eager training step by kernel name and by the op that issued them (torch.profiler):

    python benchmarks/launch_census.py {revgcn8|resgcn28|deepergcn28}"""
import os
import sys
from collections import Counter

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import deep_gcns_torch_amd  # noqa: E402

deep_gcns_torch_amd.install()
import arch_restated  # noqa: E402
import rev_restated  # noqa: E402
from deep_gcns_torch_amd import fuse, synth  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "revgcn8"
dev = torch.device("cuda:0")
torch.manual_seed(0)
if which.startswith("revgcn"):
    layers = int(which[6:])
    s = synth.SHAPES["proteins_cluster"]
    ei = synth.powerlaw_graph(s["n"], s["n_undirected"], s["seed"], device=dev)
    N, E = s["n"], ei.size(1)
    table = torch.rand(N, 8, device=dev)
    xin, nidx, ea = torch.rand(N, 8, device=dev), torch.arange(N, device=dev), torch.rand(E, 8, device=dev)
    y = (torch.rand(N, 112, device=dev) > 0.5).float()
    m = fuse.fuse_model(rev_restated.RevGCNModelFile(num_layers=layers, hidden=224, aggr="max", dropout=0.2, node_table=table,
                                                     impl="product").to(dev).train())
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)

    def step():
        opt.zero_grad(set_to_none=True)
        torch.nn.functional.binary_cross_entropy_with_logits(m(xin, nidx, ei, ea), y).backward()
        opt.step()
elif which == "resgcn28":
    m = arch_restated.DenseDeepGCN(n_blocks=28, channels=64, k=16, in_channels=9, n_classes=13).to(dev).train()
    x = torch.cat([torch.rand(8, 3, 4096, 1), torch.rand(8, 6, 4096, 1)], 1).to(dev)
    y = torch.randint(0, 13, (8, 4096), device=dev)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)

    def step():
        opt.zero_grad(set_to_none=True)
        torch.nn.functional.cross_entropy(m(x), y).backward()
        opt.step()
else:
    sh = synth.SHAPES["arxiv"]
    ei = synth.undirected_random_graph(sh["n"], sh["n_undirected"], sh["seed"], device=dev)
    x, y = torch.randn(sh["n"], 128, device=dev), torch.randint(0, 40, (sh["n"],), device=dev)
    m = fuse.fuse_model(arch_restated.DeeperGCN(num_layers=28, in_channels=128, hidden=128, num_tasks=40, dropout=0.5).to(dev).train())
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)

    def step():
        opt.zero_grad(set_to_none=True)
        torch.nn.functional.nll_loss(m(x, ei), y).backward()
        opt.step()
for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
kern = Counter()
ktime = Counter()
owner = Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        name = e.name.split("(")[0][:70]
        kern[name] += 1
        ktime[name] += e.device_time
launch = [e for e in prof.events() if e.name in ("hipLaunchKernel", "hipExtModuleLaunchKernel", "hipModuleLaunchKernel",
                                                   "hipMemsetAsync", "hipMemcpyAsync", "hipExtLaunchKernel")]
for e in launch:
    q, top = e.cpu_parent, None
    chain = []
    while q is not None:
        chain.append(q.name)
        q = q.cpu_parent
    # the innermost aten / custom op and the outermost autograd node
    inner = next((c for c in chain if c.startswith("aten::") or "Backward" in c or c[:1] == "_"), "(library call through ctypes)")
    outer = next((c for c in reversed(chain) if "evaluate_function" in c), "")
    owner[(inner, outer.replace("autograd::engine::evaluate_function: ", "bwd of "))] += 1
print(f"{which}: {sum(kern.values())} device activities, {len(launch)} launch calls in one eager step")
print("-- by kernel (count, total us)")
for k, c in kern.most_common(28):
    print(f"{c:5d} {ktime[k]:9.0f}  {k}")
print("-- by issuing op (count)")
for (i, o), c in owner.most_common(30):
    print(f"{c:5d}  {i}   [{o}]")
# host side: self CPU time per op name (both the calling thread and autograd's device thread), and the wall time of the step
cpu = Counter()
cnt = Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CPU:
        cpu[e.name[:70]] += e.self_cpu_time_total
        cnt[e.name[:70]] += 1
print(f"-- host: self CPU time by op (us, calls); total {sum(cpu.values()):.0f} us under the profiler")
for k, t in cpu.most_common(40):
    print(f"{t:9.0f} {cnt[k]:5d}  {k}")
