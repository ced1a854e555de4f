#!/usr/bin/env python
"""Two probes of the hipGraph memset-node ordering:

1. micro: one captured graph = [a long kernel that READS a buffer] -> [hipMemsetAsync of the buffer's head] -> [a kernel
   that restores it]; eager and replayed results of the read must agree if the memset node waits for the kernel in front.
2. which torch ops of a fused RevGCN training step issue device memsets (torch.profiler, python stacks)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

dev = torch.device("cuda:0")
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]

SMALL = os.environ.get("PROBE_SMALL") == "1"
n = 1 << 28                                   # 1 GiB of floats: the read takes ~0.3 ms
A = torch.ones(n, device=dev)
out = torch.zeros(64, device=dev, dtype=torch.float64)
head = 1 << 20                                # floats zeroed by the memset (4 MiB)


def body(i):
    s = A.double().sum()                      # long read of A (two kernels: cast + reduce; the cast reads all of A)
    out[i] = s
    if SMALL:                                 # 1 KiB at the END of the buffer (read by the last workgroups)
        hip.hipMemsetAsync(A.data_ptr() + (n - 256) * 4, 0, 1024, torch.cuda.current_stream().cuda_stream)
        A[n - 256:].fill_(1.0)
    else:
        hip.hipMemsetAsync(A.data_ptr(), 0, head * 4, torch.cuda.current_stream().cuda_stream)
        A[:head].fill_(1.0)                   # restore (a kernel)


for i in range(4):
    body(i)
torch.cuda.synchronize()
print("eager sums:", out[:4].tolist())
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    body(0)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for i in range(16):
        body(i)
bad = 0
for rep in range(20):
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    bad += int((out[:16] != float(n)).sum())
print(f"replayed: {bad} of {20 * 16} sums differ from {float(n)} (a memset node that waits for the kernel in front gives 0)")
print("last replay:", out[:16].tolist())

# ---- 2. memsets of a training step --------------------------------------------------------------------------------------
import deep_gcns_torch_amd  # noqa: E402

deep_gcns_torch_amd.install()
import rev_restated  # noqa: E402
from deep_gcns_torch_amd import fuse, synth  # noqa: E402

s = synth.SHAPES["proteins_cluster"]
ei = synth.powerlaw_graph(s["n"], s["n_undirected"], s["seed"], device=dev)
N, E = s["n"], ei.size(1)
table = torch.rand(N, 8, device=dev)
xin, nidx = torch.rand(N, 8, device=dev), torch.arange(N, device=dev)
ea = torch.rand(E, 8, device=dev)
y = (torch.rand(N, 112, device=dev) > 0.5).float()
m = fuse.fuse_model(rev_restated.RevGCNModelFile(num_layers=2, hidden=224, aggr="max", dropout=0.2, node_table=table,
                                                 impl="product").to(dev).train())
opt = torch.optim.Adam(m.parameters(), lr=1e-3, capturable=True)


def step():
    opt.zero_grad(set_to_none=True)
    torch.nn.functional.binary_cross_entropy_with_logits(m(xin, nidx, ei, ea), y).backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ms = [e for e in prof.events() if "emset" in e.name or "fillBuffer" in e.name]
chains = {}
for e in prof.events():
    if e.name == "hipMemsetAsync":
        chain, q = [], e.cpu_parent
        while q is not None:
            chain.append(q.name)
            top = q
            q = q.cpu_parent
        st = [f.strip()[-110:] for f in (top.stack or []) if "torch/" not in f][:4] if chain else []
        key = (" < ".join(chain[:6]), tuple(st))
        chains[key] = chains.get(key, 0) + 1
for (c, st), k in sorted(chains.items(), key=lambda kv: -kv[1]):
    print(f"{k} x hipMemsetAsync under [{c}] at {list(st)}")
print(f"memset-like events in one 2-layer step: {len(ms)}")
seen = {}
for e in prof.events():
    if e.name in ("aten::zero_", "aten::zeros", "aten::zeros_like", "aten::new_zeros", "aten::fill_"):
        st = [f for f in (e.stack or []) if "/repo/" in f or "rev_restated" in f][:3]
        key = (e.name, tuple(st))
        seen[key] = seen.get(key, 0) + 1
for (name, st), c in sorted(seen.items(), key=lambda kv: -kv[1])[:25]:
    print(c, name, " <- ".join(x.strip()[-90:] for x in st))
names = {}
for e in ms:
    names[e.name] = names.get(e.name, 0) + 1
print(names)
