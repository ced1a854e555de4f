#!/usr/bin/env python
"""Training steps of one BASELINE model only, for rocprofv3 (run twice with different step counts and difference the
kernel statistics: profiles/diff_stats.py):

    python benchmarks/model_steps.py {deepergcn28|deepergcn14|resgcn28|revgcn8|revgcn112}[_graph] STEPS

deepergcn28 / deepergcn14: the restated model file's class fused from outside (fuse.fuse_model, full recompute) on the
arxiv shape / one products cluster; resgcn28: sem_seg_dense at B = 8 x 4096; revgcn8 / revgcn112: the model file's forward
fused from outside (composed per-edge encoders), max aggregation, ogbn-proteins cluster shape; revgcn8_product /
revgcn8_power_product: the same model file WITHOUT fuse (install() alone: fused edge-GEMM kernels, max / power);
*_graph: the same step replayed as one hipGraph."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import deep_gcns_torch_amd  # noqa: E402

deep_gcns_torch_amd.install()
import arch_restated  # noqa: E402
import rev_restated  # noqa: E402
from deep_gcns_torch_amd import fuse, synth  # noqa: E402
from deep_gcns_torch_amd.graphs import GraphedStep  # noqa: E402

which, steps = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda:0")
torch.manual_seed(0)
graphed = which.endswith("_graph")
if which.startswith("deepergcn"):
    if which.startswith("deepergcn28"):
        sh = synth.SHAPES["arxiv"]
        n, L, cin, ncls = sh["n"], 28, 128, 40
        ei = synth.undirected_random_graph(n, sh["n_undirected"], sh["seed"], device=dev)
    else:
        sp = synth.SHAPES["products"]
        n, L, cin, ncls = sp["n"] // 10, 14, 100, 47
        ei = synth.undirected_random_graph(n, sp["n_undirected"] // 100, sp["seed"] + 1, device=dev)
    x, y = torch.randn(n, cin, device=dev), torch.randint(0, ncls, (n,), device=dev)
    m = fuse.fuse_model(arch_restated.DeeperGCN(num_layers=L, in_channels=cin, hidden=128, num_tasks=ncls, dropout=0.5).to(dev).train())
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, capturable=graphed, fused=True)

    def step():
        opt.zero_grad(set_to_none=True)
        torch.nn.functional.nll_loss(m(x, ei), y).backward()
        opt.step()
elif which.startswith("resgcn28"):
    m = arch_restated.DenseDeepGCN(n_blocks=28, channels=64, k=16, in_channels=9, n_classes=13).to(dev).train()
    x = torch.cat([torch.rand(8, 3, 4096, 1), torch.rand(8, 6, 4096, 1)], 1).to(dev)
    y = torch.randint(0, 13, (8, 4096), device=dev)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, capturable=graphed, fused=True)

    def step():
        opt.zero_grad(set_to_none=True)
        torch.nn.functional.cross_entropy(m(x), y).backward()
        opt.step()
else:
    layers = 112 if "112" in which else 8
    s = synth.SHAPES["proteins_cluster"]
    ei = synth.powerlaw_graph(s["n"], s["n_undirected"], s["seed"], device=dev)
    N, E = s["n"], ei.size(1)
    table = torch.rand(N, 8, device=dev)
    xin, nidx = torch.rand(N, 8, device=dev), torch.arange(N, device=dev)
    ea = torch.rand(E, 8, device=dev)
    y = (torch.rand(N, 112, device=dev) > 0.5).float()
    aggr = "power" if "power" in which else "max"
    m = rev_restated.RevGCNModelFile(num_layers=layers, hidden=224, aggr=aggr, dropout=0.2, node_table=table,
                                     impl="product").to(dev).train()
    if "product" not in which:        # *_product: install() alone, the fused edge-GEMM path of eff_gcn_modules.rev
        m = fuse.fuse_model(m)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, capturable=graphed, fused=True)

    def step():
        opt.zero_grad(set_to_none=True)
        torch.nn.functional.binary_cross_entropy_with_logits(m(xin, nidx, ei, ea), y).backward()
        opt.step()
import time  # noqa: E402

run = GraphedStep(step, warmup=2) if graphed else step
if not graphed:
    step()
torch.cuda.synchronize()
profs = None
if os.environ.get("DGCN_HOST_PROFILE"):
    # where the host time of an eager step goes: cProfile on the calling thread (forward, optimizer) and a second one on
    # autograd's device thread (every backward node of a device tensor runs there), switched on from a hook of the loss
    import cProfile
    profs = (cProfile.Profile(), cProfile.Profile())
    _backward = torch.Tensor.backward
    _armed = []

    def _arm(grad):
        if not _armed:
            _armed.append(1)
            profs[1].enable()
        return grad

    def backward(self, *a, **kw):
        self.register_hook(_arm)
        return _backward(self, *a, **kw)

    torch.Tensor.backward = backward
    profs[0].enable()
t0 = time.perf_counter()
for _ in range(steps):
    run()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
if profs is not None:
    import io
    import pstats
    profs[0].disable()
    dst = os.path.join(ROOT, "gpurun_out", f"host_profile_{which}.txt")
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    with open(dst, "w") as f:
        f.write(f"{which}: host issue time {t_issue / steps * 1e3:.3f} ms per step over {steps} steps "
                f"(cProfile on: slower than the plain run)\n")
        for name, pr in zip(("calling thread", "autograd device thread"), profs):
            out = io.StringIO()
            st = pstats.Stats(pr, stream=out)
            st.sort_stats("tottime").print_stats(40)
            st.sort_stats("cumulative").print_stats(70)
            f.write(f"\n================ {name} ================\n" + out.getvalue())
print(f"{which}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per step over {steps} steps (wall)")
