"""Whole training steps as ONE hipGraph.

The deep models of the reference launch several hundred small kernels per step (RevGCN-8 on an ogbn-proteins cluster:
~850 launches of 5 - 150 us); eager PyTorch issues them at ~25 us each from the host, so the step is HOST-bound once the
kernels are fast (measured: 21.4 ms eager against 11.1 ms of device work).  Every entry point of libdgcn is
asynchronous on the caller's stream, allocates nothing and reads nothing back to the host (include/dgcn.h), so a step
built from this package's modules can be captured by ``torch.cuda.graph`` and replayed as one submission:

    opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True)      # optimizer state updated on the device

    def step():
        opt.zero_grad(set_to_none=True)
        loss_fn(model(x, node_index, edge_index, edge_attr), y).backward()
        opt.step()

    graphed = GraphedStep(step)          # three eager steps on a side stream, then the capture
    for epoch in ...:
        x.copy_(new_x); y.copy_(new_y)   # the step reads the SAME tensors every replay: refill them in place
        graphed()

What must hold inside ``step`` (torch.cuda.graph's rules): no host synchronisation (``.item()``, ``print(tensor)``,
``torch.cuda.synchronize``), static shapes, the graph structures built beforehand (``graph_of`` caches them on the first
eager call: the warm-up steps take care of it), stochastic dilation off for the dense models (it draws from the CPU
generator, ``gcn_lib/dense/torch_edge.py:19-29``).  Device RNG (dropout masks) is graph-safe.  Models that are already
device-bound gain nothing (ResGCN-28: 24.1 -> 24.6 ms, DeeperGCN-28: 30.6 -> 31.2 ms).
"""
from __future__ import annotations

import torch

__all__ = ["GraphedStep"]


class GraphedStep:
    """``step_fn`` (no arguments, no return value used) captured into a hipGraph after ``warmup`` eager runs."""

    def __init__(self, step_fn, warmup: int = 3, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("GraphedStep needs a GPU (hipGraph capture)")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        with torch.cuda.device(self.device):
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):          # warm-up off the default stream, as torch.cuda.graph requires
                for _ in range(max(1, warmup)):
                    step_fn()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                step_fn()
            torch.cuda.synchronize()

    def __call__(self):
        self.graph.replay()

    replay = __call__
