#!/usr/bin/env python
"""The node-sized kernels around the dense edge reduce at the config-2 layer shape (B = 8, N = 4096, C = Cout = 64), timed
with events on the launch stream through the C ABI: the P | Q producer, the input gradient and the [dW | db] partials.

    python benchmarks/edgeconv_bwd_time.py [--iters 200]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    a = ap.parse_args()
    from deep_gcns_torch_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    B, C, N, Cout = 8, 64, 4096, 64
    x = torch.randn(B, C, N, device=dev)
    w = torch.randn(Cout, 2 * C, device=dev)
    bias = torch.randn(Cout, device=dev)
    pq = torch.empty(B, N, 2 * Cout, device=dev)
    dpq = torch.randn(B, N, 2 * Cout, device=dev)
    g = torch.randn(B, C, N, device=dev)
    dx = torch.empty(B, C, N, device=dev)
    parts = torch.empty(lib.dgcn_edgeconv_bwd_weight_num_partials(B, N), Cout * 2 * C + Cout, device=dev)
    s = _lib.current_stream_handle(dev)
    calls = {
        "edgeconv_pq": lambda: lib.dgcn_edgeconv_pq_f32(x.data_ptr(), x.stride(0), x.stride(1), x.stride(2), B, C, N,
                                                        w.data_ptr(), bias.data_ptr(), Cout, pq.data_ptr(), s),
        "edgeconv_bwd_input": lambda: lib.dgcn_edgeconv_bwd_input_f32(dpq.data_ptr(), w.data_ptr(), g.data_ptr(), g.stride(0),
                                                                      g.stride(1), g.stride(2), 1.0, B, C, N, Cout,
                                                                      dx.data_ptr(), s),
        "edgeconv_bwd_weight": lambda: lib.dgcn_edgeconv_bwd_weight_f32(dpq.data_ptr(), x.data_ptr(), x.stride(0), x.stride(1),
                                                                        x.stride(2), B, C, N, Cout, parts.data_ptr(), s),
        "reduce_partials": lambda: _lib.sum_partials(parts) is None,
    }
    out = {}
    with _lib.device_ctx(dev):
        for name, f in calls.items():
            for _ in range(10):
                assert not f()
            torch.cuda.synchronize()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(a.iters):
                f()
            t1.record()
            torch.cuda.synchronize()
            out[name + "_us"] = round(t0.elapsed_time(t1) / a.iters * 1e3, 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
