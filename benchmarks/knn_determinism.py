#!/usr/bin/env python
"""kNN graph build called many times on the same cloud: every call must return the first call's ids (config-2 shape,
random features; K = 16 / 224 / 432, C = 64 and 32, with and without exclude_self).

    python benchmarks/knn_determinism.py [--reps 60]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=60)
    a = ap.parse_args()
    from deep_gcns_torch_amd import dense_ops
    dev = torch.device("cuda:0")
    bad_total = 0
    for C in (64, 32):
        x = torch.randn(8, C, 4096, 1, device=dev, generator=torch.Generator(device=dev).manual_seed(C))
        for K, d in ((16, 1), (224, 14), (432, 27)):
            for ex in (False, True):
                first = dense_ops.knn_edge_index(x, K // d, d, exclude_self=ex)
                bad = 0
                for _ in range(a.reps):
                    bad += int((dense_ops.knn_edge_index(x, K // d, d, exclude_self=ex) != first).any(-1).sum())
                bad_total += bad
                print(f"C={C} K={K} exclude_self={ex}: rows differing from the first call over {a.reps} calls: {bad}", flush=True)
    print("TOTAL", bad_total)


if __name__ == "__main__":
    main()
