#!/usr/bin/env python
"""Per-dilation cost of the kNN graph build of config 2 (B = 8, N = 4096, C = 64, k = 16, d = 1 .. 28) split by kernel:
run under `rocprofv3 --kernel-trace` and read the dispatches in order (3 calls per dilation, the last one counts).

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python benchmarks/knn_redo_by_dilation.py
    python benchmarks/knn_redo_by_dilation.py --read OUT      # table: d, filter us, redo us, planes us
"""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REPS = 3

if "--read" in sys.argv:
    root = sys.argv[sys.argv.index("--read") + 1]
    path = sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = [r for r in csv.DictReader(open(path)) if "knn" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    calls, cur = [], []
    for r in rows:                      # a call = planes, filter, redo (the exact kernel closes it)
        cur.append(r)
        if "knn_dense_kernel" in r["Kernel_Name"]:
            calls.append(cur)
            cur = []
    print("| d | K | filter kernel | filter us | exact redo us | planes us |\n|---|---|---|---|---|---|")
    for d in range(1, 29):
        c = calls[(d - 1) * REPS + REPS - 1]
        dur = lambda key: sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in c if key in r["Kernel_Name"])
        name = next((r["Kernel_Name"] for r in c if "knn_filter" in r["Kernel_Name"]), "")
        name = name[name.find("knn_filter"):name.find("(")] if name else "-"
        print(f"| {d} | {16 * d} | `{name}` | {dur('knn_filter'):.1f} | {dur('knn_dense_kernel'):.1f} | {dur('knn_planes'):.1f} |")
    sys.exit(0)

import torch  # noqa: E402

import deep_gcns_torch_amd  # noqa: E402

deep_gcns_torch_amd.install()
from gcn_lib.dense import DenseDilatedKnnGraph  # noqa: E402

torch.manual_seed(0)
x = torch.randn(8, 64, 4096, 1, device="cuda:0")
for d in range(1, 29):
    g = DenseDilatedKnnGraph(16, d)
    for _ in range(REPS):
        g(x)
torch.cuda.synchronize()
