#!/usr/bin/env python
"""Steady-state per-step kernel breakdown from a rocprofv3 --kernel-trace CSV of benchmarks/bench_models.py.

    python profiles/summarize_trace.py gpurun_out/prof_models/<host>/<pid>_kernel_trace.csv --out profiles/r01_models_kernel_breakdown.md

MIOpen's first-call algorithm search and warm-up pollute `--stats`; this takes, for the dense model, the kernels
between the optimizer launches of the last 4 training steps and, for the sparse model, two eager training steps cut
the same way, and reports time per kernel per step."""
import argparse
import collections
import csv


def short(n):
    n = n.replace("void ", "").replace("at::native::", "").replace("dgcn::(anonymous namespace)::", "dgcn::")
    return n.split("(")[0][:96]


def table(seg, steps, title):
    agg = collections.defaultdict(lambda: [0, 0])
    for r in seg:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        k = short(r["Kernel_Name"])
        agg[k][0] += d
        agg[k][1] += 1
    tot = sum(v[0] for v in agg.values())
    wall = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6
    out = [f"## {title}", "",
           f"{len(seg)} kernel launches, GPU busy {tot / 1e6 / steps:.2f} ms per step, wall {wall / steps:.2f} ms per step", "",
           "| kernel | ms / step | launches / step | avg us | % of busy |", "|---|---|---|---|---|"]
    for k, (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:22]:
        out.append(f"| `{k}` | {d / 1e6 / steps:.3f} | {c / steps:.1f} | {d / c / 1e3:.1f} | {100 * d / tot:.1f} |")
    return out + [""]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    rows = sorted(csv.DictReader(open(a.trace)), key=lambda r: int(r["Start_Timestamp"]))
    first_sparse = next(i for i, r in enumerate(rows) if "gen_aggr_fwd_kernel" in r["Kernel_Name"])
    dense = rows[:first_sparse]
    adam = [i for i, r in enumerate(dense) if "multi_tensor_apply" in r["Kernel_Name"]]
    clusters = []
    for i in adam:
        if not clusters or i - clusters[-1][-1] > 200:
            clusters.append([i])
        else:
            clusters[-1].append(i)
    lines = ["# Per-step kernel breakdown of the model benchmarks (rocprofv3 --kernel-trace, steady state)", ""]
    lines += table(dense[clusters[-5][-1] + 1:clusters[-1][-1] + 1], 4,
                   "ResGCN-28 dense (B=8, N=4096, k=16): last 4 training steps")
    # sparse model: the eager training steps only (the HIP-graph capture / replay that follows repeats them), cut at the
    # optimizer launches like the dense model: the two steps before the 8th optimizer step (3 warm-up + 5 timed)
    sparse = rows[first_sparse:]
    adam = [i for i, r in enumerate(sparse) if "multi_tensor_apply" in r["Kernel_Name"]]
    clusters = []
    for i in adam:
        if not clusters or i - clusters[-1][-1] > 50:
            clusters.append([i])
        else:
            clusters[-1].append(i)
    last = min(len(clusters), 8) - 1
    lines += table(sparse[clusters[last - 2][-1] + 1:clusters[last][-1] + 1], 2,
                   "DeeperGCN-28 on the arxiv shape (re-entrant checkpointing: two forwards per step): 2 training steps")
    open(a.out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
