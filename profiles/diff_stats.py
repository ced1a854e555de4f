#!/usr/bin/env python
"""Per-step kernel breakdown from TWO rocprofv3 --kernel-trace --stats runs of benchmarks/model_steps.py with different
step counts: (stats_B - stats_A) / (steps_B - steps_A) removes start-up, warm-up and one-off launches.

    python profiles/diff_stats.py A_kernel_stats.csv STEPS_A B_kernel_stats.csv STEPS_B "title" >> profiles/r04_models_kernel_breakdown.md
"""
import csv
import sys


def short(n):
    n = n.replace("void ", "").replace("at::native::", "").replace("dgcn::(anonymous namespace)::", "dgcn::")
    n = n.replace("(anonymous namespace)::", "")
    return n.split("(")[0][:110]


def load(path):
    out = {}
    for r in csv.DictReader(open(path)):
        k = short(r["Name"])
        c, t = out.get(k, (0, 0))
        out[k] = (c + int(r["Calls"]), t + int(r["TotalDurationNs"]))
    return out


a, sa, b, sb, title = load(sys.argv[1]), int(sys.argv[2]), load(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
ds = sb - sa
rows = []
for k, (cb, tb) in b.items():
    ca, ta = a.get(k, (0, 0))
    if cb - ca > 0 and tb - ta > 0:
        rows.append((k, (tb - ta) / ds / 1e6, (cb - ca) / ds, (tb - ta) / (cb - ca) / 1e3))
rows.sort(key=lambda r: -r[1])
busy = sum(r[1] for r in rows)
launches = sum(r[2] for r in rows)
print(f"## {title}\n")
print(f"GPU busy {busy:.2f} ms per step in {launches:.0f} launches (difference of two rocprofv3 --kernel-trace --stats runs, "
      f"{sa} and {sb} steps)\n")
print("| kernel | ms / step | launches / step | avg us | % of busy |\n|---|---|---|---|---|")
other_t = other_c = 0.0
for k, t, c, avg in rows:
    if t >= 0.004 * busy and len([1 for _ in ()]) == 0:
        print(f"| `{k}` | {t:.3f} | {c:.1f} | {avg:.1f} | {100 * t / busy:.1f} |")
    else:
        other_t += t
        other_c += c
if other_c:
    print(f"| (everything below 0.4 % of busy) | {other_t:.3f} | {other_c:.1f} | {other_t / other_c * 1e3:.1f} | {100 * other_t / busy:.1f} |")
print()
