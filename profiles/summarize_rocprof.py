#!/usr/bin/env python
"""Condense rocprofv3 CSV output (gpurun_out/prof_*) into the small files committed here.

    python profiles/summarize_rocprof.py --tag r01_products --stats gpurun_out/prof_stats \
        --fetch gpurun_out/prof_fetch --write gpurun_out/prof_write \
        --shape products --graph uniform --channels 128 [--latest]

Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are
collected in SEPARATE passes (TCC has 4 slots: 3 + 2 do not fit), both are in KiB, and on gfx950
FETCH_SIZE reports exactly 1/2 of the bytes of a wide (16 B/lane) coalesced read -> doubled.
WRITE_SIZE is used as reported (it matches the known output byte count of these kernels exactly).
"""
import argparse
import csv
import json
import os
import statistics

KERNELS = ("gen_aggr_fwd_kernel", "gen_aggr_bwd_kernel", "max_mask_build", "egemm_fwd", "knn_dense", "edgeconv", "mr_aggr", "vertex_gemm",
           "mrconv")


def counter_avg(d, counter):
    out = {}
    path = os.path.join(d, [f for f in os.listdir(d) if f.endswith("counter_collection.csv")][0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        if not any(k in name for k in KERNELS):
            continue
        short = name.replace("void dgcn::(anonymous namespace)::", "").split("(")[0]
        out.setdefault(short, []).append(float(r["Counter_Value"]))
    return {k: (statistics.mean(v), len(v)) for k, v in out.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--stats")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--shape", default="products")
    ap.add_argument("--graph", default="uniform")
    ap.add_argument("--channels", type=int, default=128)
    ap.add_argument("--latest", action="store_true", help="also write profiles/traffic_latest.json (read by bench.py)")
    a = ap.parse_args()
    here = os.path.dirname(os.path.abspath(__file__))
    lines = [f"# rocprofv3 summary `{a.tag}` ({a.shape}-shaped {a.graph} graph, C={a.channels})", ""]
    if a.stats:
        path = os.path.join(a.stats, [f for f in os.listdir(a.stats) if f.endswith("kernel_stats.csv")][0])
        rows = list(csv.DictReader(open(path)))
        lines += ["## `rocprofv3 --kernel-trace --stats` (top kernels by total time)", "",
                  "| kernel | calls | avg ms | min ms | max ms | % |", "|---|---|---|---|---|---|"]
        keep = []
        for r in rows[:12]:
            name = r["Name"].replace("dgcn::(anonymous namespace)::", "dgcn::").replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            if len(name) > 90:
                name = name[:87] + "..."
            lines.append(f"| `{name}` | {r['Calls']} | {float(r['AverageNs'])/1e6:.4f} | {float(r['MinNs'])/1e6:.4f} | "
                         f"{float(r['MaxNs'])/1e6:.4f} | {r['Percentage']} |")
            keep.append(r)
        with open(os.path.join(here, f"{a.tag}_kernel_stats.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(keep)
        lines.append("")
    traffic = {}
    if a.fetch and a.write:
        fe = counter_avg(a.fetch, "FETCH_SIZE")
        wr = counter_avg(a.write, "WRITE_SIZE")
        lines += ["## HBM traffic per launch (separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes)", "",
                  "| kernel | FETCH_SIZE KiB (raw) | x2 gfx950 correction, GB | WRITE_SIZE KiB | GB | total HBM GB/launch |",
                  "|---|---|---|---|---|---|"]
        for k in fe:
            f_raw = fe[k][0]
            w_raw = wr.get(k, (0.0, 0))[0]
            fb = 2.0 * f_raw * 1024
            wb = w_raw * 1024
            traffic[k] = dict(fetch_kib_raw=f_raw, write_kib_raw=w_raw, hbm_bytes_per_launch=fb + wb)
            lines.append(f"| `{k}` | {f_raw:.0f} | {fb/1e9:.2f} | {w_raw:.0f} | {wb/1e9:.2f} | {(fb+wb)/1e9:.2f} |")
        lines.append("")
    open(os.path.join(here, f"{a.tag}_summary.md"), "w").write("\n".join(lines) + "\n")
    if traffic:
        json.dump(traffic, open(os.path.join(here, f"{a.tag}_traffic.json"), "w"), indent=1)
        if a.latest:
            fwd = [v for k, v in traffic.items() if "gen_aggr_fwd_kernel" in k]
            if fwd:
                json.dump(dict(shape=a.shape, graph=a.graph, channels=a.channels, tag=a.tag,
                               hbm_bytes_per_launch=fwd[0]["hbm_bytes_per_launch"]),
                          open(os.path.join(here, "traffic_latest.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
