#!/bin/bash
# MFMA-busy of every kernel whose name contains one of the given substrings (per-launch averages), one counter pass:
#   benchmarks/pmc_mfma.sh <out-file> "<pat1> <pat2> ..." -- <command ...>
# SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs:
#   MFMA-busy fraction = MFMA_BUSY / (1024 SIMDs x GRBM / 8).
set -u
OUT="$1"; PATS="$2"; shift 3
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pmc_m
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d /tmp/pmc_m -- "$@" > /tmp/pmc_m.log 2>&1
f=$(find /tmp/pmc_m -name "*counter_collection.csv" | head -1)
python3 - "$f" "$PATS" > "$OUT" <<'PY'
import csv, sys, collections
pats = sys.argv[2].split()
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    for p in pats:
        if p in n:
            short = n.replace('void ', '').replace('dgcn::(anonymous namespace)::', '').split('(')[0][:70]
            agg[short][r['Counter_Name']].append(float(r['Counter_Value']))
print("| kernel | launches | SQ_INSTS_MFMA | SQ_VALU_MFMA_BUSY_CYCLES | GRBM_GUI_ACTIVE / 8 | MFMA-busy |")
print("|---|---|---|---|---|---|")
for k, c in sorted(agg.items()):
    m = lambda x: sum(c[x]) / max(len(c[x]), 1)
    busy, grbm = m('SQ_VALU_MFMA_BUSY_CYCLES'), m('GRBM_GUI_ACTIVE') / 8
    print(f"| `{k}` | {len(c['GRBM_GUI_ACTIVE'])} | {m('SQ_INSTS_MFMA'):.0f} | {busy:.0f} | {grbm:.0f} | {busy / (1024 * grbm) if grbm else 0:.3f} |")
PY
cat "$OUT"
