#!/bin/bash
# HBM traffic per launch of the kernels whose name contains PATTERN: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc
# passes (never with other trace domains), KiB units, FETCH_SIZE doubled (the gfx950 correction of
# /opt/skills/guides/MI355X_MICROARCH.md, as profiles/summarize_rocprof.py applies it).
#   benchmarks/pmc_traffic.sh <kernel-name-substring> <out-file> -- <command ...>
set -u
PAT="$1"; OUT="$2"; shift 3
export TMPDIR=/tmp
cd /tmp
: > "$OUT"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_t
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_t -- "$@" > /tmp/pmc_t.log 2>&1
  f=$(find /tmp/pmc_t -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$PAT" >> "$OUT" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r['Kernel_Name']:
        agg[(r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][-60:], r['Counter_Name'])].append(float(r['Counter_Value']))
for (k, c), v in sorted(agg.items()):
    kib = sum(v) / len(v)
    gb = kib * 1024 * (2 if c == 'FETCH_SIZE' else 1) / 1e9
    print(f"{k}\t{c}\t{kib:.0f} KiB raw\t{gb:.4f} GB per launch ({len(v)} launches)")
PY
done
cat "$OUT"
