#!/bin/bash
# per-kernel durations of the kNN graph build at d = 1, 14, 27 (rocprofv3 --kernel-trace --stats), GPU box, repo root
R=$PWD; export TMPDIR=/tmp; cd /tmp
for d in 1 14 27; do
  rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/benchmarks/knn_only.py --d $d $1 > /dev/null 2>&1
  f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1)
  echo "== d=$d $1"; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if 'knn' in n: print(f"{n[:60]:60s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
done
