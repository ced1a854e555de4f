#!/bin/bash
# average duration of kernels matching a pattern in one model's step (rocprofv3 --kernel-trace --stats), GPU box, repo root:
#   bash benchmarks/kernel_avg.sh <model> "<pat1>|<pat2>"
R=$PWD; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/ka
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ka -- python $R/benchmarks/model_steps.py $1 8 > /dev/null 2>&1
python3 - "$(find /tmp/ka -name '*kernel_stats.csv' | head -1)" "$2" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r['Name']): print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
