#!/bin/bash
# L2 / memory-path counters for one kernel (per-launch averages), rocprofv3 --pmc in separate runs.
#   benchmarks/pmc_cache.sh <kernel-name-substring> <out-file> -- <command ...>
# TCC = the per-XCD L2; requests that miss it go to the Infinity Cache / HBM over the fabric.
set -u
PAT="$1"; OUT="$2"; shift 3
export TMPDIR=/tmp
cd /tmp
: > "$OUT"
for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "FETCH_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD"; do
  rm -rf /tmp/pmc_c
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_c -- "$@" > /tmp/pmc_c.log 2>&1
  f=$(find /tmp/pmc_c -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$PAT" >> "$OUT" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if sys.argv[2] in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        print(k, round(sum(v) / len(v)), len(v))
except Exception as e:
    print('ERR', sys.argv[1][-40:], str(e)[:100])
PY
done
cat "$OUT"
