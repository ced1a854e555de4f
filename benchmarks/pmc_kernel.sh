#!/bin/bash
# SQ counter passes for one kernel (per-launch averages), rocprofv3 --pmc in separate runs (never with other trace domains).
#   benchmarks/pmc_kernel.sh <kernel-name-substring> <out-file> -- <command ...>
# Run from anywhere on the GPU box; writes a small text table.
set -u
PAT="$1"; OUT="$2"; shift 3
export TMPDIR=/tmp
cd /tmp
: > "$OUT"
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"; do
  rm -rf /tmp/pmc_k
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_k -- "$@" > /tmp/pmc_k.log 2>&1
  f=$(find /tmp/pmc_k -name "*counter_collection.csv" | head -1)
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
    print('ERR', e)
    print(open('/tmp/pmc_k.log').read()[-400:])
PY
done
cat "$OUT"
