#!/bin/bash
# Per-step kernel statistics of benchmarks/model_steps.py models (3 and 13 steps each, differenced by
# profiles/diff_stats.py):   bash benchmarks/profile_models.sh OUTDIR MODEL [MODEL ...]
R=$PWD
O=$(realpath -m "$1"); shift      # (absolute: the runs below happen in /tmp)
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for m in "$@"; do
  for s in 3 13; do
    rm -rf /tmp/p_$m$s
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$m$s -- python $R/benchmarks/model_steps.py $m $s > $O/${m}_${s}.log 2>&1
    cp $(find /tmp/p_$m$s -name "*kernel_stats.csv" | head -1) $O/${m}_${s}_kernel_stats.csv
  done
done
cd $R
