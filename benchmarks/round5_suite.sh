#!/bin/bash
# Round 5: the full GPU suite, smoke, the driver-style default bench, RevGCN-8 kernel breakdown.  Writes gpurun_out/r5e/*
set -u
R=$PWD
out=$R/gpurun_out/r5h
mkdir -p $out
export PYTHONUNBUFFERED=1
echo "== full GPU suite (the driver's command, without -x)" | tee $out/00_index.log
( time timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider ) > $out/01_pytest_gpu_full.log 2>&1
echo "rc=$?" | tee -a $out/00_index.log; tail -n 12 $out/01_pytest_gpu_full.log | cut -c1-300
cp gpurun_out/test_gates.json $out/ 2>/dev/null; cp gpurun_out/revgcn112_*.json $out/ 2>/dev/null
echo "== smoke" | tee -a $out/00_index.log
timeout 300 python __graft_entry__.py --smoke > $out/02_smoke.log 2>&1; echo "rc=$?" | tee -a $out/00_index.log; tail -n 1 $out/02_smoke.log
echo "== default bench (driver style)" | tee -a $out/00_index.log
( time timeout 900 python bench.py ) > $out/03_bench_default.json 2> $out/03_bench_default.err
echo "rc=$?" | tee -a $out/00_index.log; tail -n 4 $out/03_bench_default.err
export TMPDIR=/tmp
cd /tmp
for s in 3 13; do
  rm -rf /tmp/p_rev$s
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_rev$s -- python $R/benchmarks/model_steps.py revgcn8 $s > /dev/null 2>&1
  cp $(find /tmp/p_rev$s -name "*kernel_stats.csv" | head -1) $out/revgcn8_${s}_kernel_stats.csv
done
cd $R
python benchmarks/launch_census.py revgcn8 2>/dev/null > $out/04_census_revgcn8.txt; head -1 $out/04_census_revgcn8.txt
python benchmarks/launch_census.py resgcn28 2>/dev/null > $out/04_census_resgcn28.txt; head -1 $out/04_census_resgcn28.txt
