#!/bin/bash
# rocprofv3 passes behind profiles/<tag>_*: kernel stats, then FETCH_SIZE and WRITE_SIZE in SEPARATE counter passes.
# usage (on the GPU box, from the repo root):  benchmarks/profile_products.sh [extra bench.py args]
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- $B > $R/gpurun_out/prof_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -- $B > $R/gpurun_out/prof_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -- $B > $R/gpurun_out/prof_write.log 2>&1
cd $R
find gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write -name "*.csv" | head -20
