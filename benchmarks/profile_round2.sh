#!/bin/bash
# Round-2 profile set (run on the GPU box from the repo root; raw output under gpurun_out/, summaries are copied to
# profiles/ by hand): headline kernel stats + HBM traffic, fused edge-GEMM kernel stats / traffic / SQ counters,
# RevGCN-8 train-step kernel breakdown.
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r2prof
mkdir -p $O
bash $R/benchmarks/profile_products.sh > $O/products.log 2>&1
cd /tmp
EG="python $R/benchmarks/bench_revgcn.py --layers 8 --hidden 224 --aggr max --skip-model --iters 5"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/eg_stats -- $EG > $O/eg_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/eg_fetch -- $EG > $O/eg_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/eg_write -- $EG > $O/eg_write.log 2>&1
bash $R/benchmarks/pmc_kernel.sh egemm_fwd_bf16_kernel $O/eg_sq_counters.txt -- $EG > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rev_stats -- python $R/benchmarks/bench_revgcn.py --layers 8 --hidden 224 --aggr max --rev product --iters 3 > $O/rev_stats.log 2>&1
cd $R
find $O -name "*kernel_stats.csv" -o -name "*counter_collection.csv" | head -20
