#!/bin/bash
# Round-5 profile collection on the GPU box (from the repo root): headline kernel stats + HBM counters, per-model
# kernel breakdowns (two step counts each), kNN / encoder counters.  Everything lands under gpurun_out/r5prof/.
R=$PWD
O=$R/gpurun_out/r5prof
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- $B > $O/headline_stats.log 2>&1
cp $(find /tmp/p_stats -name "*kernel_stats.csv" | head -1) $O/products_gen_aggr_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- $B > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- $B > $O/write.log 2>&1
mkdir -p $O/prof_stats; cp -r /tmp/p_stats/* $O/prof_stats/ 2>/dev/null
for m in deepergcn28 resgcn28 revgcn8 revgcn8_graph revgcn112_graph; do
  for s in 3 13; do
    rm -rf /tmp/p_$m$s
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$m$s -- python $R/benchmarks/model_steps.py $m $s > /dev/null 2>&1
    cp $(find /tmp/p_$m$s -name "*kernel_stats.csv" | head -1) $O/${m}_${s}_kernel_stats.csv
  done
done
cd $R
for d in 1 27; do bash benchmarks/pmc_kernel.sh knn_filter_bf16_kernel $O/knn_d${d}_counters.txt -- python $R/benchmarks/knn_only.py --d $d > /dev/null 2>&1; done
cd $R
bash benchmarks/pmc_kernel.sh gen_aggr_enc_fwd_kernel $O/enc_fwd_max_counters.txt -- python $R/benchmarks/enc_only.py > /dev/null 2>&1
cd $R
bash benchmarks/pmc_kernel.sh gen_aggr_enc_bwd_kernel $O/enc_bwd_max_counters.txt -- python $R/benchmarks/enc_only.py --bwd > /dev/null 2>&1
cd $R
ls $O | head -40
