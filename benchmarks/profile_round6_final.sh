#!/bin/bash
# Round 6, last session on the final library (from the repo root, GPU box): per-model kernel breakdowns (two step counts
# each, differenced by profiles/diff_stats.py), the model step times, the kNN times and one default bench line.
# Everything lands under gpurun_out/r6final/.
R=$PWD
O=$R/gpurun_out/r6final
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for m in deepergcn28 resgcn28 revgcn8 revgcn8_graph revgcn112_graph; do
  for s in 3 13; do
    rm -rf /tmp/p_$m$s
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$m$s -- python $R/benchmarks/model_steps.py $m $s > /dev/null 2>&1
    cp $(find /tmp/p_$m$s -name "*kernel_stats.csv" | head -1) $O/${m}_${s}_kernel_stats.csv
  done
done
cd $R
for m in deepergcn28 resgcn28 resgcn28_graph revgcn8 revgcn8_graph revgcn8_product revgcn8_power_product; do
  python benchmarks/model_steps.py $m 30 2>&1 | tail -1
done > $O/model_steps.txt
for m in revgcn112 revgcn112_graph; do
  python benchmarks/model_steps.py $m 10 2>&1 | tail -1
done >> $O/model_steps.txt
python benchmarks/knn_time.py --iters 50 > $O/knn_time.json 2>/dev/null
python bench.py > $O/bench_default.json 2> $O/bench_default.err
cat $O/model_steps.txt
