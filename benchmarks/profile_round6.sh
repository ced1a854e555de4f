#!/bin/bash
# Round-6 profile collection on the GPU box (from the repo root).  Everything lands under gpurun_out/r6prof/.
#  1. the headline on ONE box in ONE session: the unprofiled bench line (HIP-event launch time), the same command under
#     rocprofv3 --kernel-trace --stats (its kernel average AND the event time the line reports while profiled), then the
#     FETCH_SIZE / WRITE_SIZE passes;
#  2. per-model kernel breakdowns at HEAD (two step counts each);
#  3. MFMA-busy of the MFMA kernels north_star does not name a counter for; counters of the round-6 kNN kernels.
R=$PWD
O=$R/gpurun_out/r6prof
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3"
$B > $O/headline_unprofiled.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- $B > $O/headline_profiled.json 2> $O/headline_stats.err
cp $(find /tmp/p_stats -name "*kernel_stats.csv" | head -1) $O/products_gen_aggr_kernel_stats.csv
$B > $O/headline_unprofiled_again.json 2> /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- $B > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- $B > $O/write.log 2>&1
mkdir -p $O/prof_stats; cp -r /tmp/p_stats/* $O/prof_stats/ 2>/dev/null
for m in deepergcn28 resgcn28 resgcn28_graph revgcn8 revgcn8_graph revgcn112_graph; do
  for s in 3 13; do
    rm -rf /tmp/p_$m$s
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$m$s -- python $R/benchmarks/model_steps.py $m $s > /dev/null 2>&1
    cp $(find /tmp/p_$m$s -name "*kernel_stats.csv" | head -1) $O/${m}_${s}_kernel_stats.csv
  done
done
cd $R
bash benchmarks/pmc_mfma.sh $O/mfma_busy_resgcn28.md "vertex_gemm_kernel edgeconv_bwd dense_edge knn_filter2_kernel" -- python $R/benchmarks/model_steps.py resgcn28 3 > /dev/null 2>&1
cd $R
bash benchmarks/pmc_mfma.sh $O/mfma_busy_deepergcn28.md "rows_linear rows_tn" -- python $R/benchmarks/model_steps.py deepergcn28 3 > /dev/null 2>&1
cd $R
bash benchmarks/pmc_mfma.sh $O/mfma_busy_revgcn8.md "rows_linear rows_tn egemm" -- python $R/benchmarks/model_steps.py revgcn8_product 3 > /dev/null 2>&1
cd $R
for d in 1 27; do
  bash benchmarks/pmc_kernel.sh knn_filter2_kernel $O/knn_filter2_d${d}_counters.txt -- python $R/benchmarks/knn_only.py --d $d > /dev/null 2>&1
  cd $R
  bash benchmarks/pmc_kernel.sh knn_select_lists_kernel $O/knn_select_d${d}_counters.txt -- python $R/benchmarks/knn_only.py --d $d > /dev/null 2>&1
  cd $R
done
python benchmarks/knn_time.py --iters 50 > $O/knn_time.json 2>/dev/null
python benchmarks/knn_time.py --iters 50 --lds-lists > $O/knn_time_lds_lists.json 2>/dev/null
bash benchmarks/knn_trace_r06.sh > $O/knn_trace.txt 2>&1
ls $O | head -60
