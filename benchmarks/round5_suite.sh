#!/bin/bash
# Round 5: tests touched since the last full run + rank emulation.  Writes gpurun_out/r5c/*
set -u
out=gpurun_out/r5c
mkdir -p $out
export PYTHONUNBUFFERED=1
( time timeout 1500 python -m pytest tests/test_revgcn112_gpu.py tests/test_graphs_gpu.py tests/test_node_fused_gpu.py tests/test_dist_gpu.py tests/test_fuse_gpu.py tests/test_revgcn.py tests/test_models_gpu.py -q -m gpu -p no:cacheprovider ) > $out/01_pytest.log 2>&1
echo "rc=$?" | tee $out/00_index.log; tail -n 14 $out/01_pytest.log | cut -c1-300
cp gpurun_out/test_gates.json gpurun_out/revgcn112_*.json $out/ 2>/dev/null
timeout 600 python -m pytest tests/test_gen_aggr_gpu.py -q -m gpu -k destination_range -p no:cacheprovider > $out/02_products_range.log 2>&1
echo "range rc=$?" | tee -a $out/00_index.log; tail -n 4 $out/02_products_range.log | cut -c1-300
( time timeout 900 python bench.py --emulate-ranks 2,4,8 --no-cpu-baseline --no-extras ) > $out/03_rank_emulation.json 2> $out/03_rank_emulation.err
echo "emulate rc=$?" | tee -a $out/00_index.log; tail -n 3 $out/03_rank_emulation.err; tail -c 1500 $out/03_rank_emulation.json
