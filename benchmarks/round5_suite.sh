#!/bin/bash
set -u
out=gpurun_out/r5d
mkdir -p $out
export PYTHONUNBUFFERED=1
( time timeout 1500 python -m pytest tests/test_node_fused_gpu.py tests/test_fuse_gpu.py tests/test_revgcn.py tests/test_revgcn112_gpu.py tests/test_graphs_gpu.py tests/test_config_sizes_gpu.py tests/test_gen_aggr_gpu.py tests/test_egemm_gpu.py -q -m gpu -p no:cacheprovider -k "not products_shape and not destination_range" ) > $out/01_pytest.log 2>&1
echo "rc=$?" | tee $out/00_index.log; tail -n 12 $out/01_pytest.log | cut -c1-300
cp gpurun_out/test_gates.json $out/ 2>/dev/null
python benchmarks/launch_census.py revgcn8 2>/dev/null > $out/02_census_revgcn8.txt; head -1 $out/02_census_revgcn8.txt; grep -A14 "by issuing op" $out/02_census_revgcn8.txt
python tests/guard_alloc/revgcn_sequence.py --winner 1 --rows modelfile_fused,modelfile_fused_graph --steps 6 --replays 10 2>&1 | grep "ms per"
