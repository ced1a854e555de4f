#!/bin/bash
# counters of the two bf16 filter kernels (round 6): run on the GPU box from the repo root
R=$PWD
mkdir -p gpurun_out
bash benchmarks/pmc_kernel.sh knn_filter2_kernel $R/gpurun_out/pmc_v2_d${1:-1}.txt -- python $R/benchmarks/knn_only.py --d ${1:-1} > /dev/null
bash benchmarks/pmc_cache.sh knn_filter2_kernel $R/gpurun_out/pmcc_v2_d${1:-1}.txt -- python $R/benchmarks/knn_only.py --d ${1:-1} > /dev/null
if [ -f scratch/libdgcn_knn_noappend.so ]; then
DGCN_LIB_PATH=$R/scratch/libdgcn_knn_noappend.so bash benchmarks/pmc_kernel.sh knn_filter2_kernel $R/gpurun_out/pmc_v2na_d${1:-1}.txt -- python $R/benchmarks/knn_only.py --d ${1:-1} > /dev/null
DGCN_LIB_PATH=$R/scratch/libdgcn_knn_noappend.so bash benchmarks/pmc_cache.sh knn_filter2_kernel $R/gpurun_out/pmcc_v2na_d${1:-1}.txt -- python $R/benchmarks/knn_only.py --d ${1:-1} > /dev/null
fi
bash benchmarks/pmc_kernel.sh knn_filter_bf16_kernel $R/gpurun_out/pmc_v1_d${1:-1}.txt -- python $R/benchmarks/knn_only.py --d ${1:-1} --lds-lists > /dev/null
bash benchmarks/pmc_cache.sh knn_filter_bf16_kernel $R/gpurun_out/pmcc_v1_d${1:-1}.txt -- python $R/benchmarks/knn_only.py --d ${1:-1} --lds-lists > /dev/null
for f in pmc_v2 pmcc_v2 pmc_v2na pmcc_v2na pmc_v1 pmcc_v1; do echo "== $f"; cat gpurun_out/${f}_d${1:-1}.txt 2>/dev/null | tr '\n' ';'; echo; done
