#!/bin/bash
# counters of the round-6 bf16 filter kernel (and of variant builds): run on the GPU box from the repo root
#   bash benchmarks/knn_pmc_r06.sh <d> [variant-tag ...]      ("" = the shipped library)
R=$PWD
d=${1:-1}; shift
mkdir -p gpurun_out
for tag in "shipped" "$@"; do
  if [ "$tag" = shipped ]; then unset DGCN_LIB_PATH; else export DGCN_LIB_PATH=$R/scratch/libdgcn_knn_$tag.so; fi
  bash benchmarks/pmc_kernel.sh knn_filter2_kernel $R/gpurun_out/pmc_v2_${tag}_d$d.txt -- python $R/benchmarks/knn_only.py --d $d > /dev/null
  echo "== v2 $tag d=$d"; cat gpurun_out/pmc_v2_${tag}_d$d.txt | tr '\n' ';'; echo
done
