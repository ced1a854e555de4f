#!/bin/bash
# Phase decomposition of the bf16 kNN filter kernel: variant builds of knn_dense.hip (profiling switches, WRONG results)
# linked against the shipped objects into scratch/libdgcn_knn_<tag>.so; run each with DGCN_LIB_PATH on the GPU box:
#   bash benchmarks/knn_phase_variants.sh build            (here: hipcc cross-compiles)
#   bash benchmarks/knn_phase_variants.sh run              (GPU box: prints one knn_time.py JSON line per variant)
set -e
cd "$(dirname "$0")/.."
CS=deep_gcns_torch_amd/csrc
VARIANTS=("s1:-DKNNF_STOP_AFTER=1" "s2:-DKNNF_STOP_AFTER=2" "noappend:-DKNNF_STOP_AFTER=2 -DKNNF_NO_APPEND" "noloads:-DKNNF_STOP_AFTER=2 -DKNNF_NO_APPEND -DKNNF_NO_LOADS" "nomfma:-DKNNF_STOP_AFTER=2 -DKNNF_NO_APPEND -DKNNF_NO_MFMA2")
if [ "$1" = build ]; then
  mkdir -p scratch
  for v in "${VARIANTS[@]}"; do
    tag=${v%%:*}; flags=${v#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -Iinclude -I$CS -c $CS/knn_dense.hip -o scratch/knn_$tag.o &
  done
  wait
  for v in "${VARIANTS[@]}"; do
    tag=${v%%:*}
    objs=$(ls $CS/_obj/*.o | grep -v knn_dense.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/libdgcn_knn_$tag.so scratch/knn_$tag.o $objs
  done
  ls -la scratch/*.so
else
  echo "shipped: $(python benchmarks/knn_time.py --iters 30 | tail -1)"
  for v in "${VARIANTS[@]}"; do
    tag=${v%%:*}
    echo "$tag: $(DGCN_LIB_PATH=$PWD/scratch/libdgcn_knn_$tag.so python benchmarks/knn_time.py --iters 30 | tail -1)"
  done
fi
