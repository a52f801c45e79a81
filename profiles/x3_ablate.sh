#!/bin/bash
# Build ablated variants of the bf16-split GEMM kernel next to the product library (run HERE, hipcc
# cross-compiles), then on the GPU box:  bash profiles/x3_ablate.sh run
cd "$(dirname "$0")/../dynamic-gcnn_amd/csrc" || exit 1
if [ "$1" = run ]; then
  cd ../..
  for v in ${VARIANTS:-0 1 2 3 4 5 6 7}; do
    echo "== X3_ABLATE=$v"
    DGCNN_HIP_LIB=dynamic-gcnn_amd/csrc/variants/libdgcnn_x3_$v.so python profiles/gemm_bench.py 6 2>&1 | grep -E "FC0|merged fwd"
  done
  exit 0
fi
mkdir -p variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
for v in ${VARIANTS:-0 1 2 3 4 5 6 7}; do
  /opt/rocm/bin/hipcc $FLAGS -DX3_ABLATE=$v -c gemm_x3.hip -o variants/x3_$v.o &
done
wait
for v in ${VARIANTS:-0 1 2 3 4 5 6 7}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libdgcnn_x3_$v.so build/api.o build/knn.o build/gemm.o variants/x3_$v.o build/bn.o build/misc.o
done
ls -la variants/*.so
