#!/bin/bash
# Debug builds of libcnk with parts of the ring GEMM (conv_gemm1x1.hip) compiled out (G1_ABL bits: 1 no MFMAs, 2 no DMAs,
# 4 no fragment reads); outputs convnets_amd/_abl/libcnk_g1_<n>.so for tools/bench_one.py (CNK_LIB_OVERRIDE).
set -e
cd "$(dirname "$0")/.."
mkdir -p convnets_amd/_abl
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -Iinclude -DG1_ABL=$n \
     -c convnets_amd/csrc/conv_gemm1x1.hip -o convnets_amd/_abl/g1_$n.o &
done
wait
for n in "$@"; do
  objs=$(ls convnets_amd/_obj/*.o | grep -v conv_gemm1x1.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o convnets_amd/_abl/libcnk_g1_$n.so convnets_amd/_abl/g1_$n.o $objs
done
