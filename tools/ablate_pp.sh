#!/bin/bash
# Debug builds of libcnk with parts of the ping-pong 3x3 core compiled out (PP_ABL bits: 1 no patch DMA, 2 no weight
# DMA, 4 no LDS fragment reads, 8 no MFMA), to see which of them bounds a phase.  Outputs convnets_amd/_abl/libcnk_pp<n>.so;
# ABL_MACRO=WS_ABL: the same for the weight-stationary 64 -> 64 kernel (bits: 1 no patch DMA in the loop, 4 no LDS reads, 8 no MFMA,
# 16 no epilogue).  tools/bench_pp.py loads one of them when CNK_LIB_OVERRIDE points to it (results are garbage by construction).
set -e
cd "$(dirname "$0")/.."
mkdir -p convnets_amd/_abl
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -D${ABL_MACRO:-PP_ABL}=$n \
     -c convnets_amd/csrc/conv_pp.hip -o convnets_amd/_abl/pp_$n.o &
done
wait
for n in "$@"; do
  objs=$(ls convnets_amd/_obj/*.o | grep -v conv_pp.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o convnets_amd/_abl/libcnk_pp$n.so convnets_amd/_abl/pp_$n.o $objs
done
