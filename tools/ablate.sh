#!/bin/bash
# Debug builds of libcnk with parts of the halo 3x3 kernel compiled out (CNK_ABLATE bits:
# 1 = no MFMA phase, 2 = no global loads, 4 = LDS reads replaced by register values), to find which
# of {global->LDS stream, LDS reads, MFMA} bounds the kernel.  Outputs convnets_amd/_abl/libcnk_<n>.so;
# tools/bench_one.py loads one of them when CNK_LIB_OVERRIDE points to it.
set -e
cd "$(dirname "$0")/.."
mkdir -p convnets_amd/_abl
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -DCNK_ABLATE=$n \
     -c convnets_amd/csrc/conv_igemm.hip -o convnets_amd/_abl/igemm_$n.o &
done
wait
for n in "$@"; do
  objs=$(ls convnets_amd/_obj/*.o | grep -v conv_igemm.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o convnets_amd/_abl/libcnk_$n.so convnets_amd/_abl/igemm_$n.o $objs
done
