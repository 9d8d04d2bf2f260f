#!/bin/bash
# Debug builds of libcnk with parts of the 3x3 all-taps weight-gradient kernel compiled out (W3_ABL bits: 1 no LDS-DMA in the
# loop, 2 no fragment reads, 4 no MFMA, 8 no slab store) -> convnets_amd/_abl/libcnk_w3_<n>.so (CNK_LIB_OVERRIDE, results are garbage)
set -e
cd "$(dirname "$0")/.."
mkdir -p convnets_amd/_abl
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -DW3_ABL=$n \
     -c convnets_amd/csrc/conv_wgrad3.hip -o convnets_amd/_abl/w3_$n.o &
done
wait
for n in "$@"; do
  objs=$(ls convnets_amd/_obj/*.o | grep -v conv_wgrad3.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o convnets_amd/_abl/libcnk_w3_$n.so convnets_amd/_abl/w3_$n.o $objs
done
