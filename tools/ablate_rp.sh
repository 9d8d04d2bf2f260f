#!/bin/bash
# Debug builds of libcnk with parts of the first-layer (row-patch) forward compiled out (RP_ABL bits: 1 no prefetch DMA,
# 2 no MFMA loop, 4 no output stores, 8 no epilogue at all) to see what bounds it.  Outputs convnets_amd/_abl/libcnk_rp<n>.so;
# tools/bench_stem_conv.py loads one when CNK_LIB_OVERRIDE points to it (results are garbage by construction).
set -e
cd "$(dirname "$0")/.."
mkdir -p convnets_amd/_abl
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -Iinclude -DRP_ABL=$n \
     -c convnets_amd/csrc/conv_rowpatch.hip -o convnets_amd/_abl/rp_$n.o &
done
wait
for n in "$@"; do
  objs=$(ls convnets_amd/_obj/*.o | grep -v conv_rowpatch.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o convnets_amd/_abl/libcnk_rp$n.so convnets_amd/_abl/rp_$n.o $objs
done
