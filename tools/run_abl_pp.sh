# runs tools/bench_pp.py on the full build and on the ablated builds of tools/ablate_pp.sh (arguments: PP_ABL values)
set -o pipefail
mkdir -p gpurun_out
export PP_SHAPES=${PP_SHAPES:-1,2,3}
echo "== full" ; timeout -k 10 150 python tools/bench_pp.py 256 --check 2>&1 | grep -v amdgpu.ids || exit 1
for n in "$@"; do
  echo "== PP_ABL=$n"
  CNK_LIB_OVERRIDE=convnets_amd/_abl/libcnk_pp$n.so timeout -k 10 150 python tools/bench_pp.py 256 2>&1 | grep -v amdgpu.ids | cut -c1-75 || exit 1
done
