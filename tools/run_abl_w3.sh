export WG_SHAPES=0,1,2,3
echo "== full"; timeout -k 10 200 python tools/bench_wgrad.py 256 2>&1 | grep -v amdgpu.ids | cut -c1-110
for n in 1 2 4 8 6 7; do echo "== W3_ABL=$n"; CNK_LIB_OVERRIDE=convnets_amd/_abl/libcnk_w3_$n.so timeout -k 10 200 python tools/bench_wgrad.py 256 2>&1 | grep -v amdgpu.ids | cut -c50-100; done
