# one GPU round for the ping-pong 3x3 core: the conv / BN op tests, then tools/bench_pp.py with interleaved variants
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv or fuse or bn" > gpurun_out/pp_t.log 2>&1; rc=$?; tail -4 gpurun_out/pp_t.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
PP_VARIANTS=${PP_VARIANTS:-old,auto,0,2,3} timeout -k 10 400 python tools/bench_pp.py 256 --check 2>&1 | grep -v amdgpu.ids | cut -c1-260
