"""Every 1x1 data gradient of the ResNet-50 step as the step issues it (cold operands, rotating copies), per layer:
conv1 of a bottleneck (4f -> f): gradient addend + packed ReLU mask + fused backward of the block-input BN (saved mask bits);
conv3 (f -> 4f): fused backward of BN2 (mask re-derived); shortcut convs: plain.  Prints us, algorithmic TB/s, TFLOP/s.
    python tools/bench_dgrad1x1.py [N]      D1_ROUNDS=3  D1_ONLY=c1,c3,sc"""
import ctypes, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from convnets_amd import _lib, ops
CL = torch.channels_last
lib = _lib.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ROUNDS = int(os.environ.get('D1_ROUNDS', '3'))
ONLY = os.environ.get('D1_ONLY', 'c1,c3,sc').split(',')
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
# (kind, Cin, Cout, H, stride, count in ResNet-50)
LAYERS = [('c1', 256, 64, 56, 1, 2), ('c1', 256, 128, 56, 1, 1), ('c1', 512, 128, 28, 1, 3), ('c1', 512, 256, 28, 1, 1),
          ('c1', 1024, 256, 14, 1, 5), ('c1', 1024, 512, 14, 1, 1), ('c1', 2048, 512, 7, 1, 2),
          ('c3', 64, 256, 56, 1, 3), ('c3', 128, 512, 28, 1, 4), ('c3', 256, 1024, 14, 1, 6), ('c3', 512, 2048, 7, 1, 3),
          ('c0', 64, 64, 56, 1, 1),
          ('sc', 64, 256, 56, 1, 1), ('sc', 256, 512, 56, 2, 1), ('sc', 512, 1024, 28, 2, 1), ('sc', 1024, 2048, 14, 2, 1)]
tot = 0.0
for kind, Cin, Cout, H, s, count in LAYERS:
    if kind not in ONLY and not (kind == 'c0' and 'c1' in ONLY):
        continue
    Ho = (H - 1) // s + 1
    torch.manual_seed(3)
    per = (N * Cin * H * H * 3 + N * Cout * Ho * Ho) * 2
    copies = max(2, min(12, int(2.0e9 // per) + 1))
    dys = [torch.randn(N, Cout, Ho, Ho, device='cuda').bfloat16().contiguous(memory_format=CL) for _ in range(copies)]
    dxs = [torch.empty(N, Cin, H, H, device='cuda', dtype=torch.bfloat16).contiguous(memory_format=CL) for _ in range(copies)]
    w = torch.randn(Cout, Cin, 1, 1, device='cuda') * 0.05
    desc = ops.make_desc(dxs[0].shape, w.shape, (s, s), (0, 0), _lib.CNK_BF16)
    wf, wd = ops.PackedWeights().get(w, desc, True)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    M = N * H * H
    side = kind in ('c1', 'c3', 'c0')
    bytes_alg = (N * Cout * Ho * Ho + M * Cin) * 2.0
    fuses, adds, amasks = [None] * copies, [None] * copies, [None] * copies
    if side:
        nb = lib.cnk_conv2d_dgrad_bn_slices(ctypes.byref(desc))
        bpart = torch.zeros(max(nb, 1) * Cin * 2, device='cuda')
        mean, invstd = torch.zeros(Cin, device='cuda'), torch.ones(Cin, device='cuda')
        gamma, beta = torch.ones(Cin, device='cuda'), torch.zeros(Cin, device='cuda')
        bnx = [torch.randn(N, Cin, H, H, device='cuda').bfloat16().contiguous(memory_format=CL) for _ in range(copies)]
        keep = []
        bytes_alg += M * Cin * 2.0
        for i in range(copies):
            km = None
            if kind == 'c1':
                km = torch.randint(0, 256, (M * Cin // 8,), device='cuda', dtype=torch.uint8)
                adds[i] = torch.randn(N, Cin, H, H, device='cuda').bfloat16().contiguous(memory_format=CL)
                amasks[i] = torch.randint(0, 256, (M * Cin // 8,), device='cuda', dtype=torch.uint8)
            keep.append(km)
            fuses[i] = _lib.BnBwdFuse(bnx[i].data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                      km.data_ptr() if km is not None else None, bpart.data_ptr(), 1)
        if kind == 'c1':
            bytes_alg += M * Cin * 2.0 + 2 * M * Cin / 8

    def run(i):
        k = i % copies
        return lib.cnk_conv2d_dgrad(ctypes.byref(desc), P(dys[k]), P(wd), P(dxs[k]), P(adds[k]), P(amasks[k]),
                                    ctypes.byref(fuses[k]) if fuses[k] is not None else None, None, 0, st)
    times = []
    for rnd in range(ROUNDS):
        for i in range(2):
            assert run(i) == 0, lib.cnk_last_error()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 2 * copies
        e0.record()
        for i in range(reps):
            run(i + 2)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) * 1000 / reps)
    us = statistics.median(times)
    fl = 2.0 * N * Ho * Ho * Cout * Cin
    tot += us * count
    print('%s dgrad of %4d->%4d @%3d s%d  x%d  %7.1f us  %5.2f TB/s alg  %5.0f TF   (floor at 5.5 TB/s %6.1f us)' % (
        kind, Cin, Cout, H, s, count, us, bytes_alg / us / 1e6, fl / us / 1e6, bytes_alg / 5.5e6), flush=True)
    del dys, dxs, fuses, adds, amasks
    torch.cuda.empty_cache()
print('sum over the step (count-weighted): %.1f us' % tot)
