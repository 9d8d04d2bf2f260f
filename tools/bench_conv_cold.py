"""Forward and data-gradient conv kernels on the ResNet-50 shapes (bs from argv, default 256) with COLD operands
(rotating copies larger than the Infinity Cache): time per launch, TFLOP/s and algorithmic GB/s.
    python tools/bench_conv_cold.py [N]"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from convnets_amd import _lib, ops
CL = torch.channels_last
lib = _lib.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
SHAPES = [(64, 64, 3, 1, 1, 56), (128, 128, 3, 1, 1, 28), (256, 256, 3, 1, 1, 14), (512, 512, 3, 1, 1, 7),
          (128, 128, 3, 2, 1, 56), (256, 256, 3, 2, 1, 28), (512, 512, 3, 2, 1, 14),
          (64, 64, 1, 1, 0, 56), (64, 256, 1, 1, 0, 56), (256, 64, 1, 1, 0, 56), (256, 128, 1, 1, 0, 56), (128, 512, 1, 1, 0, 28), (512, 128, 1, 1, 0, 28),
          (512, 256, 1, 1, 0, 28), (256, 1024, 1, 1, 0, 14), (1024, 256, 1, 1, 0, 14), (1024, 512, 1, 1, 0, 14), (512, 2048, 1, 1, 0, 7), (2048, 512, 1, 1, 0, 7),
          (256, 512, 1, 2, 0, 56), (512, 1024, 1, 2, 0, 28), (1024, 2048, 1, 2, 0, 14)]
P = lambda t: ctypes.c_void_p(t.data_ptr())
tot = {'fwd': 0.0, 'dgrad': 0.0}
for Cin, Cout, k, s, p, H in SHAPES:
    Ho = (H + 2 * p - k) // s + 1
    per = (N * Cin * H * H + N * Cout * Ho * Ho) * 2
    copies = max(2, min(24, int(1.5e9 // per) + 1))
    xs = [torch.randn(N, Cin, H, H, device='cuda').bfloat16().contiguous(memory_format=CL) for _ in range(copies)]
    ys = [torch.randn(N, Cout, Ho, Ho, device='cuda').bfloat16().contiguous(memory_format=CL) for _ in range(copies)]
    w = torch.randn(Cout, Cin, k, k, device='cuda') * 0.05
    desc = ops.make_desc(xs[0].shape, w.shape, (s, s), (p, p), _lib.CNK_BF16)
    pk = ops.PackedWeights()
    wf, wd = pk.get(w, desc, True)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    nsl = lib.cnk_conv2d_fwd_bn_partial_slices(ctypes.byref(desc))
    part = torch.empty(max(nsl, 1) * Cout * 3, device='cuda')
    runs = {'fwd': lambda i: lib.cnk_conv2d_fwd(ctypes.byref(desc), P(xs[i % copies]), P(wf), None, P(ys[i % copies]), P(part) if nsl else None, None, 0, st),
            'dgrad': lambda i: lib.cnk_conv2d_dgrad(ctypes.byref(desc), P(ys[i % copies]), P(wd), P(xs[i % copies]), None, None, None, None, 0, st)}
    out = []
    for name, run in runs.items():
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
        us = e0.elapsed_time(e1) * 1000 / reps
        tot[name] += us
        fl = 2.0 * N * Ho * Ho * Cout * Cin * k * k
        out.append('%s %7.1f us %5.0f TF %5.2f TB/s' % (name, us, fl / us / 1e6, per / us / 1e6))
    print('%4d->%4d k%d s%d @%3d   %s' % (Cin, Cout, k, s, H, '   '.join(out)), flush=True)
print('sum: fwd %.1f us, dgrad %.1f us' % (tot['fwd'], tot['dgrad']))
