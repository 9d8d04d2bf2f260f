"""Each 1x1 / stride-1 forward conv of the ResNet-50 step (batch 256, cold operands) on the implicit-GEMM kernel and on the ring
GEMM (CNK_GEMM1X1 = 0 / rule / 2 / 4), with and without the BN statistics partials in the epilogue: what the statistics cost
per shape.      python tools/probe_fwd1x1.py"""
import ctypes, sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from convnets_amd import _lib, ops
CL = torch.channels_last
lib = _lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
N = 256
def bench(Cin, Cout, H, g1, stats):
    with _lib.switches(CNK_GEMM1X1=g1):
        per = (N*Cin*H*H + N*Cout*H*H)*2
        copies = max(2, min(16, int(2.0e9 // per) + 1))
        xs = [torch.randn(N, Cin, H, H, device='cuda').bfloat16().contiguous(memory_format=CL) for _ in range(copies)]
        ys = [torch.empty(N, Cout, H, H, device='cuda', dtype=torch.bfloat16).contiguous(memory_format=CL) for _ in range(copies)]
        w = torch.randn(Cout, Cin, 1, 1, device='cuda') * 0.05
        desc = ops.make_desc(xs[0].shape, w.shape, (1, 1), (0, 0), _lib.CNK_BF16)
        wf, wd = ops.PackedWeights().get(w, desc, True)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        nsl = lib.cnk_conv2d_fwd_bn_partial_slices(ctypes.byref(desc))
        part = torch.empty(max(nsl, 1) * Cout * 3, device='cuda')
        run = lambda i: lib.cnk_conv2d_fwd(ctypes.byref(desc), P(xs[i % copies]), P(wf), None, P(ys[i % copies]), P(part) if (stats and nsl) else None, None, 0, st)
        res = []
        for r in range(3):
            for i in range(2): assert run(i) == 0, lib.cnk_last_error()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 3 * copies
            e0.record()
            for i in range(reps): run(i + 2)
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) * 1000 / reps)
        return statistics.median(res), per
for Cin, Cout, H in [(128, 512, 28), (256, 1024, 14), (512, 2048, 7), (64, 256, 56), (1024, 256, 14), (512, 128, 28)]:
    out = []
    for g1 in ('0', '1', '2', '4'):
        for stats in (1, 0):
            try:
                us, per = bench(Cin, Cout, H, g1, stats)
                out.append('g1=%s st=%d %6.1f' % (g1, stats, us))
            except Exception as e:
                out.append('g1=%s st=%d fail' % (g1, stats))
    print('%4d->%4d @%2d floor %5.1f | %s' % (Cin, Cout, H, per / 5.5e6, ' | '.join(out)), flush=True)
