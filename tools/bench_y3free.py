"""Forward of a bottleneck's last block (conv 1x1 f -> 4f, BN, += identity, ReLU) per ResNet-50 stage at batch N, cold operands:
stored conv output (cnk_conv2d_fwd + cnk_bn_fwd_train) against two conv passes (statistics only; conv + BN apply in the epilogue).
    python tools/bench_y3free.py [N]"""
import ctypes, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from convnets_amd import _lib, ops
CL = torch.channels_last
lib = _lib.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
tot = [0.0, 0.0]
for Cin, Cout, H, count in [(64, 256, 56, 3), (128, 512, 28, 4), (256, 1024, 14, 6), (512, 2048, 7, 3)]:
    M = N * H * H
    copies = max(2, min(6, int(3.0e9 // (M * (Cin + 3 * Cout) * 2)) + 1))
    mk = lambda C: [torch.randn(N, C, H, H, device='cuda').bfloat16().contiguous(memory_format=CL) for _ in range(copies)]
    xs, rs, y3s, ys = mk(Cin), mk(Cout), mk(Cout), mk(Cout)
    w = torch.randn(Cout, Cin, 1, 1, device='cuda') * 0.05
    d = ops.make_desc(xs[0].shape, w.shape, (1, 1), (0, 0), _lib.CNK_BF16)
    wf, _ = ops.PackedWeights().get(w, d, False)
    gamma, beta = torch.ones(Cout, device='cuda'), torch.zeros(Cout, device='cuda')
    sm, si = torch.empty(Cout, device='cuda'), torch.empty(Cout, device='cuda')
    rm, rv = torch.zeros(Cout, device='cuda'), torch.ones(Cout, device='cuda')
    nws = lib.cnk_bn_workspace_bytes(M, Cout)
    ws = torch.empty(nws, dtype=torch.uint8, device='cuda')
    mask = torch.empty(lib.cnk_bn_mask_bytes(M, Cout, _lib.CNK_BF16), dtype=torch.uint8, device='cuda')
    D = ctypes.byref(d)
    nsl = lib.cnk_conv2d_fwd_bn_partial_slices(D)
    part = torch.zeros(max(nsl, 1), Cout, 3, device='cuda')
    with _lib.switches(CNK_GEMM1X1='0'):
        nsl0 = lib.cnk_conv2d_fwd_bn_partial_slices(D)
    part0 = torch.zeros(max(nsl0, 1), Cout, 3, device='cuda')
    coef = torch.zeros(2 * Cout, device='cuda')

    def stored(i):
        k = i % copies
        assert lib.cnk_conv2d_fwd(D, P(xs[k]), P(wf), None, P(y3s[k]), P(part), None, 0, st()) == 0
        assert lib.cnk_bn_fwd_train(P(y3s[k]), P(gamma), P(beta), P(rs[k]), P(ys[k]), P(mask), P(rm), P(rv), P(sm), P(si), P(part), nsl, M, Cout,
                                    1e-5, 0.1, 1, _lib.CNK_BF16, P(ws), nws, st()) == 0

    def twopass(i):
        k = i % copies
        assert lib.cnk_conv2d_fwd(D, P(xs[k]), P(wf), None, None, P(part0), None, 0, st()) == 0, lib.cnk_last_error()
        assert lib.cnk_bn_fwd_train(P(xs[k]), P(gamma), P(beta), None, None, None, P(rm), P(rv), P(sm), P(si), P(part0), nsl0, M, Cout, 1e-5, 0.1,
                                    1, _lib.CNK_BF16, P(ws), nws, st()) == 0
        coef.copy_(ws[:8 * Cout].view(torch.float32))
        assert lib.cnk_conv2d_fwd_bn_apply(D, P(xs[k]), P(wf), P(coef), P(rs[k]), None, 1, P(ys[k]), P(mask), st()) == 0, lib.cnk_last_error()
    res = {}
    for rnd in range(3):
        for name, fn in (('stored', stored), ('twopass', twopass)):
            fn(0); fn(1)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 2 * copies
            e0.record()
            for i in range(reps): fn(i + 2)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(name, []).append(e0.elapsed_time(e1) * 1000 / reps)
    a, b = statistics.median(res['stored']), statistics.median(res['twopass'])
    tot[0] += a * count; tot[1] += b * count
    print('%4d->%4d @%2d x%d: stored %6.1f us, two passes %6.1f us' % (Cin, Cout, H, count, a, b), flush=True)
print('count-weighted over the step: %.0f -> %.0f us' % (tot[0], tot[1]))
