"""BN backward (unfused: reduce + final + apply) and BN statistics on the block-output shapes of ResNet-50 at batch 256, bf16,
cold operands; two builds of libcnk interleaved in one process when CNK_LIB_B is given (A = the in-tree build).
    [CNK_LIB_B=path/to/other/libcnk.so] python tools/bench_bn.py"""
import ctypes, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from convnets_amd import _lib
libs = {'A': _lib.load()}
if os.environ.get('CNK_LIB_B'):
    b = ctypes.CDLL(os.environ['CNK_LIB_B'])
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(b, name)
        fn.restype, fn.argtypes = res, args
    libs['B'] = b
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
code = _lib.CNK_BF16
N = 256
tot = {k: [0.0, 0.0] for k in libs}
for C, H in ((256, 56), (512, 28), (1024, 14), (2048, 7)):
    M = N * H * H
    per = M * C * 2
    cop = max(2, min(8, int(1.2e9 // (3 * per)) + 1))
    xs = [torch.randn(M, C, device='cuda').bfloat16() for _ in range(cop)]
    dys = [torch.randn(M, C, device='cuda').bfloat16() for _ in range(cop)]
    dx = torch.empty_like(xs[0])
    gamma, beta = torch.ones(C, device='cuda'), torch.zeros(C, device='cuda')
    sm, si = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
    dg, db = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
    lib0 = libs['A']
    nws = lib0.cnk_bn_workspace_bytes(M, C)
    ws = torch.empty(nws // 4 + 4, device='cuda')

    def bwd(lib, i):
        assert lib.cnk_bn_bwd(P(dys[i % cop]), P(xs[i % cop]), None, None, P(gamma), P(beta), P(sm), P(si), P(dx), None, P(dg), P(db), 0,
                              None, 0, M, C, 0, code, P(ws), nws, st) == 0

    def stats(lib, i):
        assert lib.cnk_bn_fwd_train(P(xs[i % cop]), P(gamma), P(beta), None, None, None, None, None, P(sm), P(si), None, 0, M, C, 1e-5, 0.1,
                                    0, code, P(ws), nws, st) == 0
    out = []
    for k, lib in libs.items():
        pass
    times = {(k, n): [] for k in libs for n in ('bwd', 'stats')}
    for rnd in range(5):
        for k, lib in libs.items():
            for n, fn in (('bwd', bwd), ('stats', stats)):
                fn(lib, 0); fn(lib, 1)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                reps = 3 * cop
                for i in range(reps):
                    fn(lib, i + 2)
                e1.record()
                torch.cuda.synchronize()
                times[(k, n)].append(e0.elapsed_time(e1) * 1000 / reps)
    line = '%5d ch @%3d^2:' % (C, H)
    for k in libs:
        b_, s_ = statistics.median(times[(k, 'bwd')]), statistics.median(times[(k, 'stats')])
        tot[k][0] += b_; tot[k][1] += s_
        line += '   %s  backward %6.1f us (%.2f TB/s over 5 passes)  statistics %6.1f us (%.2f TB/s)' % (k, b_, 5 * per / b_ / 1e6, s_, per / s_ / 1e6)
    print(line, flush=True)
for k in libs:
    print('sum %s: backward %.1f us, statistics %.1f us' % (k, tot[k][0], tot[k][1]))
