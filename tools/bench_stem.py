"""Backward of the stem's BN + ReLU + MaxPool2d(3, 2, 1) at the bench shape (256 x 112 x 112 x 64, bf16), two forms
interleaved in one process: `pair` = cnk_maxpool2d_bwd + cnk_bn_bwd, `gather` = cnk_bn_bwd_maxpool.
    python tools/bench_stem.py [N]"""
import ctypes, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from convnets_amd import _lib
lib = _lib.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H = W = 112; C = 64; k, s, p = 3, 2, 1
Ho = Wo = 56
M = N * H * W
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
code = _lib.CNK_BF16
COP = 3
xs = [torch.randn(N, H, W, C, device='cuda').bfloat16() for _ in range(COP)]
dys = [torch.randn(N, Ho, Wo, C, device='cuda').bfloat16() for _ in range(COP)]
gamma, beta = torch.ones(C, device='cuda'), torch.zeros(C, device='cuda')
nws = lib.cnk_bn_workspace_bytes(M, C)
ws = torch.empty(nws // 4 + 4, device='cuda')
sm, si = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
idxs = []
for x in xs:
    assert lib.cnk_bn_fwd_train(P(x), P(gamma), P(beta), None, None, None, None, None, P(sm), P(si), None, 0, M, C, 1e-5, 0.1, 1, code, P(ws), nws, st) == 0
    y = torch.empty(N, Ho, Wo, C, dtype=torch.bfloat16, device='cuda')
    idx = torch.empty(N, Ho, Wo, C, dtype=torch.uint8, device='cuda')
    assert lib.cnk_maxpool2d_fwd_bn(P(x), ctypes.c_void_p(ws.data_ptr()), ctypes.c_void_p(ws.data_ptr() + 4 * C), 1, P(y), P(idx), N, H, W, C, k, k, s, s, p, p, code, st) == 0
    idxs.append(idx)
da, dx = torch.empty_like(xs[0]), torch.empty_like(xs[0])
dg, db = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')


def pair(i):
    j = i % COP
    assert lib.cnk_maxpool2d_bwd(P(dys[j]), P(idxs[j]), P(da), N, H, W, C, k, k, s, s, p, p, code, st) == 0
    assert lib.cnk_bn_bwd(P(da), P(xs[j]), None, None, P(gamma), P(beta), P(sm), P(si), P(dx), None, P(dg), P(db), 0, None, 0, M, C, 1, code, P(ws), nws, st) == 0


def gather(i):
    j = i % COP
    assert lib.cnk_bn_bwd_maxpool(P(dys[j]), P(idxs[j]), P(xs[j]), P(gamma), P(beta), P(sm), P(si), P(dx), P(dg), P(db), 0, N, H, W, C, k, k, s, s, p, p, 1, code, P(ws), nws, st) == 0, lib.cnk_last_error()


ypool = torch.empty(N, Ho, Wo, C, dtype=torch.bfloat16, device='cuda')
ipool = torch.empty(N, Ho, Wo, C, dtype=torch.uint8, device='cuda')


def poolfwd(i):
    j = i % COP
    assert lib.cnk_maxpool2d_fwd_bn(P(xs[j]), ctypes.c_void_p(ws.data_ptr()), ctypes.c_void_p(ws.data_ptr() + 4 * C), 1, P(ypool), P(ipool), N, H, W, C, k, k, s, s, p, p, code, st) == 0


times = {'pair': [], 'gather': [], 'poolfwd': []}
for rnd in range(4):
    for name, fn in (('pair', pair), ('gather', gather), ('poolfwd', poolfwd)):
        fn(0); fn(1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(12):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) * 1000 / 12)
for name, v in times.items():
    print('%-7s %7.1f us  (%s)' % (name, statistics.median(v), ' '.join('%.0f' % t for t in v)))
