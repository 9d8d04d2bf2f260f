"""BN training forward with residual + ReLU on the ResNet-50 block-output shapes (batch 256, cold operands): with and without the
packed ReLU mask, partial statistics supplied (the apply pass dominates).      python tools/probe_bn_apply.py"""
import ctypes, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from convnets_amd import _lib
lib = _lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
N = 256
for C, H in ((256, 56), (512, 28), (1024, 14), (2048, 7)):
    M = N * H * H
    copies = max(2, min(6, int(3.0e9 // (3 * M * C * 2)) + 1))
    xs = [torch.randn(M, C, device='cuda').bfloat16() for _ in range(copies)]
    rs = [torch.randn(M, C, device='cuda').bfloat16() for _ in range(copies)]
    ys = [torch.empty(M, C, device='cuda', dtype=torch.bfloat16) for _ in range(copies)]
    gamma, beta = torch.ones(C, device='cuda'), torch.zeros(C, device='cuda')
    sm, si = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
    nws = lib.cnk_bn_workspace_bytes(M, C)
    ws = torch.empty(nws, dtype=torch.uint8, device='cuda')
    mask = torch.empty(lib.cnk_bn_mask_bytes(M, C, _lib.CNK_BF16), dtype=torch.uint8, device='cuda')
    nsl = 64
    part = torch.zeros(nsl, C, 3, device='cuda'); part[:, :, 0] = M / nsl; part[:, :, 2] = M / nsl
    out = []
    for wm in (True, False):
        def run(i):
            k = i % copies
            return lib.cnk_bn_fwd_train(P(xs[k]), P(gamma), P(beta), P(rs[k]), P(ys[k]), P(mask) if wm else None, None, None, P(sm), P(si),
                                        P(part), nsl, M, C, 1e-5, 0.1, 1, _lib.CNK_BF16, P(ws), nws, st())
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
        us = statistics.median(res)
        out.append('%s %6.1f us %.2f TB/s' % ('mask' if wm else 'none', us, 3.0 * M * C * 2 / us / 1e6))
    print('C %4d @%2d: %s' % (C, H, ' | '.join(out)), flush=True)
