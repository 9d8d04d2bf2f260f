"""BN forward/backward through the C ABI at the ResNet-50 bs=256 shapes (tuning aid): time per call and
algorithmic GB/s; per-kernel split comes from running it under rocprofv3 --kernel-trace --stats."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from convnets_amd import ops, _lib
lib = _lib.load()
N = 256
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
def timeit(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
tot = {}
for C, H, cnt in [(64, 112, 1), (64, 56, 6), (256, 56, 4), (128, 28, 8), (512, 28, 5), (256, 14, 12), (1024, 14, 7), (512, 7, 6), (2048, 7, 4)]:
    M = N * H * H
    x = torch.randn(M, C, device='cuda', dtype=torch.bfloat16)
    res = torch.randn_like(x); y = torch.empty_like(x); dy = torch.randn_like(x); dx = torch.empty_like(x)
    g, b = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda') * 0.1
    rm, rv = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
    sm, si = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
    dg, db = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    ws = ops.workspace(lib.cnk_bn_workspace_bytes(M, C), x.device)
    mask = torch.empty(lib.cnk_bn_mask_bytes(M, C, 1), dtype=torch.uint8, device='cuda')
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    nsl = (M + 127) // 128
    part = torch.empty(nsl, C, 3, device='cuda')
    part[:, :, 0] = 128; part[:, :, 1] = 0; part[:, :, 2] = 128
    E = M * C * 2
    for mode, r, mk in (('relu', None, None), ('add+relu', res, mask)):
        f = lambda: _lib.check(lib.cnk_bn_fwd_train(P(x), P(g), P(b), P(r), P(y), P(mk), P(rm), P(rv), P(sm), P(si), P(part), nsl,
                                                    M, C, 1e-5, 0.1, 1, 1, P(ws), ws.numel(), st))
        w = lambda: _lib.check(lib.cnk_bn_bwd(P(dy), P(x), None, P(mk), P(g), P(b), P(sm), P(si), P(dx), None, P(dg), P(db), 1, None, 0,
                                              M, C, 1, 1, P(ws), ws.numel(), st))
        tf, tb = timeit(f), timeit(w)
        nf = 2 + (1 if r is not None else 0)
        print('bn C=%4d @%3d %-8s | fwd %7.1f us %5.0f GB/s | bwd %7.1f us %5.0f GB/s' % (C, H, mode, tf, nf * E / tf / 1e3, tb, 5 * E / tb / 1e3), flush=True)
        if mode == 'relu' or C in (256, 512, 1024, 2048):
            k = cnt if mode == 'relu' else {256: 3, 512: 4, 1024: 6, 2048: 3}.get(C, 0) if H in (56, 28, 14, 7) else 0
            tot['fwd'] = tot.get('fwd', 0) + k * tf; tot['bwd'] = tot.get('bwd', 0) + k * tb
print('weighted ResNet-50-like total: fwd %.2f ms  bwd %.2f ms' % (tot['fwd'] / 1e3, tot['bwd'] / 1e3))
