"""Per-op timing on the GPU box (debug/tuning aid): conv fwd/dgrad/wgrad and BN fwd/bwd at the
ResNet-50 bs=256 shapes, reported as TFLOP/s and algorithmic GB/s."""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from convnets_amd import ops, _lib

CL = torch.channels_last
dt = torch.bfloat16
N = int(os.environ.get('BN_N', 256))

def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

# (Cin, Cout, k, s, p, H)
CONVS = [(64, 64, 3, 1, 1, 56), (128, 128, 3, 1, 1, 28), (256, 256, 3, 1, 1, 14), (512, 512, 3, 1, 1, 7),
         (128, 128, 3, 2, 1, 56), (64, 256, 1, 1, 0, 56), (256, 64, 1, 1, 0, 56), (256, 128, 1, 1, 0, 56),
         (512, 128, 1, 1, 0, 28), (128, 512, 1, 1, 0, 28), (1024, 256, 1, 1, 0, 14), (256, 1024, 1, 1, 0, 14),
         (2048, 512, 1, 1, 0, 7), (512, 2048, 1, 1, 0, 7), (256, 512, 1, 2, 0, 56), (3, 64, 7, 2, 3, 224)]
which = sys.argv[1] if len(sys.argv) > 1 else 'all'
if which in ('all', 'conv'):
    for Cin, Cout, k, s, p, H in CONVS:
        x = torch.randn(N, Cin, H, H, device='cuda', dtype=dt).contiguous(memory_format=CL).requires_grad_(Cin > 3)
        w = (torch.randn(Cout, Cin, k, k, device='cuda') * 0.05).requires_grad_(True)
        pk = ops.PackedWeights()
        y = ops.Conv2dFn.apply(x, w, None, (s, s), (p, p), pk)
        dy = torch.randn_like(y)
        Ho = y.shape[2]
        fl = 2.0 * N * Ho * Ho * Cout * Cin * k * k
        by = (x.numel() + y.numel() + w.numel()) * 2
        lib = _lib.load()
        desc = ops.make_desc(x.shape, w.shape, (s, s), (p, p), _lib.CNK_BF16)
        wf, wd = pk.get(w, desc, Cin > 3)
        P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        t_f = timeit(lambda: lib.cnk_conv2d_fwd(ctypes.byref(desc), P(x), P(wf), None, P(y), None, None, 0, st))
        dx = torch.empty_like(x)
        t_d = timeit(lambda: lib.cnk_conv2d_dgrad(ctypes.byref(desc), P(dy), P(wd), P(dx), None, None, None, None, 0, st)) if Cin > 3 else float('nan')
        dw = torch.empty_like(w)
        ws = ops.workspace(lib.cnk_conv2d_workspace_bytes(ctypes.byref(desc), 2), x.device)
        t_w = timeit(lambda: lib.cnk_conv2d_wgrad(ctypes.byref(desc), P(dy), P(x), P(dw), None, 0, P(ws), ws.numel(), st))
        print('conv %4d->%4d k%d s%d @%3d | fwd %7.1f us %6.0f TF %5.0f GB/s | dgrad %7.1f us %6.0f TF | wgrad %7.1f us %6.0f TF'
              % (Cin, Cout, k, s, H, t_f * 1e3, fl / t_f / 1e9, by / t_f / 1e6, t_d * 1e3, fl / t_d / 1e9, t_w * 1e3, fl / t_w / 1e9), flush=True)
if which in ('all', 'bn'):
    for C, H in [(64, 112), (64, 56), (256, 56), (128, 28), (512, 28), (256, 14), (1024, 14), (512, 7), (2048, 7)]:
        x = torch.randn(N, C, H, H, device='cuda', dtype=dt).contiguous(memory_format=CL).requires_grad_(True)
        res = torch.randn_like(x).requires_grad_(True)
        g, b = torch.ones(C, device='cuda', requires_grad=True), torch.zeros(C, device='cuda', requires_grad=True)
        rm, rv = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
        E = x.numel() * 2
        for mode, r in (('relu', None), ('add+relu', res)):
            y = ops.BatchNormActFn.apply(x, g, b, r, rm, rv, 1e-5, 0.1, True)
            dy = torch.randn_like(y)
            t_f = timeit(lambda: ops.BatchNormActFn.apply(x, g, b, r, rm, rv, 1e-5, 0.1, True))
            def bw():
                x.grad = None
                y.backward(dy, retain_graph=True)
            t_b = timeit(bw)
            nf = 3 + (1 if r is not None else 0)
            nb = (7 if r is not None else 5) + (1 if r is not None else 0)
            print('bn C=%4d @%3d %-8s | fwd %7.1f us %5.0f GB/s (%d passes) | bwd %7.1f us %5.0f GB/s (%d passes)'
                  % (C, H, mode, t_f * 1e3, nf * E / t_f / 1e6, nf, t_b * 1e3, nb * E / t_b / 1e6, nb), flush=True)
