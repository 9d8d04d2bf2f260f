"""Weight-gradient kernels on the ResNet-50 shapes (bs from argv, default 256), operands COLD (rotating copies larger
than the Infinity Cache): time per launch (MFMA kernel + slab reduce) and TFLOP/s.  Two variants interleaved in
one process, selected by BENCH_WGRAD_VAR (default CNK_WGRAD3: 0 = the one-tap-per-workgroup kernel, 1 = the 3x3 all-taps kernel
where it applies; CNK_WGRAD_BIG: 128x128 tiles vs 256x128 / 128x256 on one 512-thread workgroup).  python tools/bench_wgrad.py [N]"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from convnets_amd import _lib
if os.environ.get('CNK_LIB_OVERRIDE'):
    _lib.LIB_PATH = os.environ['CNK_LIB_OVERRIDE']      # ablated debug build (tools/ablate_w3.sh)
from convnets_amd import ops
CL = torch.channels_last
lib = _lib.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
SHAPES = [(64, 64, 3, 1, 1, 56), (128, 128, 3, 1, 1, 28), (256, 256, 3, 1, 1, 14), (512, 512, 3, 1, 1, 7),
          (128, 128, 3, 2, 1, 56), (256, 256, 3, 2, 1, 28),
          (64, 256, 1, 1, 0, 56), (256, 64, 1, 1, 0, 56), (256, 128, 1, 1, 0, 56), (128, 512, 1, 1, 0, 28), (512, 128, 1, 1, 0, 28),
          (256, 1024, 1, 1, 0, 14), (1024, 256, 1, 1, 0, 14), (512, 2048, 1, 1, 0, 7), (2048, 512, 1, 1, 0, 7),
          (256, 512, 1, 2, 0, 56), (1024, 2048, 1, 2, 0, 14)]
if os.environ.get('WG_SHAPES'):
    SHAPES = [SHAPES[int(i)] for i in os.environ['WG_SHAPES'].split(',')]
P = lambda t: ctypes.c_void_p(t.data_ptr())
VAR = os.environ.get('BENCH_WGRAD_VAR', 'CNK_WGRAD3')
tot = {0: 0.0, 1: 0.0}
for Cin, Cout, k, s, p, H in SHAPES:
    g = torch.Generator(device='cuda').manual_seed(Cin + Cout + k + H)
    Ho = (H + 2 * p - k) // s + 1
    # COLD operands: rotate over enough copies that the 256 MB Infinity Cache cannot hold them (as in a training
    # step, where x was written a whole forward pass ago)
    per = (N * Cin * H * H + N * Cout * Ho * Ho) * 2
    copies = max(2, min(24, int(1.5e9 // per) + 1))
    xs = [torch.randn(N, Cin, H, H, device='cuda', generator=g).bfloat16().contiguous(memory_format=CL) for _ in range(copies)]
    dys = [torch.randn(N, Cout, Ho, Ho, device='cuda', generator=g).bfloat16().contiguous(memory_format=CL) for _ in range(copies)]
    desc = ops.make_desc(xs[0].shape, (Cout, Cin, k, k), (s, s), (p, p), _lib.CNK_BF16)
    nws = 0
    for mode in (0, 1):
        os.environ[VAR] = str(mode); _lib.reload_switches()
        nws = max(nws, lib.cnk_conv2d_workspace_bytes(ctypes.byref(desc), 2))
    ws = torch.empty(max(nws, 16), dtype=torch.uint8, device='cuda')
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    res, us = {}, {}
    for mode in (0, 1, 0, 1):
        os.environ[VAR] = str(mode); _lib.reload_switches()
        dw = torch.zeros(Cout, Cin, k, k, device='cuda')
        run = lambda i: lib.cnk_conv2d_wgrad(ctypes.byref(desc), P(dys[i % copies]), P(xs[i % copies]), P(dw), None, 0, P(ws), nws, st)
        for i in range(2):
            rc = run(i)
            assert rc == 0, lib.cnk_last_error()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 2 * copies
        e0.record()
        for i in range(reps):
            run(i + 2)
        e1.record()
        torch.cuda.synchronize()
        us[mode] = min(us.get(mode, 1e9), e0.elapsed_time(e1) * 1000 / reps)
        run(0)
        torch.cuda.synchronize()
        res[mode] = dw.clone()
    fl = 2.0 * N * Ho * Ho * Cout * Cin * k * k
    tot[0] += us[0]
    tot[1] += us[1]
    diff = (res[0] - res[1]).abs().max().item()
    ref = res[0].abs().max().item()
    ref = torch.nn.grad.conv2d_weight(xs[0].float(), (Cout, Cin, k, k), dys[0].float(), stride=s, padding=p)
    e0_, e1_ = [float((res[m] - ref).norm() / ref.norm()) for m in (0, 1)]
    print('%4d->%4d k%d s%d @%3d   %s=0 %7.1f us %5.0f TF   =1 %7.1f us %5.0f TF   rel err vs torch fp32 %.2e / %.2e' % (
        Cin, Cout, k, s, H, VAR, us[0], fl / us[0] / 1e6, us[1], fl / us[1] / 1e6, e0_, e1_), flush=True)
print('sum: %s=0 %.1f us, =1 %.1f us' % (VAR, tot[0], tot[1]))
