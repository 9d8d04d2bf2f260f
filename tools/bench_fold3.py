"""cnk_conv2d_bwd_bnfold (csrc/conv_fold3.hip) alone, cold operands, per workgroup count (CNK_FOLD3_NWG, re-read per setting):
    python tools/bench_fold3.py [N] [nwg ...]"""
import ctypes, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from convnets_amd import _lib, ops
CL = torch.channels_last
lib = _lib.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
NWG = [int(a) for a in sys.argv[2:]] or [256, 384, 512, 768, 1024]
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
Cin, Cout, H = 64, 256, 56
M = N * H * H
copies = 4
mk = lambda C: [torch.randn(N, C, H, H, device='cuda').bfloat16().contiguous(memory_format=CL) for _ in range(copies)]
g, a, da, u = mk(Cout), mk(Cin), mk(Cin), mk(Cin)
w = torch.randn(Cout, Cin, 1, 1, device='cuda') * 0.05
mean, invstd, coef = torch.zeros(Cout, device='cuda'), torch.ones(Cout, device='cuda'), torch.rand(3 * Cout, device='cuda')
m2, i2, g2, b2 = torch.zeros(Cin, device='cuda'), torch.ones(Cin, device='cuda'), torch.ones(Cin, device='cuda'), torch.zeros(Cin, device='cuda')
dw = torch.zeros(Cout, Cin, 1, 1, device='cuda')
for nwg in NWG:
    os.environ['CNK_FOLD3_NWG'] = str(nwg)
    _lib.reload_switches()
    desc = ops.make_desc(a[0].shape, w.shape, (1, 1), (0, 0), _lib.CNK_BF16)
    D = ctypes.byref(desc)
    wf, wd = ops.PackedWeights().get(w, desc, True)
    w_fold = torch.empty(lib.cnk_conv2d_bnfold_image_bytes(D), dtype=torch.uint8, device='cuda')
    bias_fold = torch.empty(Cin, device='cuda')
    _lib.check(lib.cnk_conv2d_bnfold_prepare(D, P(w), P(wd), P(mean), P(invstd), P(coef), P(w_fold), P(bias_fold), st()))
    n1 = lib.cnk_conv2d_bwd_bnfold_workspace_bytes(D)
    ws1 = torch.empty(n1, dtype=torch.uint8, device='cuda')
    part = torch.zeros(lib.cnk_conv2d_bwd_bnfold_bn_slices(D) * Cin * 2, device='cuda')
    fuses = [_lib.BnBwdFuse(u[i].data_ptr(), m2.data_ptr(), i2.data_ptr(), g2.data_ptr(), b2.data_ptr(), None, part.data_ptr(), 1) for i in range(copies)]
    run = lambda i: lib.cnk_conv2d_bwd_bnfold(D, P(g[i]), P(a[i]), P(w_fold), P(bias_fold), P(da[i]), None, ctypes.byref(fuses[i]), P(w), P(mean), P(invstd),
                                              P(coef), P(dw), 0, P(ws1), n1, st())
    ts = []
    for rnd in range(4):
        for i in range(2):
            assert run(i) == 0, lib.cnk_last_error()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(8):
            run((i + 2) % copies)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1000 / 8)
    t = statistics.median(ts)
    print('nwg %4d: %6.1f us  (%.2f TB/s of g + a + BN input + da; slabs %.0f MB each way)'
          % (nwg, t, M * (Cout + 3 * Cin) * 2 / t / 1e6, lib.cnk_conv2d_bwd_bnfold_bn_slices(D) * 321 * 64 * 4 / 1e6), flush=True)
