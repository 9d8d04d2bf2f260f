"""Dense 1x1 / stride-1 convs of ResNet-50 at batch N (forward with the BN statistics partials; plain data gradient): the
single-buffer implicit-GEMM kernel (CNK_GEMM1X1=0) against the four-stage ring GEMM of conv_gemm1x1.hip (CNK_GEMM1X1=2),
interleaved in ONE process on cold operands (rotating copies), median of 5 rounds.
    python tools/bench_g1.py [N] [--check]"""
import ctypes, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from convnets_amd import _lib, ops
CL = torch.channels_last
lib = _lib.load()
args = [a for a in sys.argv[1:] if not a.startswith('--')]
N = int(args[0]) if args else 256
CHECK = '--check' in sys.argv
SHAPES = [(512, 128, 28), (128, 512, 28), (512, 256, 28), (256, 1024, 14), (1024, 256, 14), (1024, 512, 14), (512, 2048, 7), (2048, 512, 7),
          (64, 256, 56), (256, 64, 56), (256, 128, 56)]
if os.environ.get('G1_SHAPES'):
    SHAPES = [SHAPES[int(i)] for i in os.environ['G1_SHAPES'].split(',')]
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
VARS = (('old', '0'), ('n128', '2'), ('n256', '4'))
tot = {v: [0.0, 0.0, 0.0] for v, _ in VARS}
for Cin, Cout, H in SHAPES:
    torch.manual_seed(5)
    per = (N * Cin * H * H + N * Cout * H * H) * 2
    copies = max(2, min(16, int(1.2e9 // per) + 1))
    xs = [torch.randn(N, Cin, H, H, device='cuda').bfloat16().contiguous(memory_format=CL) for _ in range(copies)]
    ys = [torch.randn(N, Cout, H, H, device='cuda').bfloat16().contiguous(memory_format=CL) for _ in range(copies)]
    w = torch.randn(Cout, Cin, 1, 1, device='cuda') * 0.05
    desc = ops.make_desc(xs[0].shape, w.shape, (1, 1), (0, 0), _lib.CNK_BF16)
    pk = ops.PackedWeights()
    wf, wd = pk.get(w, desc, True)
    part = torch.zeros((N * H * H // 64 + 8) * Cout * 3, device='cuda')
    info = {}
    for v, mode in VARS:
        os.environ['CNK_GEMM1X1'] = mode
        _lib.reload_switches()
        nsl = lib.cnk_conv2d_fwd_bn_partial_slices(ctypes.byref(desc))
        yo, dxo = torch.empty_like(ys[0]), torch.empty_like(xs[0])
        assert lib.cnk_conv2d_fwd(ctypes.byref(desc), P(xs[0]), P(wf), None, P(yo), P(part), None, 0, st) == 0, lib.cnk_last_error()
        assert lib.cnk_conv2d_dgrad(ctypes.byref(desc), P(ys[0]), P(wd), P(dxo), None, None, None, None, 0, st) == 0, lib.cnk_last_error()
        torch.cuda.synchronize()
        pt = part[:nsl * Cout * 3].view(nsl, Cout, 3).double()
        n, mean, m2 = pt[:, :, 0], pt[:, :, 1], pt[:, :, 2]
        tn = n.sum(0)
        tmean = (n * mean).sum(0) / tn
        tvar = (m2.sum(0) + (n * (mean - tmean) ** 2).sum(0)) / tn
        info[v] = (yo, dxo, nsl, tmean, tvar)
    same = all(torch.equal(info['old'][0], info[v][0]) and torch.equal(info['old'][1], info[v][1]) for v, _ in VARS[1:])
    msg = 'outputs %s, var rel diff %.1e' % ('bit-identical' if same else 'DIFFER', max(((info['old'][4] - info[v][4]).abs() / info['old'][4]).max().item() for v, _ in VARS[1:]))
    info['new'] = info['n256']
    if CHECK:
        yr = torch.nn.functional.conv2d(xs[0].float(), w.bfloat16().float())
        msg += ', new vs torch fp32 %.2e' % ((info['new'][0].float() - yr).norm() / yr.norm()).item()
        yb = info['new'][0].double()
        msg += ', stats vs tensor: mean %.1e var %.1e' % (((info['new'][3] - yb.mean((0, 2, 3))).abs().max() / yb.abs().mean()).item(),
                                                          ((info['new'][4] - yb.var((0, 2, 3), unbiased=False)).abs() / info['new'][4]).max().item())
    # the training step's data gradient: gradient addend + first pass of the producing BN(+ReLU) layer's backward fused in
    mean, invstd = torch.zeros(Cin, device='cuda'), torch.ones(Cin, device='cuda')
    gamma, beta = torch.ones(Cin, device='cuda'), torch.zeros(Cin, device='cuda')
    bpart = torch.zeros((N * H * H // 64 + 8) * Cin * 2, device='cuda')
    bnx = [torch.randn(N, Cin, H, H, device='cuda').bfloat16().contiguous(memory_format=CL) for _ in range(min(copies, 4))]
    fuses = [_lib.BnBwdFuse(b.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None, bpart.data_ptr(), 1) for b in bnx]
    runs = {'fwd': lambda i: lib.cnk_conv2d_fwd(ctypes.byref(desc), P(xs[i % copies]), P(wf), None, P(ys[i % copies]), P(part), None, 0, st),
            'dgrad': lambda i: lib.cnk_conv2d_dgrad(ctypes.byref(desc), P(ys[i % copies]), P(wd), P(xs[i % copies]), None, None, None, None, 0, st),
            'dgbn': lambda i: lib.cnk_conv2d_dgrad(ctypes.byref(desc), P(ys[i % copies]), P(wd), P(xs[i % copies]), None, None, ctypes.byref(fuses[i % len(fuses)]), None, 0, st)}
    times = {(v, k): [] for v, _ in VARS for k in runs}
    for rnd in range(5):
        for v, mode in VARS:
            os.environ['CNK_GEMM1X1'] = mode
            _lib.reload_switches()
        _lib.reload_switches()
            for k, run in runs.items():
                run(0); run(1)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 2 * copies
                e0.record()
                for i in range(reps):
                    assert run(i + 2) == 0
                e1.record()
                torch.cuda.synchronize()
                times[(v, k)].append(e0.elapsed_time(e1) * 1000 / reps)
    fl = 2.0 * N * H * H * Cout * Cin
    line = '%4d->%4d @%3d' % (Cin, Cout, H)
    for v, _ in VARS:
        f, d, b = (statistics.median(times[(v, k)]) for k in ('fwd', 'dgrad', 'dgbn'))
        tot[v][0] += f; tot[v][1] += d; tot[v][2] += b
        line += '  %s %5.1f / %5.1f / %5.1f' % (v, f, d, b)
    print(line + '  us fwd / dgrad / dgrad+BN;  ' + msg, flush=True)
for v in tot:
    print('sum %s: fwd %.1f us, dgrad %.1f us, dgrad with the BN backward fused %.1f us' % (v, tot[v][0], tot[v][1], tot[v][2]))
