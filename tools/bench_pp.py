"""Stride-1 3x3 conv forward (with BN statistics partials) and data gradient on the ResNet-50 shapes, cold operands,
several kernel variants INTERLEAVED in one process (cdna_hip_programming.md 5.4 rule 24):
    old   the r01 halo kernels (CNK_PP3X3=0)          auto  the ping-pong core, its own configuration choice
    0..3  the ping-pong core forced to configuration 0..3 where it fits (CNK_PP_CFG, conv_pp.hip pp_pick)
    nows  "auto" with the weight-stationary 64 -> 64 kernel off (CNK_PP_WS=0)
Prints per variant the median time per launch, TFLOP/s, a SHA-1 of the output bytes and the merged BN statistics.
    PP_VARIANTS=old,auto PP_SHAPES=0,1,2,3 python tools/bench_pp.py [N] [--check]
    (--check: also compare against a torch fp32 conv on the GPU)"""
import ctypes, hashlib, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from convnets_amd import _lib
if os.environ.get('CNK_LIB_OVERRIDE'):
    _lib.LIB_PATH = os.environ['CNK_LIB_OVERRIDE']      # ablated debug build (tools/ablate_pp.sh)
from convnets_amd import ops
CL = torch.channels_last
lib = _lib.load()
args = [a for a in sys.argv[1:] if not a.startswith('--')]
N = int(args[0]) if args else 256
CHECK = '--check' in sys.argv
SHAPES = [(64, 64, 56), (128, 128, 28), (256, 256, 14), (512, 512, 7)]
if os.environ.get('PP_SHAPES'):
    SHAPES = [SHAPES[int(i)] for i in os.environ['PP_SHAPES'].split(',')]
VARIANTS = os.environ.get('PP_VARIANTS', 'old,auto').split(',')
ROUNDS = int(os.environ.get('PP_ROUNDS', '3'))
P = lambda t: ctypes.c_void_p(t.data_ptr())
sha = lambda t: hashlib.sha1(t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()[:10]


def select(v):
    os.environ.pop('CNK_PP_CFG', None)
    os.environ['CNK_PP3X3'] = '0' if v == 'old' else '1'
    os.environ.pop('CNK_PP_FULL', None)
    os.environ.pop('CNK_PP_WS', None)
    if v == 'nows':                          # "auto" without the weight-stationary 64 -> 64 kernel
        os.environ['CNK_PP_WS'] = '0'
        v = 'auto'
    if v.endswith('f'):                      # "autof" / "0f": fused data gradients forced onto the ping-pong core
        os.environ['CNK_PP_FULL'] = '1'
        v = v[:-1]
    if v not in ('old', 'auto'):
        os.environ['CNK_PP_CFG'] = v
    _lib.reload_switches()                   # the library resolves its switches at load (csrc/common.h)


tot = {v: {'fwd': 0.0, 'dgrad': 0.0, 'dgbn': 0.0} for v in VARIANTS}
for Cin, Cout, H in SHAPES:
    torch.manual_seed(5)
    per = (N * Cin * H * H + N * Cout * H * H) * 2
    copies = max(2, min(24, int(1.5e9 // per) + 1))
    xs = [torch.randn(N, Cin, H, H, device='cuda').bfloat16().contiguous(memory_format=CL) for _ in range(copies)]
    ys = [torch.randn(N, Cout, H, H, device='cuda').bfloat16().contiguous(memory_format=CL) for _ in range(copies)]
    w = torch.randn(Cout, Cin, 3, 3, device='cuda') * 0.05
    desc = ops.make_desc(xs[0].shape, w.shape, (1, 1), (1, 1), _lib.CNK_BF16)
    pk = ops.PackedWeights()
    wf, wd = pk.get(w, desc, True)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    part = torch.zeros((N * H * H // 64 + 8) * Cout * 3, device='cuda')
    yo, dxo = torch.empty_like(ys[0]), torch.empty_like(xs[0])
    info = {}
    for v in VARIANTS:
        select(v)
        nsl = lib.cnk_conv2d_fwd_bn_partial_slices(ctypes.byref(desc))
        assert lib.cnk_conv2d_fwd(ctypes.byref(desc), P(xs[0]), P(wf), None, P(yo), P(part), None, 0, st) == 0, lib.cnk_last_error()
        assert lib.cnk_conv2d_dgrad(ctypes.byref(desc), P(ys[0]), P(wd), P(dxo), None, None, None, None, 0, st) == 0, lib.cnk_last_error()
        torch.cuda.synchronize()
        pt = part[:nsl * Cout * 3].view(nsl, Cout, 3).double()
        n, mean, m2 = pt[:, :, 0], pt[:, :, 1], pt[:, :, 2]
        tn = n.sum(0)
        tmean = (n * mean).sum(0) / tn
        tvar = (m2.sum(0) + (n * (mean - tmean) ** 2).sum(0)) / tn
        s = 'y %s dx %s slices %d n %d varsum %.8e' % (sha(yo), sha(dxo), nsl, int(tn[0]), tvar.sum())
        if CHECK:
            wb = w.bfloat16().float()
            yr = torch.nn.functional.conv2d(xs[0].float(), wb, padding=1)
            dxr = torch.nn.functional.conv_transpose2d(ys[0].float(), wb, padding=1)
            yb = yo.float()
            s += '  relerr y %.2e dx %.2e mean %.1e var %.1e' % (
                ((yb - yr).norm() / yr.norm()).item(), ((dxo.float() - dxr).norm() / dxr.norm()).item(),
                ((tmean - yb.mean((0, 2, 3)).double()).abs().max() / yb.abs().mean()).item(),
                ((tvar - yb.var((0, 2, 3), unbiased=False).double()).abs() / tvar).max().item())
        info[v] = s
    # the training step's data gradient: fused backward of the producing BN(+ReLU) layer (its input read in the epilogue,
    # ReLU mask re-derived, per-slice sums written) -- cnk_bn_bwd_fuse
    mean, invstd = torch.zeros(Cin, device='cuda'), torch.ones(Cin, device='cuda')
    gamma, beta = torch.ones(Cin, device='cuda'), torch.zeros(Cin, device='cuda')
    bpart = torch.zeros((N * H * H // 64 + 8) * Cin * 2, device='cuda')
    bnx = [torch.randn(N, Cin, H, H, device='cuda').bfloat16().contiguous(memory_format=CL) for _ in range(min(copies, 4))]
    fuses = [_lib.BnBwdFuse(b.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None, bpart.data_ptr(), 1) for b in bnx]
    runs = {'fwd': lambda i: lib.cnk_conv2d_fwd(ctypes.byref(desc), P(xs[i % copies]), P(wf), None, P(ys[i % copies]), P(part), None, 0, st),
            'dgrad': lambda i: lib.cnk_conv2d_dgrad(ctypes.byref(desc), P(ys[i % copies]), P(wd), P(xs[i % copies]), None, None, None, None, 0, st),
            'dgbn': lambda i: lib.cnk_conv2d_dgrad(ctypes.byref(desc), P(ys[i % copies]), P(wd), P(xs[i % copies]), None, None, ctypes.byref(fuses[i % len(fuses)]), None, 0, st)}
    times = {v: {'fwd': [], 'dgrad': [], 'dgbn': []} for v in VARIANTS}
    for rnd in range(ROUNDS):
        for v in VARIANTS:
            select(v)
            for name, run in runs.items():
                for i in range(2):
                    assert run(i) == 0, lib.cnk_last_error()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 2 * copies
                e0.record()
                for i in range(reps):
                    run(i + 2)
                e1.record()
                torch.cuda.synchronize()
                times[v][name].append(e0.elapsed_time(e1) * 1000 / reps)
    fl = 2.0 * N * H * H * Cout * Cin * 9
    for v in VARIANTS:
        out = []
        for name in ('fwd', 'dgrad', 'dgbn'):
            us = statistics.median(times[v][name])
            tot[v][name] += us
            out.append('%s %6.1f us %5.0f TF' % (name, us, fl / us / 1e6))
        print('%4d->%4d @%3d %5s  %s   %s' % (Cin, Cout, H, v, '   '.join(out), info[v]), flush=True)
for v in VARIANTS:
    print('sum %5s: fwd %.1f us, dgrad %.1f us, dgrad+BN-backward fusion %.1f us' % (v, tot[v]['fwd'], tot[v]['dgrad'], tot[v]['dgbn']))
