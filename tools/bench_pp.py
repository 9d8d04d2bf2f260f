"""Stride-1 3x3 conv forward (with BN statistics partials) and data gradient on the ResNet-50 shapes, cold operands.
Prints time per launch, TFLOP/s, a SHA-1 of the output bytes and the merged BN statistics, so two builds / settings
(e.g. CNK_PP3X3=0 vs 1: the r01 halo kernels vs the ping-pong core) can be compared bit for bit.
    python tools/bench_pp.py [N] [--check]      (--check: also compare against a torch fp32 conv on the GPU)"""
import ctypes, hashlib, os, sys
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
P = lambda t: ctypes.c_void_p(t.data_ptr())
sha = lambda t: hashlib.sha1(t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()[:12]
tot = {'fwd': 0.0, 'dgrad': 0.0}
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
    nsl = lib.cnk_conv2d_fwd_bn_partial_slices(ctypes.byref(desc))
    part = torch.zeros(max(nsl, 1) * Cout * 3, device='cuda')
    yo = torch.empty_like(ys[0])
    dxo = torch.empty_like(xs[0])
    # correctness artefacts first (fixed operands)
    assert lib.cnk_conv2d_fwd(ctypes.byref(desc), P(xs[0]), P(wf), None, P(yo), P(part), None, 0, st) == 0, lib.cnk_last_error()
    assert lib.cnk_conv2d_dgrad(ctypes.byref(desc), P(ys[0]), P(wd), P(dxo), None, None, None, None, 0, st) == 0, lib.cnk_last_error()
    torch.cuda.synchronize()
    pt = part.view(nsl, Cout, 3).double()
    n, mean, m2 = pt[:, :, 0], pt[:, :, 1], pt[:, :, 2]
    tn = n.sum(0)
    tmean = (n * mean).sum(0) / tn
    tvar = (m2.sum(0) + (n * (mean - tmean) ** 2).sum(0)) / tn
    info = 'y %s dx %s  n %d mean[0] %.6e var[0] %.6e varsum %.8e' % (sha(yo), sha(dxo), int(tn[0]), tmean[0], tvar[0], tvar.sum())
    if CHECK:
        wb = w.bfloat16().float()
        yr = torch.nn.functional.conv2d(xs[0].float(), wb, padding=1)
        dxr = torch.nn.functional.conv_transpose2d(ys[0].float(), wb, padding=1)
        ey = ((yo.float() - yr).norm() / yr.norm()).item()
        ed = ((dxo.float() - dxr).norm() / dxr.norm()).item()
        yb = yo.float()
        em = ((tmean - yb.mean((0, 2, 3)).double()).abs().max() / yb.abs().mean()).item()
        ev = ((tvar - yb.var((0, 2, 3), unbiased=False).double()).abs() / tvar).max().item()
        info += '  relerr y %.2e dx %.2e mean %.2e var %.2e' % (ey, ed, em, ev)
    runs = {'fwd': lambda i: lib.cnk_conv2d_fwd(ctypes.byref(desc), P(xs[i % copies]), P(wf), None, P(ys[i % copies]), P(part), None, 0, st),
            'dgrad': lambda i: lib.cnk_conv2d_dgrad(ctypes.byref(desc), P(ys[i % copies]), P(wd), P(xs[i % copies]), None, None, None, None, 0, st)}
    out = []
    for name, run in runs.items():
        for i in range(2):
            assert run(i) == 0, lib.cnk_last_error()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3 * copies
        e0.record()
        for i in range(reps):
            run(i + 2)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / reps
        tot[name] += us
        fl = 2.0 * N * H * H * Cout * Cin * 9
        out.append('%s %7.1f us %5.0f TF' % (name, us, fl / us / 1e6))
    print('%4d->%4d @%3d   %s   %s' % (Cin, Cout, H, '   '.join(out), info), flush=True)
print('sum: fwd %.1f us, dgrad %.1f us' % (tot['fwd'], tot['dgrad']))
