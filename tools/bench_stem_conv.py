"""The first conv of ResNet-50 at the bench shape (256 x 224 x 224 x 3 -> 112 x 112 x 64, 7x7 / 2, bf16): forward with the BN
statistics partials, and the weight gradient.  Cold operands (3 rotating copies), median of 4 rounds.
    python tools/bench_stem_conv.py [N]         CNK_LIB_OVERRIDE=<ablated build> for tools/ablate_rp.sh builds"""
import ctypes, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from convnets_amd import _lib
if os.environ.get('CNK_LIB_OVERRIDE'):
    _lib.LIB_PATH = os.environ['CNK_LIB_OVERRIDE']
from convnets_amd import ops
lib = _lib.load()
CL = torch.channels_last
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
COP = 3
xs = [torch.randn(N, 3, 224, 224, device='cuda').bfloat16().contiguous(memory_format=CL) for _ in range(COP)]
ys = [torch.empty(N, 64, 112, 112, device='cuda', dtype=torch.bfloat16).contiguous(memory_format=CL) for _ in range(COP)]
dys = [torch.randn(N, 64, 112, 112, device='cuda').bfloat16().contiguous(memory_format=CL) for _ in range(COP)]
w = torch.randn(64, 3, 7, 7, device='cuda') * 0.05
desc = ops.make_desc(xs[0].shape, w.shape, (2, 2), (3, 3), _lib.CNK_BF16)
pk = ops.PackedWeights()
wf, _ = pk.get(w, desc, False)
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
nsl = lib.cnk_conv2d_fwd_bn_partial_slices(ctypes.byref(desc))
part = torch.empty(max(nsl, 1), 64, 3, device='cuda')
nws = lib.cnk_conv2d_workspace_bytes(ctypes.byref(desc), _lib.PASS_WGRAD)
ws = torch.empty(nws, dtype=torch.uint8, device='cuda')
dw = torch.empty_like(w)


def fwd(i):
    assert lib.cnk_conv2d_fwd(ctypes.byref(desc), P(xs[i % COP]), P(wf), None, P(ys[i % COP]), P(part) if nsl else None, None, 0, st) == 0, lib.cnk_last_error()


def wgrad(i):
    assert lib.cnk_conv2d_wgrad(ctypes.byref(desc), P(dys[i % COP]), P(xs[i % COP]), P(dw), None, 0, P(ws), nws, st) == 0, lib.cnk_last_error()


for name, fn in (('forward', fwd), ('wgrad', wgrad)):
    t = []
    for rnd in range(4):
        fn(0); fn(1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(12):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        t.append(e0.elapsed_time(e1) * 1000 / 12)
    us = statistics.median(t)
    print('%-8s %7.1f us   %.0f TFLOP/s   (%s)' % (name, us, 2.0 * N * 112 * 112 * 64 * 147 / us / 1e6, ' '.join('%.0f' % v for v in t)))
