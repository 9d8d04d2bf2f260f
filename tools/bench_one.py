"""Runs a few launches of selected conv shapes (for counter collection under rocprofv3)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from convnets_amd import _lib
if os.environ.get('CNK_LIB_OVERRIDE'):
    _lib.LIB_PATH = os.environ['CNK_LIB_OVERRIDE']      # ablated debug build (tools/ablate.sh)
from convnets_amd import ops
CL = torch.channels_last
lib = _lib.load()
N = 256
SHAPES = [(128, 128, 3, 1, 1, 28), (64, 64, 3, 1, 1, 56), (256, 256, 3, 1, 1, 14), (256, 1024, 1, 1, 0, 14), (512, 128, 1, 1, 0, 28)]
if os.environ.get('ONE_SHAPES') == '1x1':
    SHAPES = [(256, 1024, 1, 1, 0, 14), (1024, 256, 1, 1, 0, 14), (1024, 512, 1, 1, 0, 14), (512, 2048, 1, 1, 0, 7), (2048, 512, 1, 1, 0, 7), (512, 128, 1, 1, 0, 28), (128, 512, 1, 1, 0, 28)]
for Cin, Cout, k, s, p, H in SHAPES:
    x = torch.randn(N, Cin, H, H, device='cuda', dtype=torch.bfloat16).contiguous(memory_format=CL)
    w = torch.randn(Cout, Cin, k, k, device='cuda') * 0.05
    desc = ops.make_desc(x.shape, w.shape, (s, s), (p, p), _lib.CNK_BF16)
    pk = ops.PackedWeights()
    wf, wd = pk.get(w, desc, False)
    Ho = (H + 2 * p - k) // s + 1
    y = torch.empty(N, Cout, Ho, Ho, device='cuda', dtype=torch.bfloat16).contiguous(memory_format=CL)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        lib.cnk_conv2d_fwd(ctypes.byref(desc), P(x), P(wf), None, P(y), None, None, 0, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lib.cnk_conv2d_fwd(ctypes.byref(desc), P(x), P(wf), None, P(y), None, None, 0, st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    print('%4d->%4d k%d s%d @%3d  %7.1f us  %6.0f TF' % (Cin, Cout, k, s, H, us, 2.0 * N * Ho * Ho * Cout * Cin * k * k / us / 1e6), flush=True)
