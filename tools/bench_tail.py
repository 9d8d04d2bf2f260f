"""How much of a 3x3 layer's time is launch quantisation?  Same layer, batch sizes that put the tile count
just below / just above a whole number of resident-workgroup rounds (768 slots for the halo kernel)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from convnets_amd import ops, _lib
CL = torch.channels_last
lib = _lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr())
for Cin, Cout, H, batches in [(128, 128, 28, (250, 256, 376, 380)), (64, 64, 56, (251, 256, 282)), (256, 256, 14, (250, 256, 334, 500)),
                              (512, 512, 7, (256, 500, 1000, 2000))]:
    for N in batches:
        x = torch.randn(N, Cin, H, H, device='cuda', dtype=torch.bfloat16).contiguous(memory_format=CL)
        w = torch.randn(Cout, Cin, 3, 3, device='cuda') * 0.05
        desc = ops.make_desc(x.shape, w.shape, (1, 1), (1, 1), _lib.CNK_BF16)
        wf, _ = ops.PackedWeights().get(w, desc, False)
        y = torch.empty(N, Cout, H, H, device='cuda', dtype=torch.bfloat16).contiguous(memory_format=CL)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        f = lambda: lib.cnk_conv2d_fwd(ctypes.byref(desc), P(x), P(wf), None, P(y), None, None, 0, st)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50
        tiles = -(-N * H * H // 128) * (Cout // (128 if Cout % 128 == 0 and -(-N * H * H // 128) * (Cout // 128) >= 512 else 64))
        print('%4d->%4d @%2d N=%4d tiles=%5d (%.2f x768) %7.1f us %6.0f TF' % (Cin, Cout, H, N, tiles, tiles / 768, us, 2.0 * N * H * H * Cout * Cin * 9 / us / 1e6), flush=True)
