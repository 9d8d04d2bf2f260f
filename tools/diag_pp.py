import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from convnets_amd import _lib, ops
CL = torch.channels_last
if os.environ.get("CNK_LIB_OVERRIDE"): _lib.LIB_PATH = os.environ["CNK_LIB_OVERRIDE"]
lib = _lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
for (N, C, H) in ((128, 256, 14), (128, 512, 7), (256, 512, 7), (64, 256, 14), (16, 256, 14)):
    g = torch.Generator().manual_seed(1)
    M = N * H * H
    dy = torch.randn(N, C, H, H, generator=g).bfloat16().cuda().contiguous(memory_format=CL)
    w = torch.randn(C, C, 3, 3, generator=g) * 0.02
    desc = ops.make_desc(dy.shape, w.shape, (1, 1), (1, 1), _lib.CNK_BF16)
    _, wd = ops.PackedWeights().get(w.cuda(), desc, True)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    bnx = (torch.randn(N, H, H, C, generator=g) * 1.3 + 0.2).bfloat16().cuda()
    bmean, binv = (torch.randn(C, generator=g) * 0.1).cuda(), (torch.rand(C, generator=g) + 0.5).cuda()
    bgam, bbeta = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.3).cuda()
    res = {}
    for name, env in (('pp', dict(CNK_PP3X3='1')), ('halo', dict(CNK_PP3X3='0'))):
        for k_, v_ in env.items(): os.environ[k_] = v_
        try:
            _lib.reload_switches()
        except AttributeError:
            pass
        if True:
            nb = lib.cnk_conv2d_dgrad_bn_slices(ctypes.byref(desc))
            bpart = torch.full((nb, C, 2), float('nan'), device='cuda')
            fuse = _lib.BnBwdFuse(bnx.data_ptr(), bmean.data_ptr(), binv.data_ptr(), bgam.data_ptr(), bbeta.data_ptr(), None, bpart.data_ptr(), 1)
            dx = torch.full_like(dy, float('nan'))
            assert lib.cnk_conv2d_dgrad(ctypes.byref(desc), P(dy), P(wd), P(dx), None, None, ctypes.byref(fuse), None, 0, st) == 0
            torch.cuda.synchronize()
            res[name] = (dx, bpart, nb)
    a, b = res['pp'], res['halo']
    print(N, C, H, 'slices', a[2], b[2], 'dx finite', bool(torch.isfinite(a[0].float()).all()), bool(torch.isfinite(b[0].float()).all()),
          'equal', bool(torch.equal(a[0], b[0])), 'part finite', bool(torch.isfinite(a[1]).all()), 'nan count dx', int((~torch.isfinite(a[0].float())).sum()),
          'nan part', int((~torch.isfinite(a[1])).sum()), 'sum diff', float((a[1].double().sum(0) - b[1].double().sum(0)).abs().max()), flush=True)
    if not torch.equal(a[0], b[0]):
        d = (a[0].float() != b[0].float()).permute(0, 2, 3, 1).reshape(M, C)     # [m][c]
        rows = d.any(1).nonzero().flatten()
        cols = d.any(0).nonzero().flatten()
        print('  mismatching rows', rows.numel(), 'first', rows[:8].tolist(), 'last', rows[-4:].tolist(), 'cols', cols.numel(), cols[:8].tolist(), cols[-4:].tolist())
        tiles = torch.unique(rows // 256)
        print('  row tiles affected', tiles.numel(), tiles[:16].tolist())
        pd = (a[1] != b[1][:0].new_zeros(1)).shape
        pa, pb = a[1].double(), b[1].double()
        # compare partial sums per 128-row group
        pa2 = pa.view(-1, 2, C, 2).sum(1)
        bad = ((pa2 - pb).abs() > 1e-2 * (pb.abs() + 1)).any(2).any(1).nonzero().flatten()
        print('  bad partial slices (128-row units)', bad.numel(), bad[:16].tolist())
        mx = (a[0].float() - b[0].float()).abs().max()
        print('  max abs diff', float(mx), 'dx pp absmax', float(a[0].float().abs().max()), 'halo absmax', float(b[0].float().abs().max()))
