"""The backward of a bottleneck's last conv block (conv 1x1 f -> 4f, BN) as the step issues it, unfolded against folded
(csrc/bnfold.hip), per ResNet-50 stage at batch N, cold operands (rotating copies), solo kernels in one process:
  unfolded: cnk_bn_bwd (finalise + apply) | cnk_conv2d_dgrad (+ fused BN2 reduction) | cnk_conv2d_wgrad
  folded  : cnk_bn_bwd_coef | cnk_conv2d_bnfold_prepare | cnk_conv2d_dgrad_bnfold (+ fused BN2 reduction) | cnk_conv2d_wgrad_bnfold
    python tools/bench_fold.py [N]"""
import ctypes, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from convnets_amd import _lib, ops
CL = torch.channels_last
lib = _lib.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
LAYERS = [(64, 256, 56, 3), (128, 512, 28, 4), (256, 1024, 14, 6), (512, 2048, 7, 3)]
tot = {}
for Cin, Cout, H, count in LAYERS:
    M = N * H * H
    per = M * (3 * Cout + 3 * Cin) * 2
    copies = max(2, min(8, int(2.0e9 // per) + 1))
    mk = lambda C: [torch.randn(N, C, H, H, device='cuda').bfloat16().contiguous(memory_format=CL) for _ in range(copies)]
    g, y, dxc, a, da, u = mk(Cout), mk(Cout), mk(Cout), mk(Cin), mk(Cin), mk(Cin)
    w = torch.randn(Cout, Cin, 1, 1, device='cuda') * 0.05
    desc = ops.make_desc(a[0].shape, w.shape, (1, 1), (0, 0), _lib.CNK_BF16)
    wf, wd = ops.PackedWeights().get(w, desc, True)
    mean, invstd, gamma, beta = (torch.zeros(Cout, device='cuda'), torch.ones(Cout, device='cuda'), torch.ones(Cout, device='cuda'),
                                 torch.zeros(Cout, device='cuda'))
    mean2, invstd2, gamma2, beta2 = (torch.zeros(Cin, device='cuda'), torch.ones(Cin, device='cuda'), torch.ones(Cin, device='cuda'),
                                     torch.zeros(Cin, device='cuda'))
    nsl_in = 4 * ((M + 127) // 128) // 4                          # the partial list a consumer's epilogue leaves: one record per 128 rows
    part_in = torch.randn(nsl_in, Cout, 2, device='cuda') * 0.01
    dgamma, dbeta = torch.zeros(Cout, device='cuda'), torch.zeros(Cout, device='cuda')
    coef = torch.empty(3 * Cout, device='cuda')
    nbn = lib.cnk_bn_workspace_bytes(M, Cout)
    wsbn = torch.empty(nbn, dtype=torch.uint8, device='cuda')
    nws = lib.cnk_conv2d_workspace_bytes(ctypes.byref(desc), _lib.PASS_WGRAD)
    ws = torch.empty(nws, dtype=torch.uint8, device='cuda')
    dw = torch.zeros(Cout, Cin, 1, 1, device='cuda')
    nb = lib.cnk_conv2d_dgrad_bn_slices(ctypes.byref(desc))
    bpart = torch.zeros(max(nb, 1) * Cin * 2, device='cuda')
    fuses = [_lib.BnBwdFuse(u[i].data_ptr(), mean2.data_ptr(), invstd2.data_ptr(), gamma2.data_ptr(), beta2.data_ptr(), None,
                            bpart.data_ptr(), 1) for i in range(copies)]
    w_fold = torch.empty(lib.cnk_conv2d_bnfold_image_bytes(ctypes.byref(desc)), dtype=torch.uint8, device='cuda')
    bias_fold = torch.empty(Cin, device='cuda')
    D = ctypes.byref(desc)
    runs = {
        'bn_bwd': lambda i: lib.cnk_bn_bwd(P(g[i]), P(y[i]), None, None, P(gamma), P(beta), P(mean), P(invstd), P(dxc[i]), None, P(dgamma), P(dbeta), 0,
                                           P(part_in), nsl_in, M, Cout, 0, _lib.CNK_BF16, P(wsbn), nbn, st()),
        'dgrad': lambda i: lib.cnk_conv2d_dgrad(D, P(dxc[i]), P(wd), P(da[i]), None, None, ctypes.byref(fuses[i]), None, 0, st()),
        'wgrad': lambda i: lib.cnk_conv2d_wgrad(D, P(dxc[i]), P(a[i]), P(dw), None, 0, P(ws), nws, st()),
        'coef': lambda i: lib.cnk_bn_bwd_coef(None, None, P(part_in), nsl_in, P(gamma), P(mean), P(invstd), P(dgamma), P(dbeta), 0, P(coef), M, Cout, _lib.CNK_BF16, P(wsbn), nbn, st()),
        'prep': lambda i: lib.cnk_conv2d_bnfold_prepare(D, P(w), P(wd), P(mean), P(invstd), P(coef), P(w_fold), P(bias_fold), st()),
        'dgrad_f': lambda i: lib.cnk_conv2d_dgrad_bnfold(D, P(g[i]), P(a[i]), P(w_fold), P(bias_fold), P(da[i]), None, None, ctypes.byref(fuses[i]), st()),
        'wgrad_f': lambda i: lib.cnk_conv2d_wgrad_bnfold(D, P(g[i]), P(y[i]), P(a[i]), P(mean), P(invstd), P(coef), P(dw), 0, P(ws), nws, st()),
    }
    if lib.cnk_conv2d_bwd_bnfold_supported(D):
        # round 5: both folded gradients in one pass (conv_fold3.hip); its fused-reduction partials have their own slice count
        n1 = lib.cnk_conv2d_bwd_bnfold_workspace_bytes(D)
        ws1 = torch.empty(n1, dtype=torch.uint8, device='cuda')
        bpart1 = torch.zeros(lib.cnk_conv2d_bwd_bnfold_bn_slices(D) * Cin * 2, device='cuda')
        fuses1 = [_lib.BnBwdFuse(u[i].data_ptr(), mean2.data_ptr(), invstd2.data_ptr(), gamma2.data_ptr(), beta2.data_ptr(), None,
                                 bpart1.data_ptr(), 1) for i in range(copies)]
        runs['one_pass'] = lambda i: lib.cnk_conv2d_bwd_bnfold(D, P(g[i]), P(a[i]), P(w_fold), P(bias_fold), P(da[i]), None, ctypes.byref(fuses1[i]),
                                                              P(w), P(mean), P(invstd), P(coef), P(dw), 0, P(ws1), n1, st())
    res = {}
    for rnd in range(3):
        for name, run in runs.items():
            for i in range(2):
                assert run(i) == 0, lib.cnk_last_error()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 2 * copies
            e0.record()
            for i in range(reps):
                run((i + 2) % copies)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(name, []).append(e0.elapsed_time(e1) * 1000 / reps)
    r = {k: statistics.median(v) for k, v in res.items()}
    unf_main, fold_main = r['bn_bwd'] + r['dgrad'], r['coef'] + r['prep'] + r['dgrad_f']
    print('%4d->%4d @%2d x%d | bn_bwd %6.1f dgrad %6.1f wgrad %6.1f | coef %5.1f prep %5.1f dgrad %6.1f wgrad %6.1f | main stream %6.1f -> %6.1f, side %6.1f -> %6.1f us'
          % (Cin, Cout, H, count, r['bn_bwd'], r['dgrad'], r['wgrad'], r['coef'], r['prep'], r['dgrad_f'], r['wgrad_f'], unf_main, fold_main,
             r['wgrad'], r['wgrad_f']), flush=True)
    if 'one_pass' in r:
        print('           one pass (data gradient + weight gradient, csrc/conv_fold3.hip): %6.1f us  (the two folded launches: %6.1f + %6.1f)'
              % (r['one_pass'], r['dgrad_f'], r['wgrad_f']), flush=True)
    for k, v in (('main_unf', unf_main), ('main_fold', fold_main), ('side_unf', r['wgrad']), ('side_fold', r['wgrad_f'])):
        tot[k] = tot.get(k, 0.0) + v * count
print('per step (count-weighted): main stream %.0f -> %.0f us, side stream %.0f -> %.0f us' % (tot['main_unf'], tot['main_fold'], tot['side_unf'], tot['side_fold']))
