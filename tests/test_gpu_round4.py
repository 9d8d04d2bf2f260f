"""GPU parity (-m gpu), round-4 additions: the backward of conv(1x1) -> BatchNorm2d with the BN apply pass folded into the
conv's own backward GEMMs (csrc/bnfold.hip; reference sites: conv2d_block /root/reference/src/basemodel.py:999-1015 as the
bottleneck's last block uses it, /root/reference/src/resnet.py:172-176; both backward passes run inside
/root/reference/src/basemodel.py:278).

  * cnk_conv2d_wgrad_bnfold forms the gradient w.r.t. the conv output while it stages its tiles: BIT-IDENTICAL to
    cnk_bn_bwd (apply pass) followed by cnk_conv2d_wgrad;
  * cnk_conv2d_dgrad_bnfold (K = [g | conv input], per-step weight image + bias from cnk_conv2d_bnfold_prepare) against an
    fp64 restatement of BN backward + data gradient from the same bf16 operands, next to the unfolded kernel pair, on every
    kernel the dispatch can land on (implicit-GEMM gather 64- / 128-wide, ring GEMM 256x128 / 256x256; launch counters);
  * the same with the fused backward reduction of the BN layer in FRONT of the conv riding in the epilogue;
  * whole-model: a bottleneck ResNet step with the fold on equals the step with the fold off within bf16 rounding, and the
    fold actually ran (ops.FOLD_COUNTS).
"""
import ctypes
import math

import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu
CL = torch.channels_last
CNT_GATHER, CNT_G1_2, CNT_G1_4 = 6, 7, 8
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _lib_and_counter():
    from convnets_amd import _lib
    lib = _lib.load()
    cnt = lib.cnk_debug_counter
    cnt.restype, cnt.argtypes = ctypes.c_int64, [ctypes.c_int]
    return _lib, lib, cnt


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class FoldCase:
    """One conv(1x1, Cin -> Cout) -> BN layer pair in the middle of a backward pass: the conv input a (a ReLU output), the
    conv output y, the masked gradient g w.r.t. the BN output and the partial sums a consumer's epilogue would have left."""

    def __init__(self, N, Cin, Cout, H, W, seed, stride=1):
        from convnets_amd import _lib, ops
        lib = _lib.load()
        gen = torch.Generator().manual_seed(seed)
        dev = 'cuda'
        self.N, self.Cin, self.Cout, self.H, self.W, self.stride = N, Cin, Cout, H, W, stride
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        self.Ho, self.Wo = Ho, Wo
        self.M = M = N * Ho * Wo                     # rows of the conv output / BN tensors
        a = torch.relu(torch.randn(N, Cin, H, W, generator=gen)).bfloat16()
        w = torch.randn(Cout, Cin, 1, 1, generator=gen) / math.sqrt(Cin)
        self.a = a.to(dev).contiguous(memory_format=CL)
        self.w = w.to(dev)
        self.desc = d = ops.make_desc(self.a.shape, self.w.shape, (stride, stride), (0, 0), _lib.CNK_BF16)
        self.w_fwd = torch.empty(lib.cnk_conv2d_packed_bytes(ctypes.byref(d), _lib.PASS_FWD), dtype=torch.uint8, device=dev)
        self.w_dgrad = torch.empty(lib.cnk_conv2d_packed_bytes(ctypes.byref(d), _lib.PASS_DGRAD), dtype=torch.uint8, device=dev)
        _lib.check(lib.cnk_conv2d_pack_weights(ctypes.byref(d), P(self.w), P(self.w_fwd), P(self.w_dgrad), _stream()))
        self.y = torch.empty((N, Cout, Ho, Wo), dtype=torch.bfloat16, device=dev, memory_format=CL)
        _lib.check(lib.cnk_conv2d_fwd(ctypes.byref(d), P(self.a), P(self.w_fwd), None, P(self.y), None, None, 0, _stream()))
        y64 = self.y.permute(0, 2, 3, 1).reshape(M, Cout).double().cpu()
        mean = y64.mean(0)
        var = y64.var(0, unbiased=False)
        invstd = 1.0 / torch.sqrt(var + 1e-5)
        self.gamma = (0.5 + torch.rand(Cout, generator=gen)).to(dev)
        self.mean, self.invstd = mean.float().to(dev), invstd.float().to(dev)
        # masked gradient with a per-channel offset (B1 != 0) and a component along xhat (B2 != 0)
        xhat = ((y64 - mean) * invstd).float()
        g = torch.randn(M, Cout, generator=gen) * 0.5 + 0.1 * torch.randn(Cout, generator=gen) + 0.2 * xhat * torch.randn(Cout, generator=gen)
        g = g * (torch.rand(M, Cout, generator=gen) > 0.4)
        self.g = g.bfloat16().reshape(N, Ho, Wo, Cout).permute(0, 3, 1, 2).to(dev).contiguous(memory_format=CL)
        g64 = self.g.permute(0, 2, 3, 1).reshape(M, Cout).double().cpu()
        m32, i32 = self.mean.double().cpu(), self.invstd.double().cpu()
        S = g64.sum(0)
        Q = (g64 * (y64 - m32)).sum(0) * i32
        self.partial = torch.stack([S, Q], 1).reshape(1, Cout, 2).float().to(dev)
        self.y64, self.g64 = y64, g64

    def coef(self, lib, _lib):
        C, dev = self.Cout, 'cuda'
        dgamma, dbeta = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        coef = torch.empty(3 * C, device=dev)
        nws = lib.cnk_bn_workspace_bytes(self.M, C)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        _lib.check(lib.cnk_bn_bwd_coef(None, None, P(self.partial), 1, P(self.gamma), P(self.mean), P(self.invstd), P(dgamma), P(dbeta), 0,
                                       P(coef), self.M, C, _lib.CNK_BF16, P(ws), nws, _stream()))
        return coef, dgamma, dbeta

    def unfolded_dxconv(self, lib, _lib):
        """cnk_bn_bwd with the external partials: the tensor the fold never writes."""
        C, dev = self.Cout, 'cuda'
        dgamma, dbeta = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        dxc = torch.empty_like(self.g, memory_format=CL)
        nws = lib.cnk_bn_workspace_bytes(self.M, C)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        beta = torch.zeros(C, device=dev)
        _lib.check(lib.cnk_bn_bwd(P(self.g), P(self.y), None, None, P(self.gamma), P(beta), P(self.mean), P(self.invstd), P(dxc), None,
                                  P(dgamma), P(dbeta), 0, P(self.partial), 1, self.M, C, 0, _lib.CNK_BF16, P(ws), nws, _stream()))
        return dxc, dgamma, dbeta

    def fp64_da(self, coef):
        """BN backward + data gradient in fp64 from the bf16 operands and the fp32 coefficients the kernels use."""
        C = self.Cout
        c = coef.double().cpu()
        A, B1, B2 = c[:C], c[C:2 * C], c[2 * C:]
        m, i = self.mean.double().cpu(), self.invstd.double().cpu()
        dxc = A * (self.g64 - B1 - (self.y64 - m) * i * B2)
        da = dxc @ self.w.double().cpu().reshape(C, self.Cin)
        if self.stride > 1:                          # the gradient lands on the sampled pixels, zeros elsewhere
            full = torch.zeros(self.N, self.H, self.W, self.Cin, dtype=torch.float64)
            full[:, ::self.stride, ::self.stride, :] = da.reshape(self.N, self.Ho, self.Wo, self.Cin)
            da = full.reshape(-1, self.Cin)
        return dxc, da

    def a_sampled64(self):
        """The conv input at the pixels the conv reads, [M][Cin] fp64."""
        a = self.a.permute(0, 2, 3, 1)[:, ::self.stride, ::self.stride, :]
        return a.reshape(self.M, self.Cin).double().cpu()


# (N, Cin, Cout, H, W, CNK_GEMM1X1, expected launch counter)
FOLD_CASES = {
    'gather64_n8_64_256_14': (8, 64, 256, 14, 14, '1', CNT_GATHER),          # 64-wide tiles, K = 256 + 64, ragged last tile (M = 1568)
    'gather64_n16_128_512_9x13': (16, 128, 512, 9, 13, '1', CNT_GATHER),     # odd image, Nout = 128 below the 128-wide rule
    'gather128_n96_128_512_28': (96, 128, 512, 28, 28, '0', CNT_GATHER),     # 588 tiles of 128 x 128, K = 512 + 128
    'ring2_n64_256_1024_14': (64, 256, 1024, 14, 14, '1', CNT_G1_2),         # the rule's choice at this size: 256 x 128 tiles
    'ring4_n64_256_1024_14': (64, 256, 1024, 14, 14, '4', CNT_G1_4),         # 256 x 256 tiles forced
    'ring2_n40_512_2048_7': (40, 512, 2048, 7, 7, '2', CNT_G1_2),            # 7x7, K = 2048 + 512, M = 1960 (ragged)
}


@pytest.mark.parametrize('case', sorted(FOLD_CASES))
def test_bnfold_dgrad_wgrad_vs_fp64_and_unfolded(case):
    from convnets_amd import ops
    _lib, lib, cnt = _lib_and_counter()
    N, Cin, Cout, H, W, g1, counter = FOLD_CASES[case]
    fc = FoldCase(N, Cin, Cout, H, W, seed=4 + len(case))
    d, dev = fc.desc, 'cuda'
    assert lib.cnk_conv2d_bnfold_supported(ctypes.byref(d)) == 1
    coef, dgamma, dbeta = fc.coef(lib, _lib)
    dxc, dgamma_u, dbeta_u = fc.unfolded_dxconv(lib, _lib)
    assert torch.equal(dgamma, dgamma_u) and torch.equal(dbeta, dbeta_u)
    dxc64, da64 = fc.fp64_da(coef)
    assert rel_l2(dxc.permute(0, 2, 3, 1).reshape(fc.M, Cout).double().cpu().numpy(), dxc64.numpy()) < 4e-3

    # ---- weight gradient: bit-identical to the unfolded pair at the same split depth
    nws = lib.cnk_conv2d_workspace_bytes(ctypes.byref(d), _lib.PASS_WGRAD)
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    for share in (1.0, 0.5):
        budget = max(int(nws * share) // 256 * 256, (Cout * Cin * 4 + 255) // 256 * 256)
        dw_u = torch.zeros(Cout, Cin, 1, 1, device=dev)
        dw_f = torch.zeros(Cout, Cin, 1, 1, device=dev)
        _lib.check(lib.cnk_conv2d_wgrad(ctypes.byref(d), P(dxc), P(fc.a), P(dw_u), None, 0, P(ws), budget, _stream()))
        _lib.check(lib.cnk_conv2d_wgrad_bnfold(ctypes.byref(d), P(fc.g), P(fc.y), P(fc.a), P(fc.mean), P(fc.invstd), P(coef), P(dw_f), 0,
                                               P(ws), budget, _stream()))
        assert torch.equal(dw_u, dw_f), 'folded weight gradient differs from bn_bwd + wgrad (share %.1f)' % share
    dw64 = dxc64.t() @ fc.a_sampled64()
    assert rel_l2(dw_f.reshape(Cout, Cin).double().cpu().numpy(), dw64.numpy()) < 4e-3
    # accumulate flag
    dw_acc = torch.ones(Cout, Cin, 1, 1, device=dev)
    _lib.check(lib.cnk_conv2d_wgrad_bnfold(ctypes.byref(d), P(fc.g), P(fc.y), P(fc.a), P(fc.mean), P(fc.invstd), P(coef), P(dw_acc), 1,
                                           P(ws), nws, _stream()))
    dw_one = torch.zeros(Cout, Cin, 1, 1, device=dev)
    _lib.check(lib.cnk_conv2d_wgrad_bnfold(ctypes.byref(d), P(fc.g), P(fc.y), P(fc.a), P(fc.mean), P(fc.invstd), P(coef), P(dw_one), 0,
                                           P(ws), nws, _stream()))
    assert torch.equal(dw_acc, dw_one + 1.0)

    # ---- data gradient
    w_fold = torch.empty(lib.cnk_conv2d_bnfold_image_bytes(ctypes.byref(d)), dtype=torch.uint8, device=dev)
    bias_fold = torch.empty(Cin, device=dev)
    _lib.check(lib.cnk_conv2d_bnfold_prepare(ctypes.byref(d), P(fc.w), P(fc.w_dgrad), P(fc.mean), P(fc.invstd), P(coef), P(w_fold),
                                             P(bias_fold), _stream()))
    # the image against its definition (fp64): [A W ; -W^T diag(E) W], bias
    C = Cout
    c64 = coef.double().cpu()
    A, B1, B2 = c64[:C], c64[C:2 * C], c64[2 * C:]
    m64, i64 = fc.mean.double().cpu(), fc.invstd.double().cpu()
    w64 = fc.w.double().cpu().reshape(C, Cin)
    E = A * i64 * B2
    img = w_fold.view(torch.bfloat16).reshape(Cin, C + Cin).double().cpu()
    assert rel_l2(img[:, :C].numpy(), (w64 * A[:, None]).t().numpy()) < 3e-3
    Mref = w64.t() @ (w64 * E[:, None])
    assert rel_l2(img[:, C:].numpy(), (-Mref).t().numpy()) < 8e-3          # bf16 operands AND bf16 result
    bref = (w64 * (E * m64 - A * B1)[:, None]).sum(0)
    assert rel_l2(bias_fold.double().cpu().numpy(), bref.numpy()) < 1e-5

    with _lib.switches(CNK_GEMM1X1=g1):
        c0 = cnt(counter)
        da_f = torch.empty_like(fc.a, memory_format=CL)
        _lib.check(lib.cnk_conv2d_dgrad_bnfold(ctypes.byref(d), P(fc.g), P(fc.a), P(w_fold), P(bias_fold), P(da_f), None, None, None, _stream()))
        assert cnt(counter) == c0 + 1, 'the folded data gradient did not run on the expected kernel'
        da_u = torch.empty_like(fc.a, memory_format=CL)
        _lib.check(lib.cnk_conv2d_dgrad(ctypes.byref(d), P(dxc), P(fc.w_dgrad), P(da_u), None, None, None, None, 0, _stream()))
    torch.cuda.synchronize()
    ref = da64.numpy()
    e_f = rel_l2(da_f.permute(0, 2, 3, 1).reshape(-1, Cin).double().cpu().numpy(), ref)
    e_u = rel_l2(da_u.permute(0, 2, 3, 1).reshape(-1, Cin).double().cpu().numpy(), ref)
    print('%s: folded %.2e unfolded %.2e (vs fp64)' % (case, e_f, e_u))
    assert e_f < 4e-3 and e_u < 4e-3


@pytest.mark.parametrize('case', ['gather64_n8_64_256_14', 'gather128_n96_128_512_28', 'ring2_n64_256_1024_14'])
def test_bnfold_dgrad_with_fused_bn_backward_in_front(case):
    """The folded data gradient also carries, in its epilogue, the backward reduction of the BN(+ReLU) layer that produced the
    conv input (cnk_bn_bwd_fuse): masked output and the two partial sums against the unfolded launch with the same record."""
    _lib, lib, cnt = _lib_and_counter()
    N, Cin, Cout, H, W, g1, counter = FOLD_CASES[case]
    fc = FoldCase(N, Cin, Cout, H, W, seed=11)
    d, dev = fc.desc, 'cuda'
    coef, _, _ = fc.coef(lib, _lib)
    dxc, _, _ = fc.unfolded_dxconv(lib, _lib)
    w_fold = torch.empty(lib.cnk_conv2d_bnfold_image_bytes(ctypes.byref(d)), dtype=torch.uint8, device=dev)
    bias_fold = torch.empty(Cin, device=dev)
    _lib.check(lib.cnk_conv2d_bnfold_prepare(ctypes.byref(d), P(fc.w), P(fc.w_dgrad), P(fc.mean), P(fc.invstd), P(coef), P(w_fold),
                                             P(bias_fold), _stream()))
    # the BN layer in front: a = relu(bn(u)); mask re-derived from u (no residual)
    gen = torch.Generator().manual_seed(5)
    u = torch.randn(N, Cin, H, W, generator=gen).bfloat16().to(dev).contiguous(memory_format=CL)
    mean2 = (0.1 * torch.randn(Cin, generator=gen)).to(dev)
    invstd2 = (0.8 + 0.4 * torch.rand(Cin, generator=gen)).to(dev)
    gamma2 = (0.5 + torch.rand(Cin, generator=gen)).to(dev)
    beta2 = (0.1 * torch.randn(Cin, generator=gen)).to(dev)
    with _lib.switches(CNK_GEMM1X1=g1):
        slices = lib.cnk_conv2d_dgrad_bn_slices(ctypes.byref(d))
        assert slices > 0
        outs = []
        for folded in (True, False):
            part = torch.zeros(slices, Cin, 2, device=dev)
            fuse = _lib.BnBwdFuse(u.data_ptr(), mean2.data_ptr(), invstd2.data_ptr(), gamma2.data_ptr(), beta2.data_ptr(), None,
                                  part.data_ptr(), 1)
            da = torch.empty_like(fc.a, memory_format=CL)
            if folded:
                _lib.check(lib.cnk_conv2d_dgrad_bnfold(ctypes.byref(d), P(fc.g), P(fc.a), P(w_fold), P(bias_fold), P(da), None, None,
                                                       ctypes.byref(fuse), _stream()))
            else:
                _lib.check(lib.cnk_conv2d_dgrad(ctypes.byref(d), P(dxc), P(fc.w_dgrad), P(da), None, None, ctypes.byref(fuse), None, 0,
                                                _stream()))
            outs.append((da, part))
    torch.cuda.synchronize()
    (da_f, part_f), (da_u, part_u) = outs
    _, da64 = fc.fp64_da(coef)
    u64 = u.permute(0, 2, 3, 1).reshape(fc.M, Cin).double().cpu()
    sc = (gamma2 * invstd2).double().cpu()
    keep = (u64 * sc + (beta2.double().cpu() - mean2.double().cpu() * sc)) > 0
    ref = (da64 * keep).numpy()
    f = da_f.permute(0, 2, 3, 1).reshape(fc.M, Cin).double().cpu()
    assert rel_l2(f.numpy(), ref) < 4e-3
    assert rel_l2(da_u.permute(0, 2, 3, 1).reshape(fc.M, Cin).double().cpu().numpy(), ref) < 4e-3
    # masked elements are exactly zero in both; the partial sums are those of the tensor each launch wrote
    assert torch.all(f[~keep] == 0)
    S = f.sum(0)
    Q = (f * (u64 - mean2.double().cpu())).sum(0) * invstd2.double().cpu()
    pf = part_f.double().cpu().sum(0)
    scale = max(float(S.abs().max()), float(Q.abs().max()), 1.0)
    assert float((pf[:, 0] - S).abs().max()) / scale < 2e-5
    assert float((pf[:, 1] - Q).abs().max()) / scale < 2e-5


def test_bottleneck_resnet_step_with_and_without_the_fold():
    """ResNet-26 (bottleneck blocks), 64x64, bs 32, bf16: the step with the BN backward of every block's last conv folded
    into that conv equals the unfolded step to bf16 rounding -- forward identical (same loss bits), gradients of the folded
    convs' own weights bit-identical chains aside, all gradients within the bf16 noise two valid roundings show
    (DESIGN.md section 4).  Residual branches damped (x 0.1) as in the well-conditioned fixture, so the bound means something."""
    from convnets_amd import ops
    from oracle import torch_ref as tr
    prev, prev_min = ops.BN_FOLD, ops.BN_FOLD_MIN_ELEMS
    ops.BN_FOLD_MIN_ELEMS = 0              # (the size rule would leave this small net unfolded)
    try:
        res = {}
        for fold in (True, False):
            import convnets_amd as ca
            ops.BN_FOLD = fold
            torch.manual_seed(21)
            s = ca.Settings(kind=26, input_size=(3, 64, 64), num_classes=10, batch_size=32, dropout_rate=0, mixed_precision=True,
                            device='cuda:0')
            model = ca.ResNet(s)
            tr.damp_residual_branches(model, 0.1)
            model = model.cuda()
            model.init_optimizer()
            model.train()
            x, y = tr.synthetic_batch(32, (3, 64, 64), 10, seed=21)
            ops.FOLD_COUNTS['folded'] = 0
            loss, _ = model.train_batch((x, y))
            torch.cuda.synchronize()
            # train_batch has stepped the optimizer; the gradient arena still holds this step's gradients
            grads = {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters() if p.grad is not None}
            res[fold] = (float(loss), grads, ops.FOLD_COUNTS['folded'])
    finally:
        ops.BN_FOLD, ops.BN_FOLD_MIN_ELEMS = prev, prev_min
    (l1, g1, n1), (l0, g0, n0) = res[True], res[False]
    assert n0 == 0 and n1 >= 7, 'the fold ran %d times (ResNet-26 has 8 bottlenecks; the last one has no fused reduction)' % n1
    assert l1 == l0                                  # the forward pass does not change
    num = sum(float(((g1[k] - g0[k]) ** 2).sum()) for k in g0)
    den = sum(float((g0[k] ** 2).sum()) for k in g0)
    err = math.sqrt(num / den)
    print('fold on vs off: whole-gradient L2-rel %.3e' % err)
    assert err < 1e-2                                # measured 3.3e-3 (round 4): three times that
    worst = max((math.sqrt(float(((g1[k] - g0[k]) ** 2).sum()) / max(float((g0[k] ** 2).sum()), 1e-30)), k) for k in g0
                if float((g0[k] ** 2).sum()) > 1e-12)
    print('worst tensor', worst)
    assert worst[0] < 8e-2                           # measured 5.7e-2 (the stem BN bias: a sum over the whole backward chain)


def test_graphed_step_captured_at_zero_learning_rate():
    """a step captured while a warm-up schedule holds lr == 0 used to freeze lr_captured / bc1 at 0 and turn it
    into 0 / 0 (NaN for every parameter) once the rate moved.  Two replays (lr 0, then 0.01) against two eager steps of a
    twin model; a changed weight decay is refused."""
    import convnets_amd as ca
    from convnets_amd.graph import GraphedTrainStep
    from oracle import torch_ref as tr

    def build():
        torch.manual_seed(21)
        s = ca.Settings(kind=18, input_size=(3, 32, 32), num_classes=10, batch_size=16, dropout_rate=0, mixed_precision=True,
                        device='cuda:0')
        m = ca.ResNet(s).cuda()
        m.init_optimizer()
        m.train()
        return m
    x, y = tr.synthetic_batch(16, (3, 32, 32), 10, seed=21)
    x, y = x.cuda(), y.cuda()
    eager, graphed = build(), build()
    for m in (eager, graphed):
        m.optimizer.param_groups[0]['lr'] = 0.0
    step = GraphedTrainStep(graphed, (x, y))
    assert graphed.optimizer.param_groups[0]['lr'] == 0.0           # the nominal capture rate does not leak out
    p0 = graphed.optimizer.flat_p.clone()
    step.replay()
    eager.train_batch((x, y))
    torch.cuda.synchronize()
    assert torch.equal(graphed.optimizer.flat_p, p0), 'a step at lr == 0 moved the parameters'
    for m in (eager, graphed):
        m.optimizer.param_groups[0]['lr'] = 0.01
    step.replay()
    eager.train_batch((x, y))
    torch.cuda.synchronize()
    pg, pe = graphed.optimizer.flat_p, eager.optimizer.flat_p
    assert torch.isfinite(pg).all() and not torch.equal(pg, p0)
    assert torch.allclose(pg, pe, rtol=2e-6, atol=1e-8), float((pg - pe).abs().max())
    assert torch.equal(graphed.optimizer.exp_avg, eager.optimizer.exp_avg)
    graphed.optimizer.param_groups[0]['weight_decay'] = 0.5
    with pytest.raises(RuntimeError):
        step.replay()


@pytest.mark.parametrize('mixed', [False, True])
def test_deferred_shortcut_bn_apply_is_bit_identical(mixed):
    """The projection shortcut's BN (conv -> BN, no activation; /root/reference/src/resnet.py:90-91) is applied INSIDE the
    apply pass of the block's last BN (`output += identity`, resnet.py:191-194; cnk_bn_fwd_train_res_affine) instead of in a
    pass of its own: same loss, same gradients, same running statistics, bit for bit -- bottleneck and basic blocks."""
    import convnets_amd as ca
    from convnets_amd import layers
    from oracle import torch_ref as tr
    prev = layers.DEFER_SHORTCUT_BN
    out = {}
    try:
        for kind in (26, 18):
            for defer in (True, False):
                layers.DEFER_SHORTCUT_BN = defer
                torch.manual_seed(21)
                s = ca.Settings(kind=kind, input_size=(3, 32, 32), num_classes=10, batch_size=16, dropout_rate=0,
                                mixed_precision=mixed, device='cuda:0')
                model = ca.ResNet(s).cuda()
                model.init_optimizer()
                model.train()
                x, y = tr.synthetic_batch(16, (3, 32, 32), 10, seed=21)
                loss, _ = model.train_batch((x, y))
                torch.cuda.synchronize()
                out[(kind, defer)] = (float(loss), model.optimizer.flat_g.clone(), model.optimizer.flat_p.clone(),
                                      [b.clone() for b in model.buffers()])
    finally:
        layers.DEFER_SHORTCUT_BN = prev
    for kind in (26, 18):
        (l1, g1, p1, b1), (l0, g0, p0, b0) = out[(kind, True)], out[(kind, False)]
        assert l1 == l0
        assert torch.equal(g1, g0) and torch.equal(p1, p0)
        assert all(torch.equal(a, b) for a, b in zip(b1, b0))


def test_fold_of_a_projection_shortcut_bn_with_gradient_handover():
    """The projection shortcut of a stage's first block (conv 1x1 -> BN, no activation; /root/reference/src/resnet.py:90-91):
    its BN backward runs the reduction pass only (cnk_bn_bwd_coef with (dy, x)) and the apply pass is folded into the conv's
    backward, whose data gradient also takes the main branch's gradient as addend (ops.GradLink).  Fold on / off on a
    ResNet-26 step: same loss, gradients within bf16 rounding; the stride-1 shortcut of the first stage folds."""
    import convnets_amd as ca
    from convnets_amd import ops
    from oracle import torch_ref as tr
    prev, prev_min = ops.BN_FOLD, ops.BN_FOLD_MIN_ELEMS
    ops.BN_FOLD_MIN_ELEMS = 0
    res = {}
    try:
        for fold in (True, False):
            ops.BN_FOLD = fold
            torch.manual_seed(5)
            s = ca.Settings(kind=26, input_size=(3, 64, 64), num_classes=10, batch_size=32, dropout_rate=0, mixed_precision=True,
                            device='cuda:0')
            model = ca.ResNet(s)
            tr.damp_residual_branches(model, 0.1)
            model = model.cuda()
            model.init_optimizer()
            model.train()
            x, y = tr.synthetic_batch(32, (3, 64, 64), 10, seed=5)
            ops.FOLD_COUNTS['folded'] = 0
            loss, _ = model.train_batch((x, y))
            torch.cuda.synchronize()
            res[fold] = (float(loss), {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters()},
                         ops.FOLD_COUNTS['folded'])
    finally:
        ops.BN_FOLD, ops.BN_FOLD_MIN_ELEMS = prev, prev_min
    (l1, g1, n1), (l0, g0, n0) = res[True], res[False]
    assert n0 == 0 and n1 >= 8, n1                  # 7 last-block convs + the stride-1 projection shortcut of stage 1
    assert l1 == l0
    key = 'features.2.dim_synch.0.weight'           # the folded shortcut conv's own weight gradient
    e = float((g1[key] - g0[key]).norm() / g0[key].norm())
    print('shortcut conv weight gradient, fold on vs off: %.2e' % e)
    assert e < 2e-2
    num = sum(float(((g1[k] - g0[k]) ** 2).sum()) for k in g0)
    den = sum(float((g0[k] ** 2).sum()) for k in g0)
    assert math.sqrt(num / den) < 3e-2


def test_bnfold_random_geometries():
    """Twelve random (N, H, W, Cin, Cout) draws, small and ragged (M below one tile, odd images, one K step): the
    folded data gradient against the fp64 restatement, the folded weight gradient bit-identical to bn_bwd + wgrad."""
    _lib, lib, cnt = _lib_and_counter()
    rng = np.random.RandomState(4)
    dev = 'cuda'
    worst = 0.0
    for it in range(12):
        N = int(rng.randint(1, 9))
        H, W = int(rng.randint(3, 20)), int(rng.randint(3, 20))
        Cin, Cout = 64 * int(rng.randint(1, 5)), 64 * int(rng.randint(1, 9))
        fc = FoldCase(N, Cin, Cout, H, W, seed=100 + it)
        d = fc.desc
        coef, _, _ = fc.coef(lib, _lib)
        dxc, _, _ = fc.unfolded_dxconv(lib, _lib)
        _, da64 = fc.fp64_da(coef)
        w_fold = torch.empty(lib.cnk_conv2d_bnfold_image_bytes(ctypes.byref(d)), dtype=torch.uint8, device=dev)
        bias_fold = torch.empty(Cin, device=dev)
        _lib.check(lib.cnk_conv2d_bnfold_prepare(ctypes.byref(d), P(fc.w), P(fc.w_dgrad), P(fc.mean), P(fc.invstd), P(coef), P(w_fold),
                                                 P(bias_fold), _stream()))
        da = torch.empty_like(fc.a, memory_format=CL)
        _lib.check(lib.cnk_conv2d_dgrad_bnfold(ctypes.byref(d), P(fc.g), P(fc.a), P(w_fold), P(bias_fold), P(da), None,
                                               None, None, _stream()))
        nws = lib.cnk_conv2d_workspace_bytes(ctypes.byref(d), _lib.PASS_WGRAD)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        dw_u, dw_f = torch.zeros(Cout, Cin, 1, 1, device=dev), torch.zeros(Cout, Cin, 1, 1, device=dev)
        _lib.check(lib.cnk_conv2d_wgrad(ctypes.byref(d), P(dxc), P(fc.a), P(dw_u), None, 0, P(ws), nws, _stream()))
        _lib.check(lib.cnk_conv2d_wgrad_bnfold(ctypes.byref(d), P(fc.g), P(fc.y), P(fc.a), P(fc.mean), P(fc.invstd), P(coef), P(dw_f), 0, P(ws),
                                               nws, _stream()))
        torch.cuda.synchronize()
        e = rel_l2(da.permute(0, 2, 3, 1).reshape(-1, Cin).double().cpu().numpy(), da64.numpy())
        worst = max(worst, e)
        assert e < 5e-3, (it, N, H, W, Cin, Cout, e)
        assert torch.equal(dw_u, dw_f), (it, N, H, W, Cin, Cout)
    print('worst folded data gradient vs fp64 over 12 draws: %.2e' % worst)
