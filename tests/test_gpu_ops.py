"""GPU parity tests (run with -m gpu on the MI355X box): every hot-path operator, called through
the C ABI (convnets_amd.ops -> ctypes -> libcnk.so), against
  (1) the golden vectors produced from the reference's own factory-built layers (tests/golden/ops.npz),
  (2) the oracle (oracle/cpu_ops.c / torch-CPU) on seeded inputs covering edge tiles, ragged sizes,
      strides and every kernel family (MFMA implicit-GEMM, MFMA wgrad, direct).
Tolerances: fp32 path L2-rel <= 1e-5 (wgrad 2e-5: long fp32 reductions);
bf16-operand path L2-rel <= 1e-2 against the fp32 oracle.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), 'golden')
OPS = np.load(os.path.join(G, 'ops.npz'))
CL = torch.channels_last
TOL = {torch.float32: 1e-5, torch.bfloat16: 1e-2}
TOLW = {torch.float32: 2e-5, torch.bfloat16: 1e-2}
DTYPES = [torch.float32, torch.bfloat16]


def dev(a, dtype=torch.float32):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda().to(dtype)
    return t.contiguous(memory_format=CL) if t.dim() == 4 else t


def host(t):
    return t.detach().float().cpu().contiguous().numpy()


def _cases(prefix):
    return sorted({k.split('/')[1] for k in OPS.files if k.startswith(prefix + '/')})


def run_conv(x, w, b, dy, stride, pad, dtype):
    from convnets_amd import ops
    xt = dev(x, dtype).requires_grad_(True)
    wt = dev(w).contiguous().requires_grad_(True)
    bt = dev(b).requires_grad_(True) if b is not None else None
    y = ops.Conv2dFn.apply(xt, wt, bt, (stride, stride), (pad, pad), ops.PackedWeights())
    y.backward(dev(dy, dtype))
    torch.cuda.synchronize()
    return host(y), host(xt.grad), host(wt.grad), (host(bt.grad) if bt is not None else None)


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('name', _cases('conv'))
def test_conv_golden(name, dtype):
    g = lambda k: OPS[f'conv/{name}/{k}']
    N, Ci, H, W, Co, k, st, p, has_b = g('cfg')
    y, dx, dw, db = run_conv(g('x'), g('w'), g('b') if has_b else None, g('dy'), int(st), int(p), dtype)
    assert rel_l2(y, g('y')) < TOL[dtype], 'fwd'
    assert rel_l2(dx, g('dx')) < TOL[dtype], 'dgrad'
    assert rel_l2(dw, g('dw')) < TOLW[dtype], 'wgrad'
    if has_b:
        assert rel_l2(db, g('db')) < TOLW[dtype], 'dbias'


# (N, Cin, H, W, Cout, k, stride, pad): ragged M (not a tile multiple), multi-tile N, both strides,
# 5x5, every (BMW,BNW) wgrad tile, direct-path shapes, 1-pixel images
RANDOM_CONVS = [
    (3, 64, 17, 19, 128, 3, 1, 1), (2, 128, 9, 9, 64, 1, 2, 0), (5, 64, 7, 7, 64, 3, 2, 1),
    (2, 256, 14, 14, 256, 3, 1, 1), (2, 128, 28, 28, 512, 1, 1, 0), (1, 512, 7, 7, 2048, 1, 1, 0),
    (2, 64, 13, 13, 192, 5, 1, 2), (4, 192, 6, 6, 384, 3, 1, 1), (1, 64, 1, 1, 64, 1, 1, 0),
    (2, 64, 56, 56, 64, 3, 1, 1), (1, 1024, 14, 14, 2048, 1, 2, 0), (2, 3, 37, 41, 64, 7, 2, 3),
    (3, 6, 10, 10, 16, 5, 1, 0), (2, 32, 9, 9, 64, 5, 2, 1), (2, 16, 8, 8, 24, 3, 1, 1),
    (1, 128, 5, 5, 128, 3, 2, 1), (2, 64, 8, 8, 64, 3, 1, 0),
    # row-patch first-layer kernels: ResNet stem (real width), VGG/AlexNet/LeNet-style first layers
    (2, 3, 224, 224, 64, 7, 2, 3), (2, 3, 40, 40, 64, 7, 2, 3), (3, 3, 24, 32, 64, 3, 1, 1),
    (1, 3, 64, 64, 64, 11, 4, 2), (2, 1, 32, 32, 16, 5, 1, 2), (2, 4, 18, 16, 48, 3, 2, 1),
]


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('cfg', RANDOM_CONVS, ids=lambda c: 'x'.join(map(str, c)))
def test_conv_vs_oracle(cfg, dtype):
    N, Ci, H, W, Co, k, st, p = cfg
    g = torch.Generator().manual_seed(hash(cfg) % (2 ** 31))
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) * (2.0 / (Ci * k * k)) ** 0.5
    if dtype == torch.bfloat16:      # compare on identical (bf16-representable) operands
        x, w = x.bfloat16().float(), w
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    yr = F.conv2d(xr.double(), wr.double(), stride=st, padding=p)
    dy = torch.randn(yr.shape, generator=g)
    if dtype == torch.bfloat16:
        dy = dy.bfloat16().float()
    yr.backward(dy.double())
    y, dx, dw, _ = run_conv(x.numpy(), w.numpy(), None, dy.numpy(), st, p, dtype)
    assert rel_l2(y, yr.detach().numpy()) < TOL[dtype], 'fwd'
    assert rel_l2(dx, xr.grad.numpy()) < TOL[dtype], 'dgrad'
    assert rel_l2(dw, wr.grad.numpy()) < TOLW[dtype], 'wgrad'


def test_conv_small_case_vs_c_oracle():
    from oracle import cpu_ops as orc
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 64, 6, 5)).astype(np.float32)
    w = (rng.standard_normal((64, 64, 3, 3)) * 0.05).astype(np.float32)
    dy = rng.standard_normal((2, 64, 3, 3)).astype(np.float32)
    y, dx, dw, _ = run_conv(x, w, None, dy, 2, 1, torch.float32)
    assert rel_l2(y, orc.conv2d_fwd(x, w, None, 2, 1)) < 1e-5
    assert rel_l2(dx, orc.conv2d_dgrad(dy, w, (6, 5), 2, 1)) < 1e-5
    assert rel_l2(dw, orc.conv2d_wgrad(dy, x, 3, 2, 1)) < 2e-5


def run_bn(x, gamma, beta, res, dy, relu, dtype, rm=None, rv=None):
    from convnets_amd import ops
    xt = dev(x, dtype).requires_grad_(True)
    gt, bt = dev(gamma).requires_grad_(True), dev(beta).requires_grad_(True)
    rt = dev(res, dtype).requires_grad_(True) if res is not None else None
    C = x.shape[1]
    rm = torch.zeros(C, device='cuda') if rm is None else rm
    rv = torch.ones(C, device='cuda') if rv is None else rv
    y = ops.BatchNormActFn.apply(xt, gt, bt, rt, rm, rv, 1e-5, 0.1, relu)
    y.backward(dev(dy, dtype))
    torch.cuda.synchronize()
    return (host(y), host(xt.grad), host(gt.grad), host(bt.grad),
            host(rt.grad) if rt is not None else None, host(rm), host(rv))


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('shape', [(4, 64, 10, 10), (2, 256, 7, 7), (3, 2048, 2, 2), (2, 192, 5, 5),
                                   (2, 6, 9, 9), (1, 16, 1, 3), (8, 64, 33, 31), (2, 128, 40, 40)],
                         ids=lambda s: 'x'.join(map(str, s)))
@pytest.mark.parametrize('mode', ['plain', 'relu', 'add_relu'])
def test_batchnorm_vs_oracle(shape, mode, dtype):
    g = torch.Generator().manual_seed(sum(shape) + len(mode))
    x = torch.randn(shape, generator=g) * 2 + 0.5
    res = torch.randn(shape, generator=g) if mode == 'add_relu' else None
    dy = torch.randn(shape, generator=g)
    gamma, beta = torch.rand(shape[1], generator=g) + 0.5, torch.randn(shape[1], generator=g) * 0.3
    if dtype == torch.bfloat16:
        x, dy = x.bfloat16().float(), dy.bfloat16().float()
        res = res.bfloat16().float() if res is not None else None
    xr = x.clone().double().requires_grad_(True)
    gr, br = gamma.clone().double().requires_grad_(True), beta.clone().double().requires_grad_(True)
    rr = res.clone().double().requires_grad_(True) if res is not None else None
    rm, rv = torch.zeros(shape[1], dtype=torch.float64), torch.ones(shape[1], dtype=torch.float64)
    z = F.batch_norm(xr, rm, rv, gr, br, True, 0.1, 1e-5)
    if rr is not None:
        z = z + rr
    yr = torch.relu(z) if mode != 'plain' else z
    yr.backward(dy.double())
    y, dx, dg, db, dres, rm2, rv2 = run_bn(x.numpy(), gamma.numpy(), beta.numpy(),
                                           None if res is None else res.numpy(), dy.numpy(),
                                           mode != 'plain', dtype)
    t = 2e-5 if dtype == torch.float32 else 1e-2
    assert rel_l2(y, yr.detach().numpy()) < t, 'y'
    assert rel_l2(rm2, rm.numpy()) < 1e-5 and rel_l2(rv2, rv.numpy()) < 1e-5, 'running stats'
    assert rel_l2(dx, xr.grad.numpy()) < (1e-4 if dtype == torch.float32 else 2e-2), 'dx'
    assert rel_l2(dg, gr.grad.numpy()) < (5e-5 if dtype == torch.float32 else 2e-2), 'dgamma'
    assert rel_l2(db, br.grad.numpy()) < (5e-5 if dtype == torch.float32 else 2e-2), 'dbeta'
    if res is not None:
        assert rel_l2(dres, rr.grad.numpy()) < t, 'dres'


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('conv', [(3, 1, 1), (3, 2, 1), (1, 1, 0), (1, 2, 0)], ids=['k3s1', 'k3s2', 'k1s1', 'k1s2'])
@pytest.mark.parametrize('mode', ['plain', 'relu', 'add_relu'])
def test_bn_backward_fused_into_conv_dgrad(mode, conv, dtype):
    """BN(+ReLU) -> conv: the conv's dgrad epilogue masks the gradient and reduces it for the BN
    backward (cnk_bn_bwd_fuse).  Checked against torch fp64 autograd and against the unfused path."""
    from convnets_amd import layers, ops
    k, s, p = conv
    N, C, H, W, Co = 3, 64, 13, 11, 128
    g = torch.Generator().manual_seed(7 * k + s + len(mode))
    x = torch.randn(N, C, H, W, generator=g) * 1.5 + 0.3
    res = torch.randn(N, C, H, W, generator=g) if mode == 'add_relu' else None
    w = torch.randn(Co, C, k, k, generator=g) * 0.1
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
        res = res.bfloat16().float() if res is not None else None
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dz = torch.randn(N, Co, Ho, Wo, generator=g)
    if dtype == torch.bfloat16:
        dz = dz.bfloat16().float()

    def run(fuse):
        ops.BN_BWD_FUSE = fuse
        try:
            bn = layers.BatchNorm2d(C).cuda().train()
            cv = layers.Conv2d(C, Co, kernel_size=k, stride=s, padding=p, bias=False).cuda()
            with torch.no_grad():
                bn.weight.copy_(gamma); bn.bias.copy_(beta); cv.weight.copy_(w)
            xt = dev(x.numpy(), dtype).requires_grad_(True)
            rt = dev(res.numpy(), dtype).requires_grad_(True) if res is not None else None
            before = dict(ops.FUSE_COUNTS)
            y = bn(xt, residual=rt, relu=mode != 'plain')
            z = cv(y)
            z.backward(dev(dz.numpy(), dtype))
            torch.cuda.synchronize()
            hits = ops.FUSE_COUNTS['bn_bwd_fused'] - before['bn_bwd_fused']
            return (host(xt.grad), host(bn.weight.grad), host(bn.bias.grad), host(cv.weight.grad),
                    host(rt.grad) if rt is not None else None, hits)
        finally:
            ops.BN_BWD_FUSE = True

    fz, un = run(True), run(False)
    assert fz[5] == 1 and un[5] == 0, 'fused path must run exactly when enabled'
    xr = x.double().requires_grad_(True)
    rr = res.double().requires_grad_(True) if res is not None else None
    gr, br, wr = gamma.double().requires_grad_(True), beta.double().requires_grad_(True), w.double().requires_grad_(True)
    a = F.batch_norm(xr, None, None, gr, br, True, 0.1, 1e-5)
    if rr is not None:
        a = a + rr
    if mode != 'plain':
        a = torch.relu(a)
    if dtype == torch.bfloat16:
        a = a + (a.detach().float().bfloat16().double() - a.detach())     # the conv sees the bf16-rounded activation
    F.conv2d(a, wr, None, s, p).backward(dz.double())
    ref = (xr.grad.numpy(), gr.grad.numpy(), br.grad.numpy(), wr.grad.numpy(), rr.grad.numpy() if rr is not None else None)
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    for name, got, other, want in zip(('dx', 'dgamma', 'dbeta', 'dw', 'dres'), fz, un, ref):
        if want is None:
            continue
        assert rel_l2(got, want) < tol, name + ' vs fp64'
        # same masked gradient, same rounding; only the fp32 summation order of the two sums differs
        assert rel_l2(got, other) < (1e-5 if dtype == torch.float32 else 2e-3), name + ' fused vs unfused'


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('name', _cases('block'))
def test_conv_block_golden(name, dtype):
    """conv -> BN -> ReLU exactly as the reference's conv2d_block builds it, train and eval mode."""
    import convnets_amd as ca
    g = lambda k: OPS[f'block/{name}/{k}']
    N, Ci, H, W, Co, k, st, p, act = [int(v) for v in g('cfg')]
    s = ca.Settings(kind=0, input_size=(Ci, H, W), num_classes=10, mixed_precision=dtype == torch.bfloat16)
    net = ca.MultiClassBaseModel(s)
    blk = net.conv2d_block(num_filters=Co, kernel_size=k, stride=st, padding=p, activation=bool(act)).cuda()
    assert (net.in_channels, net.height, net.width) == tuple(g('y').shape[1:])
    with torch.no_grad():
        blk[0].weight.copy_(dev(g('w')))
        blk[1].weight.copy_(dev(g('gamma')))
        blk[1].bias.copy_(dev(g('beta')))
    blk.train()
    x = dev(g('x'), dtype).requires_grad_(True)
    y = blk(x)
    y.backward(dev(g('dy'), dtype))
    torch.cuda.synchronize()
    # fp32: tight against the reference's vectors.  bf16: loose against them (rounding flips ReLU
    # masks of near-zero activations, which BN's backward amplifies: ~4e-2 on this 400-sample batch)
    # and TIGHT against a torch-CPU emulation that rounds to bf16 at the same points as the kernels.
    t, tg = (2e-5, 1e-4) if dtype == torch.float32 else (1.5e-2, 8e-2)
    assert rel_l2(host(y), g('y')) < t
    assert rel_l2(host(blk[1].running_mean), g('running_mean')) < t
    assert rel_l2(host(blk[1].running_var), g('running_var')) < t
    assert int(blk[1].num_batches_tracked) == 1
    assert rel_l2(host(blk[0].weight.grad), g('dw')) < tg
    assert rel_l2(host(blk[1].weight.grad), g('dgamma')) < tg
    assert rel_l2(host(blk[1].bias.grad), g('dbeta')) < tg
    if name != 'blkstem':
        assert rel_l2(host(x.grad), g('dx')) < tg
    if dtype == torch.bfloat16:
        r = lambda v: v.bfloat16().float()
        tn = lambda k_: torch.from_numpy(g(k_))
        xe = r(tn('x'))
        we = r(tn('w'))          # implicit-GEMM and small-C paths both take bf16 weight operands
        c = r(F.conv2d(xe, we, stride=st, padding=p)).requires_grad_(True)
        ge, be = tn('gamma').clone().requires_grad_(True), tn('beta').clone().requires_grad_(True)
        z = F.batch_norm(c, None, None, ge, be, True, 0.1, 1e-5)
        ye = r(torch.relu(z) if act else z)
        ye.backward(r(tn('dy')))        # mask from the rounded output, like the kernel
        dc = r(c.grad)
        dwe = torch.nn.grad.conv2d_weight(xe, we.shape, dc, stride=st, padding=p)
        assert rel_l2(host(y), ye.detach().numpy()) < 4e-3
        assert rel_l2(host(blk[0].weight.grad), dwe.numpy()) < 4e-3
        assert rel_l2(host(blk[1].weight.grad), ge.grad.numpy()) < 4e-3
        if name != 'blkstem':
            dxe = r(torch.nn.grad.conv2d_input(xe.shape, we, dc, stride=st, padding=p))
            assert rel_l2(host(x.grad), dxe.numpy()) < 4e-3
    blk.eval()
    with torch.no_grad():
        ye = blk(dev(g('x'), dtype))
    assert rel_l2(host(ye), g('y_eval')) < t


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('name', _cases('pool'))
def test_maxpool_golden(name, dtype):
    from convnets_amd import ops
    g = lambda k: OPS[f'pool/{name}/{k}']
    N, C, H, W, k, st, p = [int(v) for v in g('cfg')]
    x = torch.from_numpy(g('x'))
    dy = torch.from_numpy(g('dy'))
    if dtype == torch.bfloat16:      # bf16 rounding changes which element is the max: re-derive
        x, dy = x.bfloat16().float(), dy.bfloat16().float()
        xr = x.clone().requires_grad_(True)
        yr = F.max_pool2d(xr, k, st, p)
        yr.backward(dy)
        y_ref, dx_ref = yr.detach().numpy(), xr.grad.numpy()
    else:
        y_ref, dx_ref = g('y'), g('dx')
    xt = dev(x.numpy(), dtype).requires_grad_(True)
    y = ops.MaxPool2dFn.apply(xt, (k, k), (st, st), (p, p))
    y.backward(dev(dy.numpy(), dtype))
    torch.cuda.synchronize()
    assert np.array_equal(host(y), y_ref)
    if dtype == torch.float32:
        assert np.array_equal(host(xt.grad), dx_ref)
    else:                            # overlapping windows sum two bf16 grads: one rounding
        assert rel_l2(host(xt.grad), dx_ref) < 5e-3


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
def test_gap_relu_add_nchw(dtype):
    from convnets_amd import ops
    x = dev(OPS['gap/x'], dtype).requires_grad_(True)
    y = ops.GlobalAvgPoolFn.apply(x)
    y.backward(dev(OPS['gap/dy']).contiguous())
    torch.cuda.synchronize()
    t = 1e-6 if dtype == torch.float32 else 1e-2
    assert y.dtype == torch.float32 and rel_l2(host(y), OPS['gap/y']) < t
    assert rel_l2(host(x.grad), OPS['gap/dx']) < t
    g = torch.Generator().manual_seed(3)
    for shape in [(2, 64, 5, 7), (3, 7, 3, 3), (1, 1, 1, 1)]:
        a, b, dy = (torch.randn(shape, generator=g) for _ in range(3))
        if dtype == torch.bfloat16:
            a, b, dy = a.bfloat16().float(), b.bfloat16().float(), dy.bfloat16().float()
        at, bt = dev(a.numpy(), dtype).requires_grad_(True), dev(b.numpy(), dtype).requires_grad_(True)
        r = ops.ReLUFn.apply(at)
        r.backward(dev(dy.numpy(), dtype))
        assert np.array_equal(host(r), torch.relu(a).numpy())
        assert np.array_equal(host(at.grad), (dy * (a > 0)).numpy())
        at.grad = None
        s = ops.AddReLUFn.apply(at, bt)
        s.backward(dev(dy.numpy(), dtype))
        ref = torch.relu(a + b)
        assert rel_l2(host(s), ref.numpy()) < t
        assert np.array_equal(host(at.grad), (dy * (host(s) > 0)).numpy())
        xin = torch.randn(shape, generator=g)
        out = ops.nchw_to_nhwc(xin.cuda(), dtype)
        assert out.is_contiguous(memory_format=CL)
        assert rel_l2(host(out), xin.numpy()) < t


def test_errors_are_loud():
    from convnets_amd import ops
    from convnets_amd._lib import CnkError
    x = torch.randn(2, 64, 8, 8, device='cuda').contiguous(memory_format=CL)
    with pytest.raises(RuntimeError):
        ops.ReLUFn.apply(torch.randn(4))                                   # CPU tensor: no fallback
    with pytest.raises(ValueError):
        ops.Conv2dFn.apply(x, torch.randn(64, 32, 3, 3, device='cuda'), None, (1, 1), (1, 1), ops.PackedWeights())
    with pytest.raises(CnkError):                                          # kernel larger than padded input
        ops.Conv2dFn.apply(x, torch.randn(64, 64, 11, 11, device='cuda'), None, (1, 1), (0, 0), ops.PackedWeights())
    with pytest.raises(TypeError):
        ops.ReLUFn.apply(x.half())


def test_flat_adam_matches_torch():
    """cnk_adam_step over the flat arena vs torch.optim.Adam(lr, weight_decay) (basemodel.py:64-67)
    on identical gradients for several steps."""
    from convnets_amd.optim import FlatAdam
    g = torch.Generator().manual_seed(11)
    shapes = [(64, 3, 7, 7), (64,), (1000, 77), (5,), (1,)]
    ps = [torch.randn(s, generator=g) for s in shapes]
    ref = [p.clone().requires_grad_(True) for p in ps]
    mine = [torch.nn.Parameter(p.clone().cuda()) for p in ps]
    opt_ref = torch.optim.Adam(ref, lr=0.01, weight_decay=1e-4)
    opt = FlatAdam(mine, lr=0.01, weight_decay=1e-4)
    for step in range(5):
        grads = [torch.randn(s, generator=g) * (10.0 ** (step - 2)) for s in shapes]
        opt.zero_grad()
        for p, q, gr in zip(ref, mine, grads):
            p.grad = gr.clone()
            q.grad.add_(gr.cuda())
        opt_ref.step()
        opt.step()
        torch.cuda.synchronize()
        for p, q in zip(ref, mine):
            assert rel_l2(host(q), p.detach().numpy()) < 2e-6, step
    assert all(q.data_ptr() >= opt.flat_p.data_ptr() for q in mine)


@pytest.mark.parametrize('max_norm', [0.5, 1e6], ids=['clips', 'no-op'])
def test_grad_clip_matches_torch(max_norm):
    """gradient_clipping (basemodel.py:131-178): clip_grad_norm_ and clip_grad_value_ over the flat
    gradient arena == torch's per-parameter utilities, including the returned total norm."""
    from convnets_amd.optim import FlatAdam
    g = torch.Generator().manual_seed(3)
    shapes = [(64, 3, 7, 7), (64,), (150,), (1000, 77), (5,), (256, 64, 1, 1)]
    ref = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ref]
    opt = FlatAdam(mine, lr=0.01)
    opt.zero_grad()
    for p, q in zip(ref, mine):
        gr = torch.randn(p.shape, generator=g) * 0.3
        p.grad = gr.clone()
        q.grad.add_(gr.cuda())
    want_norm = torch.nn.utils.clip_grad_norm_(ref, max_norm=max_norm)
    got_norm = opt.clip_grad_norm(max_norm)
    torch.cuda.synchronize()
    assert abs(float(got_norm) - float(want_norm)) <= 2e-6 * float(want_norm)
    for p, q in zip(ref, mine):
        assert rel_l2(host(q.grad), p.grad.numpy()) < 2e-6
    torch.nn.utils.clip_grad_value_(ref, clip_value=0.05)
    opt.clip_grad_value(0.05)
    torch.cuda.synchronize()
    for p, q in zip(ref, mine):
        assert np.array_equal(host(q.grad), p.grad.numpy()) or rel_l2(host(q.grad), p.grad.numpy()) < 2e-6
    assert float(opt.flat_g.abs().max()) <= float(np.float32(0.05))


@pytest.mark.parametrize('shape', [(256, 2048, 1000), (64, 400, 120), (5, 84, 10), (33, 130, 67), (128, 4096, 4096)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_linear_vs_torch(shape):
    """nn.Linear forward / dx / dW / db (csrc/head.hip) against torch fp64 on the host."""
    from convnets_amd import ops
    B, K, N = shape
    g = torch.Generator().manual_seed(B + K + N)
    x, w, b = torch.randn(B, K, generator=g), torch.randn(N, K, generator=g) * 0.05, torch.randn(N, generator=g)
    dy = torch.randn(B, N, generator=g)
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    F.linear(xr, wr, br).backward(dy.double())
    xt, wt, bt = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    y = ops.LinearFn.apply(xt, wt, bt)
    y.backward(dy.cuda())
    torch.cuda.synchronize()
    assert rel_l2(host(y), F.linear(x.double(), w.double(), b.double()).numpy()) < 1e-5
    assert rel_l2(host(xt.grad), xr.grad.numpy()) < 1e-5
    assert rel_l2(host(wt.grad), wr.grad.numpy()) < 2e-5
    assert rel_l2(host(bt.grad), br.grad.numpy()) < 1e-5


@pytest.mark.parametrize('shape', [(256, 1000), (64, 10), (7, 3), (33, 2500)], ids=lambda s: 'x'.join(map(str, s)))
def test_softmax_cross_entropy_sum_vs_torch(shape):
    from convnets_amd import layers
    B, N = shape
    g = torch.Generator().manual_seed(B * N)
    z = torch.randn(B, N, generator=g) * 3
    t = torch.randint(0, N, (B,), generator=g)
    zr = z.double().requires_grad_(True)
    lr = F.cross_entropy(zr, t, reduction='sum')
    (lr * 0.7).backward()
    zt = z.cuda().requires_grad_(True)
    cf = layers.CrossEntropyLoss(reduction='sum')
    loss = cf(zt, t.cuda())
    (loss * 0.7).backward()
    torch.cuda.synchronize()
    assert abs(float(loss) - float(lr)) <= 2e-6 * abs(float(lr))
    assert rel_l2(host(zt.grad), zr.grad.numpy()) < 2e-6
    assert np.array_equal(cf.last_pred.cpu().numpy(), z.argmax(dim=1).numpy())


def test_dropout_statistics_and_backward():
    from convnets_amd import layers
    torch.manual_seed(5)
    d = layers.Dropout(p=0.3).cuda().train()
    x = torch.randn(512, 4096, device='cuda').requires_grad_(True)
    y = d(x)
    y.backward(torch.ones_like(y))
    torch.cuda.synchronize()
    keep = (y != 0)
    rate = float(keep.float().mean())
    assert abs(rate - 0.7) < 3e-3, rate                       # 2M samples: sigma = 3e-4
    assert torch.equal(y[keep], (x.detach() / 0.7)[keep])
    assert torch.equal(x.grad, keep.float() / 0.7)
    y2 = d(x)                                                 # a new call draws a new pattern
    assert not torch.equal(y2 != 0, keep)
    torch.manual_seed(5)
    d.eval()
    assert d(x) is x


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('shape', [(3, 24, 7, 5), (2, 64, 14, 14), (5, 13), (1, 8, 1, 1), (4, 1003)], ids=str)
def test_dropout_feature_maps_and_ragged_sizes(shape, dtype):
    """Feature-map dropout (densenet.py dense layers, squeezenet.py head) natively in bf16 / fp32 on NHWC or plain
    tensors, sizes that are not a multiple of the 8-element mask byte: same keep pattern for both dtypes (the hash
    depends on seed, call counter and memory position only), exact scaling, backward through the bit-packed mask."""
    from convnets_amd import layers, ops
    g = torch.Generator().manual_seed(11)
    x = torch.randn(*shape, generator=g)
    x = x.bfloat16().float() if dtype == torch.bfloat16 else x
    keeps = []
    for dt in (dtype, torch.float32):
        torch.manual_seed(3)
        ops._dropout_calls[0] = 7
        xt = x.cuda().to(dt)
        xt = xt.contiguous(memory_format=CL) if xt.dim() == 4 else xt
        xt.requires_grad_(True)
        y = layers.Dropout(p=0.4).cuda().train()(xt)
        assert y.dtype == dt and y.shape == xt.shape
        dy = torch.randn(*shape, generator=torch.Generator().manual_seed(1)).cuda().to(dt)
        dy = dy.contiguous(memory_format=CL) if dy.dim() == 4 else dy
        y.backward(dy)
        torch.cuda.synchronize()
        want = (xt.detach().float() * (1.0 / (1.0 - 0.4))).to(dt)          # one rounding, as in the kernel
        keep = y != 0
        assert torch.equal(y[keep], want[keep])
        assert torch.equal(xt.grad[keep], (dy.float() * (1.0 / (1.0 - 0.4))).to(dt)[keep]) and float(xt.grad[~keep].abs().sum()) == 0
        keeps.append(keep | (xt.detach() == 0))                                 # (a zero input reads as dropped)
    assert torch.equal(keeps[0], keeps[1])


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('cfg', [(64, 64, 3, 1, 1, True, False), (64, 128, 3, 2, 1, True, False),
                                 (128, 256, 1, 1, 0, True, True), (256, 64, 1, 2, 0, False, False)],
                         ids=['k3s1', 'k3s2', 'k1-res', 'k1s2-noact'])
def test_eval_conv_block_folds_bn_into_conv(cfg, dtype):
    """Eval-mode conv2d_block (+ the residual tail of a ResNet block): BN folded into the conv, ReLU /
    add in the conv epilogue (cnk_conv2d_fwd_act) against the unfused definition in fp64."""
    from convnets_amd import layers
    Cin, Cout, k, s, p, act, with_res = cfg
    g = torch.Generator().manual_seed(Cin + Cout + k)
    blk = layers.ConvBlock(layers.Conv2d(Cin, Cout, kernel_size=k, stride=s, padding=p, bias=False),
                           layers.BatchNorm2d(Cout), *([layers.ReLU(inplace=True)] if act else [])).cuda()
    with torch.no_grad():
        blk[0].weight.copy_(torch.randn(Cout, Cin, k, k, generator=g) * 0.05)
        blk[1].weight.copy_(torch.rand(Cout, generator=g) + 0.5)
        blk[1].bias.copy_(torch.randn(Cout, generator=g) * 0.2)
        blk[1].running_mean.copy_(torch.randn(Cout, generator=g) * 0.3)
        blk[1].running_var.copy_(torch.rand(Cout, generator=g) + 0.5)
    blk.eval()
    x = torch.randn(5, Cin, 14, 14, generator=g)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    Ho = (14 + 2 * p - k) // s + 1
    res = torch.randn(5, Cout, Ho, Ho, generator=g) if with_res else None
    if res is not None and dtype == torch.bfloat16:
        res = res.bfloat16().float()
    with torch.no_grad():
        y = blk(dev(x.numpy(), dtype), residual=None if res is None else dev(res.numpy(), dtype), relu_after_add=with_res)
        assert blk.__dict__.get('_fold') is not None and blk._fold.stamp is not None, 'folded path did not run'
        z = F.conv2d(x.double(), blk[0].weight.detach().double().cpu(), None, s, p)
        z = F.batch_norm(z, blk[1].running_mean.double().cpu(), blk[1].running_var.double().cpu(),
                         blk[1].weight.detach().double().cpu(), blk[1].bias.detach().double().cpu(), False, 0.1, 1e-5)
        if res is not None:
            z = z + res.double()
        if act or with_res:
            z = torch.relu(z)
    assert rel_l2(host(y), z.numpy()) < (1e-5 if dtype == torch.float32 else 1e-2)
    # a training forward moves the running statistics: the folded image must be rebuilt
    blk.train()
    blk(dev(x.numpy(), dtype), residual=None if res is None else dev(res.numpy(), dtype), relu_after_add=with_res)
    blk.eval()
    with torch.no_grad():
        y2 = blk(dev(x.numpy(), dtype), residual=None if res is None else dev(res.numpy(), dtype), relu_after_add=with_res)
    assert rel_l2(host(y2), host(y)) > 1e-4


DWB = np.load(os.path.join(G, 'dwblocks.npz'))


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('cfg', [(64, 3, 1, 1, 14, 14), (128, 3, 2, 1, 13, 11), (32, 5, 1, 2, 9, 9), (8, 3, 2, 0, 7, 8),
                                 (256, 3, 1, 1, 56, 56)], ids=lambda c: 'C%d-k%d-s%d-p%d-%dx%d' % c)
@pytest.mark.parametrize('bias', [False, True], ids=['nobias', 'bias'])
def test_depthwise_conv_vs_torch(cfg, bias, dtype):
    """conv2d_depthwise (basemodel.py:981-997): forward, dx, dw, dbias against torch fp64 on the host."""
    from convnets_amd import layers
    C, k, s, p, H, W = cfg
    N = 3 if H < 50 else 2
    g = torch.Generator().manual_seed(C + k + s + H)
    x, w = torch.randn(N, C, H, W, generator=g), torch.randn(C, 1, k, k, generator=g) * 0.3
    b = torch.randn(C, generator=g) * 0.1 if bias else None
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    conv = layers.Conv2d(C, C, kernel_size=k, stride=s, padding=p, groups=C, bias=bias).cuda()
    with torch.no_grad():
        conv.weight.copy_(w)
        if bias:
            conv.bias.copy_(b)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    br = b.double().requires_grad_(True) if bias else None
    yr = F.conv2d(xr, wr, br, s, p, 1, C)
    dy = torch.randn(yr.shape, generator=g)
    if dtype == torch.bfloat16:
        dy = dy.bfloat16().float()
    yr.backward(dy.double())
    xt = dev(x.numpy(), dtype).requires_grad_(True)
    y = conv(xt)
    y.backward(dev(dy.numpy(), dtype))
    torch.cuda.synchronize()
    t = 1e-5 if dtype == torch.float32 else 1e-2
    assert rel_l2(host(y), yr.detach().numpy()) < t
    assert rel_l2(host(xt.grad), xr.grad.numpy()) < t
    assert rel_l2(host(conv.weight.grad), wr.grad.numpy()) < (2e-5 if dtype == torch.float32 else 1e-2)
    if bias:
        assert rel_l2(host(conv.bias.grad), br.grad.numpy()) < t


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('name', sorted({k.split('/')[0] for k in DWB.files if '/' in k}))
def test_depthwise_separable_block_golden(name, dtype):
    """Conv2dBlockDW (mobilenet_v1.py:92-125) built by this package's factory reproduces the reference
    block's output, gradients and running statistics (tests/golden/dwblocks.npz)."""
    import convnets_amd as ca
    from convnets_amd import layers
    C, F_, H, stride, bn = (int(v) for v in DWB[f'{name}/cfg'])
    s = ca.Settings(kind=0, input_size=(C, H, H), num_classes=10, batch_size=4, batch_norm=bool(bn),
                    dropout_rate=0, mixed_precision=dtype == torch.bfloat16)
    net = ca.MultiClassBaseModel(s)
    blk = layers.Conv2dBlockDW(net, num_filters=F_, kernel_size=3, stride=stride, padding=1).cuda()
    with torch.no_grad():
        for k, p in blk.named_parameters():
            p.copy_(torch.from_numpy(DWB[f'{name}/param/{k}']))
    blk.train()
    xt = dev(DWB[f'{name}/x'], dtype).requires_grad_(True)
    y = blk(xt)
    y.backward(dev(DWB[f'{name}/dy'], dtype))
    torch.cuda.synchronize()
    # bf16 vs the fp32 reference is loose by nature: two conv->BN->ReLU stages, rounding flips the ReLU mask
    # of near-zero activations (the tight check is the fp32 one)
    t = (2e-5, 2e-4) if dtype == torch.float32 else (2e-2, 1e-1)
    assert rel_l2(host(y), DWB[f'{name}/y']) < t[0]
    assert rel_l2(host(xt.grad), DWB[f'{name}/dx']) < t[1]
    for k, p in blk.named_parameters():
        assert rel_l2(host(p.grad), DWB[f'{name}/grad/{k}']) < t[1], k
    for k, b in blk.named_buffers():
        if 'running' in k:
            assert rel_l2(host(b), DWB[f'{name}/buf/{k}']) < (1e-5 if dtype == torch.float32 else 1e-2), k


def _random_conv_cases(n, seed):
    rng = np.random.RandomState(seed)
    cases = []
    chans = [1, 3, 6, 16, 32, 48, 64, 96, 128, 192, 256]
    while len(cases) < n:
        cin, cout = int(rng.choice(chans)), int(rng.choice(chans))
        kh, kw = int(rng.randint(1, 6)), int(rng.randint(1, 6))
        sh, sw = int(rng.randint(1, 4)), int(rng.randint(1, 4))
        ph, pw = int(rng.randint(0, kh)), int(rng.randint(0, kw))
        H, W = int(rng.randint(1, 20)), int(rng.randint(1, 20))
        N = int(rng.randint(1, 5))
        if H + 2 * ph < kh or W + 2 * pw < kw or kh * kw > 25:
            continue
        cases.append((N, cin, H, W, cout, kh, kw, sh, sw, ph, pw, bool(rng.randint(0, 2))))
    # narrow outputs behind wide inputs (Inception's 1x1 reduce convs): the data gradient has a 16/24-deep reduction
    cases += [(2, 192, 9, 9, 16, 1, 1, 1, 1, 0, 0, False), (3, 512, 7, 7, 24, 1, 1, 1, 1, 0, 0, True),
              (2, 128, 8, 8, 16, 3, 3, 1, 1, 1, 1, False), (2, 480, 6, 6, 16, 1, 1, 2, 2, 0, 0, False)]
    return cases


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('case', _random_conv_cases(40, 11), ids=lambda c: 'n%d-c%d-%dx%d-o%d-k%dx%d-s%d%d-p%d%d-b%d' % c)
def test_conv_random_shapes_vs_torch(case, dtype):
    """Randomly drawn geometries (non-square kernels / strides / paddings, 1-pixel images, channel counts
    on and off every kernel family's grid) through whichever kernel the dispatcher picks, against torch
    fp64: forward, dx, dw, dbias."""
    from convnets_amd import ops
    N, cin, H, W, cout, kh, kw, sh, sw, ph, pw, bias = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn(cout, cin, kh, kw, generator=g) / np.sqrt(cin * kh * kw)
    b = torch.randn(cout, generator=g) if bias else None
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    br = b.double().requires_grad_(True) if bias else None
    yr = F.conv2d(xr, wr, br, (sh, sw), (ph, pw))
    dy = torch.randn(yr.shape, generator=g)
    if dtype == torch.bfloat16:
        dy = dy.bfloat16().float()
    yr.backward(dy.double())
    xt = dev(x.numpy(), dtype).requires_grad_(True)
    wt = w.cuda().requires_grad_(True)
    bt = b.cuda().requires_grad_(True) if bias else None
    y = ops.Conv2dFn.apply(xt, wt, bt, (sh, sw), (ph, pw), ops.PackedWeights())
    y.backward(dev(dy.numpy(), dtype))
    torch.cuda.synchronize()
    t = 2e-5 if dtype == torch.float32 else 1.5e-2
    assert rel_l2(host(y), yr.detach().numpy()) < t, 'fwd'
    assert rel_l2(host(xt.grad), xr.grad.numpy()) < t, 'dgrad'
    assert rel_l2(host(wt.grad), wr.grad.numpy()) < (5e-5 if dtype == torch.float32 else 1.5e-2), 'wgrad'
    if bias:
        assert rel_l2(host(bt.grad), br.grad.numpy()) < t, 'dbias'


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('cfg', [(128, 32, 3, 1, 1, 14), (256, 32, 3, 2, 1, 13), (64, 2, 3, 1, 1, 9), (512, 32, 3, 1, 1, 7),
                                 (128, 8, 1, 1, 0, 10), (1024, 32, 3, 2, 1, 6)], ids=lambda c: 'C%d-g%d-k%d-s%d-p%d-%d' % c)
def test_grouped_conv_vs_torch(cfg, dtype):
    """Grouped conv (ResNeXt cardinality, resnext.py:82-96) as a block-diagonal conv per 64-channel chunk on the
    implicit-GEMM kernels: forward, dx, dw against torch fp64 (groups= in F.conv2d)."""
    from convnets_amd import layers
    C, groups, k, s, p, H = cfg
    g = torch.Generator().manual_seed(C + groups + k + s)
    N = 3
    x = torch.randn(N, C, H, H, generator=g)
    w = torch.randn(C, C // groups, k, k, generator=g) / np.sqrt(C // groups * k * k)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    conv = layers.Conv2d(C, C, kernel_size=k, stride=s, padding=p, groups=groups, bias=False).cuda()
    with torch.no_grad():
        conv.weight.copy_(w)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, s, p, 1, groups)
    dy = torch.randn(yr.shape, generator=g)
    if dtype == torch.bfloat16:
        dy = dy.bfloat16().float()
    yr.backward(dy.double())
    xt = dev(x.numpy(), dtype).requires_grad_(True)
    y = conv(xt)
    y.backward(dev(dy.numpy(), dtype))
    torch.cuda.synchronize()
    t = 1e-5 if dtype == torch.float32 else 1e-2
    assert rel_l2(host(y), yr.detach().numpy()) < t, 'fwd'
    assert rel_l2(host(xt.grad), xr.grad.numpy()) < t, 'dgrad'
    assert rel_l2(host(conv.weight.grad), wr.grad.numpy()) < (2e-5 if dtype == torch.float32 else 1e-2), 'wgrad'


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('cfg', [(64, 64, 1, 3, 1, 2, 2, 14), (128, 128, 32, 3, 1, 2, 2, 14), (128, 128, 32, 3, 2, 2, 2, 13),
                                 (64, 96, 1, 3, 1, 3, 3, 11), (256, 256, 32, 3, 2, 3, 3, 12), (40, 72, 1, 3, 2, 2, 1, 9)],
                         ids=lambda c: 'Ci%d-Co%d-g%d-k%d-s%d-d%d-p%d-%d' % c)
def test_dilated_conv_vs_torch(cfg, dtype):
    """Dilated (and dilated grouped) conv -- the SK paths, sknet.py:186-198 (dilation=1+i, padding=1+i,
    groups=32): forward, dx, dw against torch fp64."""
    from convnets_amd import layers
    Ci, Co, groups, k, s, dl, p, H = cfg
    g = torch.Generator().manual_seed(Ci + Co + groups + dl + s)
    N = 3
    x = torch.randn(N, Ci, H, H, generator=g)
    w = torch.randn(Co, Ci // groups, k, k, generator=g) / np.sqrt(Ci // groups * k * k)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    conv = layers.Conv2d(Ci, Co, kernel_size=k, stride=s, padding=p, dilation=dl, groups=groups, bias=False).cuda()
    with torch.no_grad():
        conv.weight.copy_(w)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, s, p, dl, groups)
    dy = torch.randn(yr.shape, generator=g)
    if dtype == torch.bfloat16:
        dy = dy.bfloat16().float()
    yr.backward(dy.double())
    xt = dev(x.numpy(), dtype).requires_grad_(True)
    y = conv(xt)
    assert tuple(y.shape) == tuple(yr.shape)
    y.backward(dev(dy.numpy(), dtype))
    torch.cuda.synchronize()
    t = 1e-5 if dtype == torch.float32 else 1e-2
    assert rel_l2(host(y), yr.detach().numpy()) < t, 'fwd'
    assert rel_l2(host(xt.grad), xr.grad.numpy()) < t, 'dgrad'
    assert rel_l2(host(conv.weight.grad), wr.grad.numpy()) < (2e-5 if dtype == torch.float32 else 1e-2), 'wgrad'


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('cfg', [(3, 64, 7, 7, 2), (2, 128, 14, 9, 2), (5, 256, 1, 1, 2), (2, 72, 5, 5, 3), (1, 2048, 3, 3, 2),
                                 (2, 4096, 2, 2, 4), (4, 64, 33, 31, 1)], ids=lambda c: 'N%d-C%d-%dx%d-P%d' % c)
def test_sk_fuse_and_select_vs_torch(cfg, dtype):
    """The two fused passes of the selective-kernel unit (sknet.py:222-237): pooled = GAP(sum_i U_i) and
    out = sum_i U_i * softmax_i(logits), forward and every gradient (maps through BOTH reads, logits) vs torch
    fp64 -- with the SKLink hand-off (one pass writes dU_i) and without it."""
    from convnets_amd import ops
    N, C, H, W, P = cfg
    g = torch.Generator().manual_seed(N * C + H + P)
    maps = [torch.randn(N, C, H, W, generator=g) for _ in range(P)]
    if dtype == torch.bfloat16:
        maps = [m.bfloat16().float() for m in maps]
    mix = torch.randn(C, C, generator=g) / np.sqrt(C)           # stands in for descriptor + attentions
    shift = [torch.randn(C, generator=g) for _ in range(P)]
    dy = torch.randn(N, C, H, W, generator=g)
    if dtype == torch.bfloat16:
        dy = dy.bfloat16().float()

    mr = [m.double().requires_grad_(True) for m in maps]
    st = torch.stack(mr, 1)
    pooled_r = st.sum(1).mean((2, 3))
    lr = [(pooled_r @ mix.double()) * (i + 1) + shift[i].double() for i in range(P)]
    for l in lr:
        l.retain_grad()
    out_r = (st * torch.softmax(torch.stack(lr, 1), 1)[..., None, None]).sum(1)
    out_r.backward(dy.double())

    for linked in (True, False):
        mt = [dev(m.numpy(), dtype).requires_grad_(True) for m in maps]
        link = ops.SKLink() if linked else None
        pooled = ops.SKPoolFn.apply(link, *mt)
        assert pooled.dtype == torch.float32 and tuple(pooled.shape) == (N, C, 1, 1)
        flat = pooled.view(N, C)
        lt = [((flat @ mix.cuda()) * (i + 1) + shift[i].cuda()).view(N, C, 1, 1) for i in range(P)]
        for l in lt:
            l.retain_grad()
        out = ops.SKSelectFn.apply(link, P, *mt, *lt)
        out.backward(dev(dy.numpy(), dtype))
        torch.cuda.synchronize()
        t = 1e-5 if dtype == torch.float32 else 1e-2
        assert rel_l2(pooled.detach().cpu().numpy().reshape(N, C), pooled_r.detach().numpy()) < (1e-5 if dtype == torch.float32 else 1e-6 + 1e-5)
        assert rel_l2(host(out), out_r.detach().numpy()) < t, 'select fwd'
        for i in range(P):
            assert rel_l2(lt[i].grad.cpu().numpy().reshape(N, C), lr[i].grad.numpy()) < (1e-4 if dtype == torch.float32 else 2e-2), 'dlogits'
            assert rel_l2(host(mt[i].grad), mr[i].grad.numpy()) < t, 'dmaps (linked=%s)' % linked
        if linked:
            assert link.stash is None and not link.pool_pending


def test_conv_and_pool_backward_after_layout_copy():
    """An NCHW-contiguous input that requires grad is re-laid to NHWC inside forward (under no_grad): the data
    gradient / pool indices must still be produced (regression: requires_grad was read after the copy)."""
    from convnets_amd import layers
    g = torch.Generator().manual_seed(4)
    conv = layers.Conv2d(64, 128, kernel_size=1, bias=False).cuda()
    pool = layers.MaxPool2d(kernel_size=2, stride=2)
    x = torch.randn(2, 64, 8, 8, generator=g).cuda().requires_grad_(True)        # plain NCHW, fp32
    y = pool(conv(x * 1.0))
    y.sum().backward()
    torch.cuda.synchronize()
    xr = x.detach().cpu().double().requires_grad_(True)
    F.max_pool2d(F.conv2d(xr, conv.weight.detach().cpu().double()), 2, 2).sum().backward()
    assert rel_l2(host(x.grad), xr.grad.numpy()) < 1e-5


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('cfg', [(64, 8, 8, 2, 2, 0), (32, 9, 7, 3, 2, 1), (8, 5, 5, 2, 1, 1), (6, 4, 6, 2, 2, 0)],
                         ids=lambda c: 'C%d-%dx%d-k%d-s%d-p%d' % c)
def test_avgpool_vs_torch(cfg, dtype):
    """nn.AvgPool2d (avgpool2d, basemodel.py:1026-1033): forward and input gradient against torch."""
    from convnets_amd import layers
    C, H, W, k, s, p = cfg
    g = torch.Generator().manual_seed(C + H + k)
    x = torch.randn(3, C, H, W, generator=g)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    xr = x.double().requires_grad_(True)
    yr = F.avg_pool2d(xr, k, s, p)
    dy = torch.randn(yr.shape, generator=g)
    if dtype == torch.bfloat16:
        dy = dy.bfloat16().float()
    yr.backward(dy.double())
    xt = dev(x.numpy(), dtype).requires_grad_(True)
    y = layers.AvgPool2d(kernel_size=k, stride=s, padding=p)(xt)
    y.backward(dev(dy.numpy(), dtype))
    torch.cuda.synchronize()
    t = 1e-6 if dtype == torch.float32 else 5e-3
    assert rel_l2(host(y), yr.detach().numpy()) < t
    assert rel_l2(host(xt.grad), xr.grad.numpy()) < t


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
def test_input_pipeline_vs_torch(dtype):
    """RandomCrop(32, padding=4) + RandomHorizontalFlip + ToTensor + Normalize (mngrdata.py:139-165) with the random
    draws supplied by the caller, against the same transforms written with torch ops on the host."""
    from convnets_amd import ops
    g = torch.Generator().manual_seed(9)
    N, H, W, C, pad, size = 6, 32, 32, 3, 4, 32
    img = torch.randint(0, 256, (N, H, W, C), generator=g, dtype=torch.uint8)
    mean, std = [0.478, 0.472, 0.430], [0.242, 0.238, 0.258]
    yx = torch.stack([torch.randint(0, 2 * pad + 1, (N,), generator=g), torch.randint(0, 2 * pad + 1, (N,), generator=g)], dim=1)
    flip = torch.randint(0, 2, (N,), generator=g).bool()
    out = ops.input_pipeline(img.cuda(), mean, std, dtype, crop_size=size, padding=pad, crop_yx=yx, flip=flip)
    assert out.shape == (N, C, size, size) and out.is_contiguous(memory_format=CL)
    ref = []
    for n in range(N):
        t = F.pad(img[n].permute(2, 0, 1).double() / 255.0, (pad, pad, pad, pad))          # ToTensor on the padded image
        t = t[:, yx[n, 0]:yx[n, 0] + size, yx[n, 1]:yx[n, 1] + size]
        if flip[n]:
            t = t.flip(-1)
        ref.append((t - torch.tensor(mean).double().view(3, 1, 1)) / torch.tensor(std).double().view(3, 1, 1))
    ref = torch.stack(ref)
    assert rel_l2(host(out), ref.numpy()) < (1e-6 if dtype == torch.float32 else 4e-3)
    # plain ToTensor + Normalize (validation transforms, mngrdata.py:176-190)
    out2 = ops.input_pipeline(img.cuda(), mean, std, dtype)
    ref2 = (img.permute(0, 3, 1, 2).double() / 255.0 - torch.tensor(mean).double().view(1, 3, 1, 1)) / torch.tensor(std).double().view(1, 3, 1, 1)
    assert rel_l2(host(out2), ref2.numpy()) < (1e-6 if dtype == torch.float32 else 4e-3)


def _random_bn_cases(n, seed):
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        out.append((int(rng.randint(1, 9)), int(rng.choice([1, 3, 8, 24, 64, 72, 200, 512])), int(rng.randint(1, 15)),
                    int(rng.randint(1, 15)), str(rng.choice(['plain', 'relu', 'add_relu']))))
    return out


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('case', _random_bn_cases(16, 5), ids=lambda c: 'n%d-c%d-%dx%d-%s' % c)
def test_batchnorm_random_shapes_vs_torch(case, dtype):
    """Randomly drawn BN geometries (1-pixel maps, channel counts off the vector width, tiny batches) in the three
    fused modes, against torch fp64."""
    N, C, H, W, mode = case
    if N * H * W < 2:
        N = 2                                                 # torch refuses batch statistics over one value
    g = torch.Generator().manual_seed(N * 1000 + C * 10 + H)
    shape = (N, C, H, W)
    x = torch.randn(shape, generator=g) * 1.5 + 0.2
    res = torch.randn(shape, generator=g) if mode == 'add_relu' else None
    dy = torch.randn(shape, generator=g)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    if dtype == torch.bfloat16:
        x, dy = x.bfloat16().float(), dy.bfloat16().float()
        res = res.bfloat16().float() if res is not None else None
    xr = x.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rr = res.double().requires_grad_(True) if res is not None else None
    z = F.batch_norm(xr, None, None, gr, br, True, 0.1, 1e-5)
    if rr is not None:
        z = z + rr
    yr = torch.relu(z) if mode != 'plain' else z
    yr.backward(dy.double())
    y, dx, dg, db, dres, _, _ = run_bn(x.numpy(), gamma.numpy(), beta.numpy(), None if res is None else res.numpy(),
                                       dy.numpy(), mode != 'plain', dtype)
    t = 5e-5 if dtype == torch.float32 else 1.5e-2
    assert rel_l2(y, yr.detach().numpy()) < t, 'y'
    # tiny reductions (a handful of values per channel) amplify rounding in dx: scale the bound with 1/sqrt(count)
    amp = max(1.0, 8.0 / np.sqrt(N * H * W))
    assert rel_l2(dx, xr.grad.numpy()) < (2e-4 if dtype == torch.float32 else 3e-2) * amp, 'dx'
    assert rel_l2(dg, gr.grad.numpy()) < (1e-4 if dtype == torch.float32 else 3e-2) * amp, 'dgamma'
    assert rel_l2(db, br.grad.numpy()) < (1e-4 if dtype == torch.float32 else 3e-2), 'dbeta'
    if res is not None:
        assert rel_l2(dres, rr.grad.numpy()) < t, 'dres'


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
def test_input_pipeline_with_random_affine_vs_pillow_golden(dtype):
    """The full train transform list of src/mngrdata.py:146-153 -- RandomCrop(32, padding=4), RandomHorizontalFlip,
    RandomAffine(degrees=15, shear=15, scale=(0.75, 1.25)), ToTensor, Normalize -- in one HIP pass.  (1) The affine stage
    alone reproduces Pillow's outputs (tests/golden/affine.npz) bit for bit; (2) the whole list equals the CPU
    restatement (oracle/transforms_ref.py, itself pinned by the same fixture) on random draws."""
    from convnets_amd import ops
    from oracle import transforms_ref as tf
    A = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'affine.npz'))
    for i in range(int(A['n_cases'])):
        img, draw, want = A[f'case{i}/img'], A[f'case{i}/draw'], A[f'case{i}/out']
        out = ops.input_pipeline(torch.from_numpy(img[None]).cuda(), [0.0] * 3, [1.0 / 255.0] * 3, torch.float32,
                                 affine=draw[None])
        got = host(out)[0].transpose(1, 2, 0)                                  # (v/255 - 0) / (1/255) = v
        assert np.array_equal(np.rint(got).astype(np.uint8), want) and np.abs(got - want).max() < 1e-3, i
    rng = np.random.RandomState(4)
    N, pad, size = 8, 4, 32
    imgs = rng.randint(0, 256, (N, 32, 32, 3)).astype(np.uint8)
    mean, std = [0.478, 0.472, 0.430], [0.242, 0.238, 0.258]
    yx = rng.randint(0, 2 * pad + 1, (N, 2))
    flip = rng.randint(0, 2, N).astype(bool)
    aff = np.stack([rng.uniform(-15, 15, N), rng.uniform(0.75, 1.25, N), rng.uniform(-15, 15, N)], axis=1)
    out = ops.input_pipeline(torch.from_numpy(imgs).cuda(), mean, std, dtype, crop_size=size, padding=pad,
                             crop_yx=torch.from_numpy(yx), flip=torch.from_numpy(flip), affine=aff)
    ref = tf.train_transform(imgs, mean, std, size, pad, yx, flip, aff)
    assert out.is_contiguous(memory_format=CL)
    assert np.abs(host(out) - ref).max() < (2e-6 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('shape', [(3, 64, 14, 14), (2, 256, 56, 56), (4, 2048, 7, 7), (2, 72, 9, 11), (3, 20, 5, 5),
                                   (2, 4096, 8, 8), (1, 6, 3, 3)], ids=str)
def test_se_unit_squeeze_and_scale_vs_torch(shape, dtype):
    """The SE unit's two tensor passes (senet.py:240-252): squeeze = AdaptiveAvgPool2d(1) and x * gate.expand_as(x),
    forward and backward (dx, dgate), on the streaming 16-byte-lane kernels for channel counts on the vector grid
    and the scalar kernels otherwise; large and small maps."""
    from convnets_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    N, C, H, W = shape
    x = torch.randn(*shape, generator=g)
    x = x.bfloat16().float() if dtype == torch.bfloat16 else x
    gate = torch.rand(N, C, generator=g)
    dy = torch.randn(*shape, generator=g)
    dy = dy.bfloat16().float() if dtype == torch.bfloat16 else dy
    dp = torch.randn(N, C, 1, 1, generator=g)
    xr, gr = x.double().requires_grad_(True), gate.double().requires_grad_(True)
    yr = xr * gr.view(N, C, 1, 1)
    pr = xr.mean((2, 3), keepdim=True)
    (yr * dy.double()).sum().backward()
    dx_scale, dgate = xr.grad.clone(), gr.grad.clone()
    xr.grad = None
    (pr * dp.double()).sum().backward()
    xt = dev(x.numpy(), dtype).requires_grad_(True)
    gt = gate.cuda().requires_grad_(True)
    y = ops.ChannelScaleFn.apply(xt, gt)
    y.backward(dev(dy.numpy(), dtype))
    torch.cuda.synchronize()
    t = 1e-5 if dtype == torch.float32 else 1e-2
    assert rel_l2(host(y), yr.detach().numpy()) < t
    assert rel_l2(host(xt.grad), dx_scale.numpy()) < t
    assert rel_l2(gt.grad.cpu().numpy(), dgate.numpy()) < (2e-5 if dtype == torch.float32 else 1e-4)   # fp32 sums of exact products
    xt.grad = None
    p = ops.GlobalAvgPoolFn.apply(xt)
    assert p.dtype == torch.float32 and tuple(p.shape) == (N, C, 1, 1)
    p.backward(dp.cuda())
    torch.cuda.synchronize()
    assert rel_l2(p.detach().cpu().numpy(), pr.detach().numpy()) < 2e-6
    assert rel_l2(host(xt.grad), xr.grad.numpy()) < t
