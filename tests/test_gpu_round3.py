"""GPU parity (-m gpu), round-3 additions:
  * the ping-pong 3x3 core (csrc/conv_pp.hip; reference sites src/resnet.py:171, src/vggnet.py:69) per CONFIGURATION
    (256x128 / 512x128 / 256x256 tiles) and per EPILOGUE mode (plain, BN statistics, the full data-gradient epilogue
    with gradient addend + packed ReLU mask + fused BN backward) against torch fp64 on the CPU from the bf16-rounded
    operands, against the tensor the kernel itself wrote (statistics), and bit for bit against the r01 halo kernel
    wherever the K steps are the same 64 channels -- with a launch counter asserting that the forced configuration
    is the one that ran;
  * BASELINE-size (bs 256) unforced checks of the layers whose dispatch lands on those configurations;
  * AlexNet at its BASELINE size (224x224, bs 128, fp32: the 11x11/4 direct kernel at N = 128) as a property step;
  * a batch sweep N in {32 .. 256}: whatever the work-per-CU dispatch rules choose, the results stay the same.
"""
import ctypes
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu
CL = torch.channels_last

# csrc/common.h launch counters
CNT_PP = {0: 0, 1: 1, 2: 2, 3: 3}
CNT_HALO, CNT_HALO_PF, CNT_GATHER, CNT_G1_2, CNT_G1_4, CNT_PP_FULL = 4, 5, 6, 7, 8, 9
PP_WROWS = {0: 64, 1: 64, 2: 128, 3: 128}          # rows of one statistics slice per configuration (PPCfg::WROWS)


def host(t):
    return t.detach().float().cpu().contiguous().numpy()


def _lib_and_counter():
    from convnets_amd import _lib
    lib = _lib.load()
    cnt = lib.cnk_debug_counter
    cnt.restype, cnt.argtypes = ctypes.c_int64, [ctypes.c_int]
    return _lib, lib, cnt


P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _merge_stats(part, M):
    pt = part.double()
    assert torch.isfinite(pt).all()
    n, mean, m2 = pt[:, :, 0], pt[:, :, 1], pt[:, :, 2]
    tn = n.sum(0)
    assert torch.all(tn == M)
    tmean = (n * mean).sum(0) / tn
    tvar = (m2.sum(0) + (n * (mean - tmean) ** 2).sum(0)) / tn
    return tmean, tvar


def _nhwc(t):                       # logical NCHW (any memory format) -> [N, H, W, C] values on the CPU, fp64
    return t.detach().permute(0, 2, 3, 1).double().cpu().contiguous()


# (N, C, H, W): every shape passes pp_pick's fits() for the configurations it is paired with below
PP_SHAPES = {
    'n16_256_14': (16, 256, 14, 14),        # M = 3136 = 12 x 256 + 64: ragged last row tile for every configuration
    'n8_128_28': (8, 128, 28, 28),          # two 64-channel chunks (four 32-channel ones), long rows
    'n48_512_7': (48, 512, 7, 7),           # 7x7: a tile spans five images, eight chunks
    'n16_128_9x13': (16, 128, 9, 13),       # odd, non-square image: row wraps at every 13th pixel
}
PP_FWD_CASES = [(0, 'n16_256_14'), (0, 'n8_128_28'), (0, 'n48_512_7'), (0, 'n16_128_9x13'),
                (2, 'n16_256_14'), (2, 'n8_128_28'), (2, 'n48_512_7'),
                (3, 'n16_256_14'), (3, 'n48_512_7')]


def _operands(shape, seed):
    N, C, H, W = shape
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, C, H, W, generator=g).bfloat16()
    dy = torch.randn(N, C, H, W, generator=g).bfloat16()
    w = torch.randn(C, C, 3, 3, generator=g) * (1.0 / math.sqrt(9 * C))
    return x, dy, w


@pytest.mark.parametrize('cfg,shape', PP_FWD_CASES)
def test_pingpong_forward_per_configuration(cfg, shape):
    """Forward with BN statistics partials (epilogue STATS) and without (PLAIN), configuration forced."""
    _lib, lib, cnt = _lib_and_counter()
    from convnets_amd import ops
    N, C, H, W = PP_SHAPES[shape]
    M = N * H * W
    x, _, w = _operands(PP_SHAPES[shape], 100 + cfg)
    xg = x.cuda().contiguous(memory_format=CL)
    desc = ops.make_desc(xg.shape, w.shape, (1, 1), (1, 1), _lib.CNK_BF16)
    wf, _ = ops.PackedWeights().get(w.cuda(), desc, True)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ref = F.conv2d(x.double(), w.bfloat16().double(), padding=1)
    with _lib.switches(CNK_PP3X3='1', CNK_PP_CFG=str(cfg)):
        nsl = lib.cnk_conv2d_fwd_bn_partial_slices(ctypes.byref(desc))
        assert nsl == (M + PP_WROWS[cfg] - 1) // PP_WROWS[cfg], 'the forced configuration does not take this shape'
        part = torch.full((nsl, C, 3), float('nan'), device='cuda')
        y, y2 = torch.empty_like(xg), torch.empty_like(xg)
        c0 = cnt(CNT_PP[cfg])
        assert lib.cnk_conv2d_fwd(ctypes.byref(desc), P(xg), P(wf), None, P(y), P(part), None, 0, st) == 0, lib.cnk_last_error()
        assert lib.cnk_conv2d_fwd(ctypes.byref(desc), P(xg), P(wf), None, P(y2), None, None, 0, st) == 0, lib.cnk_last_error()
        torch.cuda.synchronize()
        assert cnt(CNT_PP[cfg]) - c0 == 2, 'configuration %d did not run' % cfg
    assert torch.equal(y, y2)                                       # PLAIN and STATS epilogues write the same rows
    assert rel_l2(host(y), ref.float().numpy()) < 4e-3
    tmean, tvar = _merge_stats(part, M)
    yd = y.double()
    assert ((tmean - yd.mean((0, 2, 3))).abs().max() / yd.abs().mean()).item() < 2e-6      # statistics of the tensor it wrote
    assert ((tvar - yd.var((0, 2, 3), unbiased=False)).abs() / tvar).max().item() < 2e-5
    if cfg == 0:
        # 64-channel K steps: same MFMA order as the r01 halo kernel => the same bits
        with _lib.switches(CNK_PP3X3='0'):
            yh = torch.empty_like(xg)
            h0 = cnt(CNT_HALO) + cnt(CNT_HALO_PF)
            assert lib.cnk_conv2d_fwd(ctypes.byref(desc), P(xg), P(wf), None, P(yh), None, None, 0, st) == 0, lib.cnk_last_error()
            torch.cuda.synchronize()
            assert cnt(CNT_HALO) + cnt(CNT_HALO_PF) - h0 == 1
        assert torch.equal(y, yh)


def _dgrad_reference(dy, w, addend, amask_bits, bnx, keep_bits, mean, invstd, padding=1):
    """fp64 restatement of the data-gradient epilogue (conv_pp.hip pp_epilogue FULL == conv_igemm.hip igemm_epilogue):
    the accumulator is rounded to bf16, the (mask-gated) addend is added and the sum rounded again, the BN layer's ReLU
    mask zeroes elements, and the per-channel sums of d' and d' * (x - mean) * invstd are taken over the result."""
    dx = F.conv_transpose2d(dy.double(), w.bfloat16().double(), padding=padding)
    v = _nhwc(dx.float().bfloat16())                                       # [N, H, W, C]
    if addend is not None:
        a = addend.double().cpu()
        if amask_bits is not None:
            a = a * amask_bits
        v = (v + a).float().bfloat16().double()
    if keep_bits is not None:
        v = v * keep_bits
    s1 = v.sum((0, 1, 2))
    s2 = (v * (bnx.double().cpu() - mean.double().cpu())).sum((0, 1, 2)) * invstd.double().cpu() if bnx is not None else None
    return v, s1, s2


def _unpack_bits(mask_bytes, shape_nhwc):
    b = mask_bytes.cpu().numpy()
    bits = np.unpackbits(b[:, None], axis=1, bitorder='little').reshape(shape_nhwc)     # bit e of byte k = channel 8k + e
    return torch.from_numpy(bits.astype(np.float64))


@pytest.mark.parametrize('shape', ['n16_256_14', 'n8_128_28', 'n48_512_7', 'n16_128_9x13'])
@pytest.mark.parametrize('maskmode', ['bits', 'remask'])
def test_pingpong_data_gradient_full_epilogue(shape, maskmode):
    """Configuration 0 (the only one the data gradients use): plain data gradient, and the data gradient as the training
    step issues it -- gradient addend gated by a packed ReLU mask + the first pass of the producing BN layer's backward
    (mask from saved bits or re-derived from x*scale+shift) -- on the ping-pong core, against fp64 and against the halo
    kernel (bit-identical dx, sums to fp32 summation order)."""
    _lib, lib, cnt = _lib_and_counter()
    from convnets_amd import ops
    N, C, H, W = PP_SHAPES[shape]
    M = N * H * W
    x, dy, w = _operands(PP_SHAPES[shape], 7)
    g = torch.Generator().manual_seed(19)
    dyg = dy.cuda().contiguous(memory_format=CL)
    desc = ops.make_desc(dyg.shape, w.shape, (1, 1), (1, 1), _lib.CNK_BF16)
    _, wd = ops.PackedWeights().get(w.cuda(), desc, True)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    bnx = (torch.randn(N, H, W, C, generator=g) * 1.3 + 0.2).bfloat16().cuda()
    addend = torch.randn(N, H, W, C, generator=g).bfloat16().cuda()
    amask = torch.randint(0, 256, (M * C // 8,), generator=g, dtype=torch.uint8).cuda()
    kmask = torch.randint(0, 256, (M * C // 8,), generator=g, dtype=torch.uint8).cuda()
    bmean, binv = (torch.randn(C, generator=g) * 0.1).cuda(), (torch.rand(C, generator=g) + 0.5).cuda()
    bgam, bbeta = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.3).cuda()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    out = {}
    for name, env in (('pp', dict(CNK_PP3X3='1', CNK_PP_CFG='0', CNK_PP_FULL='1')), ('halo', dict(CNK_PP3X3='0'))):
        with _lib.switches(**env):
            nb = lib.cnk_conv2d_dgrad_bn_slices(ctypes.byref(desc))
            assert nb == (M + (64 if name == 'pp' else 128) - 1) // (64 if name == 'pp' else 128)
            bpart = torch.full((nb, C, 2), float('nan'), device='cuda')
            fuse = _lib.BnBwdFuse(bnx.data_ptr(), bmean.data_ptr(), binv.data_ptr(), bgam.data_ptr(), bbeta.data_ptr(),
                                  kmask.data_ptr() if maskmode == 'bits' else None, bpart.data_ptr(), 1)
            dxp, dxf = torch.empty_like(dyg), torch.empty_like(dyg)
            c0, f0 = cnt(CNT_PP[0]), cnt(CNT_PP_FULL)
            assert lib.cnk_conv2d_dgrad(ctypes.byref(desc), P(dyg), P(wd), P(dxp), None, None, None, None, 0, st) == 0, lib.cnk_last_error()
            assert lib.cnk_conv2d_dgrad(ctypes.byref(desc), P(dyg), P(wd), P(dxf), P(addend), P(amask), ctypes.byref(fuse), None, 0, st) == 0, lib.cnk_last_error()
            torch.cuda.synchronize()
            if name == 'pp':
                assert cnt(CNT_PP[0]) - c0 == 2 and cnt(CNT_PP_FULL) - f0 == 1, 'the ping-pong core did not run'
            else:
                assert cnt(CNT_PP[0]) == c0
            assert torch.isfinite(bpart).all()
            out[name] = (dxp, dxf, bpart.double().sum(0))
    # bit-identical between the two kernels
    assert torch.equal(out['pp'][0], out['halo'][0])
    assert torch.equal(out['pp'][1], out['halo'][1])
    assert ((out['pp'][2] - out['halo'][2]).abs().max() / out['halo'][2].abs().max()).item() < 1e-5
    # fp64 reference
    shp = (N, H, W, C)
    abits = _unpack_bits(amask, shp)
    if maskmode == 'bits':
        keep = _unpack_bits(kmask, shp)
    else:
        sc = (bgam * binv).double().cpu()
        sf = bbeta.double().cpu() - bmean.double().cpu() * sc
        # the kernel evaluates x*scale+shift in fp32: elements within rounding of zero may flip; compare around them
        act = bnx.double().cpu() * sc + sf
        keep = (act > 0).double()
        unsure = act.abs() < 1e-5 * (bnx.double().cpu().abs() * sc.abs() + sf.abs() + 1e-30)
        assert unsure.float().mean().item() < 1e-4
    vplain, _, _ = _dgrad_reference(dy, w, None, None, None, None, None, None)
    assert rel_l2(_nhwc(out['pp'][0]).float().numpy(), vplain.float().numpy()) < 4e-3
    v, s1, s2 = _dgrad_reference(dy, w, addend, abits, bnx, keep, bmean, binv)
    assert rel_l2(_nhwc(out['pp'][1]).float().numpy(), v.float().numpy()) < 4e-3
    got = out['pp'][2].cpu()
    assert ((got[:, 0] - s1).abs().max() / s1.abs().max()).item() < 5e-3
    assert ((got[:, 1] - s2).abs().max() / s2.abs().max()).item() < 5e-3
    # the sums must be those of the tensor the kernel wrote (fp32 sums of bf16 values: tight)
    wrote = _nhwc(out['pp'][1])
    assert ((got[:, 0] - wrote.sum((0, 1, 2))).abs().max() / wrote.abs().sum((0, 1, 2)).max()).item() < 1e-5
    q = (wrote * (bnx.double().cpu() - bmean.double().cpu())).sum((0, 1, 2)) * binv.double().cpu()
    assert ((got[:, 1] - q).abs().max() / q.abs().max()).item() < 1e-4


# BASELINE-size layers whose dispatch lands on the 512x128 / 256x256 forward tiles and the 256x128 data gradient
@pytest.mark.parametrize('C,H,fwd_cfgs', [(128, 28, (2,)), (256, 14, (3,)), (512, 7, (0,))])
def test_fullsize_3x3_layers_on_their_dispatched_configuration(C, H, fwd_cfgs):
    """bs 256, unforced: the forward must land on the ping-pong configuration the benchmark times, the plain data gradient
    on configuration 0; y / dx against torch fp32 on the CPU from the bf16-rounded operands."""
    _lib, lib, cnt = _lib_and_counter()
    from convnets_amd import ops
    N = 256
    x, dy, w = _operands((N, C, H, H), 3)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    yr = F.conv2d(x.float(), w.bfloat16().float(), padding=1)
    dxr = F.conv_transpose2d(dy.float(), w.bfloat16().float(), padding=1)
    xg, dyg = x.cuda().contiguous(memory_format=CL), dy.cuda().contiguous(memory_format=CL)
    desc = ops.make_desc(xg.shape, w.shape, (1, 1), (1, 1), _lib.CNK_BF16)
    wf, wd = ops.PackedWeights().get(w.cuda(), desc, True)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    with _lib.switches(CNK_PP3X3=None, CNK_PP_CFG=None, CNK_PP_FULL=None):
        nsl = lib.cnk_conv2d_fwd_bn_partial_slices(ctypes.byref(desc))
        part = torch.full((nsl, C, 3), float('nan'), device='cuda')
        y, dx = torch.empty_like(xg), torch.empty_like(xg)
        before = [cnt(i) for i in range(4)]
        assert lib.cnk_conv2d_fwd(ctypes.byref(desc), P(xg), P(wf), None, P(y), P(part), None, 0, st) == 0, lib.cnk_last_error()
        mid = [cnt(i) for i in range(4)]
        assert lib.cnk_conv2d_dgrad(ctypes.byref(desc), P(dyg), P(wd), P(dx), None, None, None, None, 0, st) == 0, lib.cnk_last_error()
        torch.cuda.synchronize()
        after = [cnt(i) for i in range(4)]
    ran_fwd = [i for i in range(4) if mid[i] > before[i]]
    ran_dg = [i for i in range(4) if after[i] > mid[i]]
    assert len(ran_fwd) == 1 and ran_fwd[0] in fwd_cfgs, 'forward ran on %s' % ran_fwd
    assert ran_dg == [0], 'data gradient ran on %s' % ran_dg
    assert nsl == (N * H * H + PP_WROWS[ran_fwd[0]] - 1) // PP_WROWS[ran_fwd[0]]
    assert rel_l2(host(y), yr.numpy()) < 4e-3
    assert rel_l2(host(dx), dxr.numpy()) < 4e-3
    tmean, tvar = _merge_stats(part, N * H * H)
    yd = y.double()
    assert ((tmean - yd.mean((0, 2, 3))).abs().max() / yd.abs().mean()).item() < 2e-6
    assert ((tvar - yd.var((0, 2, 3), unbiased=False)).abs() / tvar).max().item() < 2e-5


def test_fullsize_alexnet_fp32_step_properties():
    """BASELINE config 2: AlexNet 224x224, bs 128, fp32 (11x11/4 direct kernel at N = 128, 5x5 and 3x3 implicit GEMM, the
    split-K fp32 head).  Size-independent properties of real steps: initial loss 128*ln(1000), finite gradients, moving
    weights, the loss goes down."""
    import convnets_amd as ca
    torch.manual_seed(21)
    s = ca.Settings(kind=0, input_size=(3, 224, 224), num_classes=1000, batch_size=128, mixed_precision=False, learning_rate=1e-4,
                    dropout_rate=0)
    model = ca.AlexNet(s).cuda()
    model.init_optimizer()
    model.train()
    gen = torch.Generator(device='cuda').manual_seed(21)
    x = torch.randn(128, 3, 224, 224, generator=gen, device='cuda')
    y = torch.randint(0, 1000, (128,), generator=gen, device='cuda')
    before = model.optimizer.flat_p.clone()
    loss1, correct = model.train_batch((x, y))
    torch.cuda.synchronize()
    assert abs(float(loss1) - 128 * math.log(1000)) < 0.05 * 128 * math.log(1000)
    assert 0 <= int(correct) <= 128
    gflat = model.optimizer.flat_g
    assert bool(torch.isfinite(gflat).all()) and float(gflat.abs().max()) > 0
    assert float((model.optimizer.flat_p - before).abs().max()) > 5e-5
    for _ in range(3):
        lossn, _ = model.train_batch((x, y))
    assert float(lossn) < float(loss1)
    # first-layer check at full size against torch fp32 on the CPU: 11x11 / stride 4 at N = 128
    conv1 = [m for m in model.modules() if isinstance(m, torch.nn.Conv2d)][0]
    assert conv1.kernel_size == (11, 11) and conv1.stride == (4, 4)
    with torch.no_grad():
        xc = x[:128].contiguous(memory_format=CL)
        yg = conv1(xc)
        torch.cuda.synchronize()
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        yc = F.conv2d(x[:128].cpu(), conv1.weight.detach().cpu(), None if conv1.bias is None else conv1.bias.detach().cpu(),
                      stride=conv1.stride, padding=conv1.padding)
    assert rel_l2(host(yg), yc.numpy()) < 1e-5
    del model
    torch.cuda.empty_cache()


@pytest.mark.parametrize('kind,Cin,Cout,H', [('3x3', 128, 128, 28), ('3x3', 256, 256, 14), ('1x1', 1024, 256, 14), ('1x1', 512, 2048, 7),
                                             ('1x1', 512, 256, 28)])
def test_dispatch_rules_batch_sweep(kind, Cin, Cout, H):
    """The tile choices (pp_pick, gemm1x1_pick) are rules in work per CU (tiles per CU, K steps), not batch-256 windows:
    whatever they choose at N in {32, 64, 96, 128, 192, 256}, forward (with statistics) and data gradient equal the
    reference kernels' results -- bit for bit for the 1x1 ring GEMM and the 64-channel-step 3x3 configuration, within
    bf16 accumulation-order noise (<= 2e-3) for the 32-channel-step 3x3 tiles."""
    _lib, lib, cnt = _lib_and_counter()
    from convnets_amd import ops
    k, pad = (3, 1) if kind == '3x3' else (1, 0)
    g = torch.Generator().manual_seed(5)
    w = torch.randn(Cout, Cin, k, k, generator=g) * (1.0 / math.sqrt(k * k * Cin))
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for N in (32, 64, 96, 128, 192, 256):
        xg = torch.randn(N, Cin, H, H, generator=g).bfloat16().cuda().contiguous(memory_format=CL)
        dyg = torch.randn(N, Cout, H, H, generator=g).bfloat16().cuda().contiguous(memory_format=CL)
        desc = ops.make_desc(xg.shape, w.shape, (1, 1), (pad, pad), _lib.CNK_BF16)
        wf, wd = ops.PackedWeights().get(w.cuda(), desc, True)
        M = N * H * H
        res = {}
        for name, env in (('rule', dict(CNK_PP3X3=None, CNK_PP_CFG=None, CNK_GEMM1X1=None)), ('base', dict(CNK_PP3X3='0', CNK_GEMM1X1='0'))):
            with _lib.switches(**env):
                nsl = lib.cnk_conv2d_fwd_bn_partial_slices(ctypes.byref(desc))
                part = torch.full((nsl, Cout, 3), float('nan'), device='cuda')
                y, dx = torch.empty_like(dyg), torch.empty_like(xg)
                c0 = [cnt(i) for i in range(10)]
                assert lib.cnk_conv2d_fwd(ctypes.byref(desc), P(xg), P(wf), None, P(y), P(part), None, 0, st) == 0, lib.cnk_last_error()
                c1 = [cnt(i) for i in range(10)]
                assert lib.cnk_conv2d_dgrad(ctypes.byref(desc), P(dyg), P(wd), P(dx), None, None, None, None, 0, st) == 0, lib.cnk_last_error()
                torch.cuda.synchronize()
                fwd_on = [i for i in range(10) if c1[i] > c0[i]]
                res[name] = (y, dx, _merge_stats(part, M), fwd_on)
        y, dx, (tm, tv), fwd_on = res['rule']
        yb, dxb, (bm, bv), _ = res['base']
        same_order = not (2 in fwd_on or 3 in fwd_on)          # the 32-channel-step 3x3 tiles accumulate in another order
        if same_order:
            assert torch.equal(y, yb), (N, fwd_on)
        else:
            assert rel_l2(host(y), host(yb)) < 2e-3, (N, fwd_on)
        assert torch.equal(dx, dxb), N
        assert ((tm - bm).abs().max() / yb.float().abs().mean()).item() < (1e-6 if same_order else 1e-4)
        assert ((tv - bv).abs() / bv).max().item() < (1e-5 if same_order else 1e-3)


@pytest.mark.parametrize('slices,C,M', [(6272, 64, 802816), (6272, 256, 802816), (1568, 512, 200704), (3136, 128, 401408), (1100, 72, 140800)])
def test_bn_partials_merged_and_finalised_in_one_launch(slices, C, M):
    """csrc/bn.hip bn_merge_final_kernel: long BN partial lists are merged AND finalised by one launch
    (write-through stores, a ticket per 64-channel group, the last block finalises) instead of bn_premerge + bn_finalize /
    bn_bwd_final.  Hand-off test as MI355X_MICROARCH.md prescribes: many launches back to back on the SAME scratch with
    DIFFERENT data each (a stale share is the previous launch's data and shows), consumer L1-warm, a streaming kernel
    running beside it on a second stream (uneven load), every output word checked: against an fp64 merge, bit for bit
    between repeated launches, and against the two-launch path (CNK_BN_FUSED_FINAL=0)."""
    _lib, lib, cnt = _lib_and_counter()
    g = torch.Generator(device='cuda').manual_seed(slices + C)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    nws = lib.cnk_bn_workspace_bytes(M, C)
    ws = torch.empty(nws, dtype=torch.uint8, device='cuda')
    rows = M // slices
    assert rows * slices == M
    NB = 6
    parts, expect = [], []
    for b in range(NB):
        p = torch.empty(slices, C, 3, device='cuda')
        p[:, :, 0] = rows
        p[:, :, 1] = torch.randn(slices, C, generator=g, device='cuda') * (0.5 + b) + b
        p[:, :, 2] = (torch.rand(slices, C, generator=g, device='cuda') + 0.5) * rows * (1 + b)
        parts.append(p)
        pd = p.double()
        mean = pd[:, :, 1].mean(0)
        var = (pd[:, :, 2].sum(0) + rows * ((pd[:, :, 1] - mean) ** 2).sum(0)) / M
        expect.append((mean, var))
    gamma, beta = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda')
    side = torch.cuda.Stream()
    big = torch.empty(64 << 20, device='cuda')

    def run(fused, reps):
        outs = []
        with _lib.switches(CNK_BN_FUSED_FINAL='1' if fused else '0'):
            c0 = cnt(10)
            for r in range(reps):
                b = r % NB
                sm, si = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
                rm, rv = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
                if r % 3 == 0:
                    with torch.cuda.stream(side):
                        big.mul_(1.0001)                      # streaming traffic beside the merge (uneven load)
                assert lib.cnk_bn_fwd_train(P(parts[b]), P(gamma), P(beta), None, None, None, P(rm), P(rv), P(sm), P(si), P(parts[b]), slices,
                                            M, C, 1e-5, 0.1, 1, _lib.CNK_BF16, P(ws), nws, st) == 0, lib.cnk_last_error()
                coef = ws.view(torch.float32)[:2 * C].clone()       # scale | shift, read in stream order
                outs.append((b, sm, si, rm, rv, coef))
            torch.cuda.synchronize()
            assert (cnt(10) - c0 == reps) == fused, 'fused finalisation ran %d of %d times' % (cnt(10) - c0, reps)
        return outs

    fused = run(True, 120)
    plain = run(False, NB)
    first = {}
    for b, sm, si, rm, rv, coef in fused:
        mean, var = expect[b]
        assert ((sm.double() - mean).abs().max() / mean.abs().max()).item() < 2e-6
        assert ((1.0 / si.double() ** 2 - 1e-5 - var).abs() / var).max().item() < 2e-5
        inv = si.double()
        assert ((coef[:C].double() - gamma.double() * inv).abs().max() / (gamma.double() * inv).abs().max()).item() < 1e-6
        assert ((coef[C:].double() - (beta.double() - sm.double() * gamma.double() * inv)).abs().max()).item() < 1e-4
        assert ((rm.double() - 0.1 * mean).abs().max() / mean.abs().max()).item() < 2e-6
        key = first.setdefault(b, (sm, si, rm, rv, coef))
        for a, c in zip(key, (sm, si, rm, rv, coef)):
            assert torch.equal(a, c), 'repeated launches differ (a stale share was read)'
    for b, sm, si, rm, rv, coef in plain:
        assert ((sm - first[b][0]).abs().max() / first[b][0].abs().max()).item() < 1e-6
        assert ((si - first[b][1]).abs() / first[b][1]).max().item() < 1e-5
    # backward: the same launch finalises dgamma / dbeta and the apply coefficients (cnk_bn_bwd with ext_partial)
    if C % 8:
        return
    x = torch.randn(M, C, device='cuda', generator=g).bfloat16()
    dy = torch.randn(M, C, device='cuda', generator=g).bfloat16()
    sm, si = torch.randn(C, device='cuda') * 0.1, torch.rand(C, device='cuda') + 0.5
    bparts = [torch.randn(slices, C, 2, device='cuda', generator=g) * (1 + b) for b in range(3)]
    res = {}
    for fused_on in (True, False):
        with _lib.switches(CNK_BN_FUSED_FINAL='1' if fused_on else '0'):
            c0 = cnt(10)
            for r in range(9 if fused_on else 3):
                b = r % 3
                dx, dg, db = torch.empty_like(x), torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
                assert lib.cnk_bn_bwd(P(dy), P(x), None, None, P(gamma), P(beta), P(sm), P(si), P(dx), None, P(dg), P(db), 0,
                                      P(bparts[b]), slices, M, C, 1, _lib.CNK_BF16, P(ws), nws, st) == 0, lib.cnk_last_error()
                torch.cuda.synchronize()
                want_b, want_g = bparts[b][:, :, 0].double().sum(0), bparts[b][:, :, 1].double().sum(0)
                assert ((db.double() - want_b).abs().max() / want_b.abs().max()).item() < 1e-5
                assert ((dg.double() - want_g).abs().max() / want_g.abs().max()).item() < 1e-5
                assert bool(torch.isfinite(dx.float()).all())
                prev = res.setdefault((fused_on, b), (dx, dg, db))
                assert all(torch.equal(a, c) for a, c in zip(prev, (dx, dg, db)))
            assert (cnt(10) - c0 > 0) == fused_on
    for b in range(3):
        assert rel_l2(host(res[(True, b)][0]), host(res[(False, b)][0])) < 1e-3          # bf16 dx from coefficients equal to ~1e-7


@pytest.mark.parametrize('style,N,Cin,Cout,H', [('c1', 8, 256, 64, 14), ('c1', 5, 512, 128, 9), ('c3', 8, 64, 256, 14), ('c3', 3, 128, 256, 11),
                                                ('c1', 2, 256, 256, 7)])
def test_1x1_data_gradients_with_their_training_step_epilogues(style, N, Cin, Cout, H):
    """The 1x1 data gradients exactly as the bottleneck blocks issue them, at op level against fp64 (round 3: the epilogue's
    coefficient loads were restructured).  c1: conv1 of a bottleneck (src/resnet.py:170): gradient addend + packed mask +
    fused BN backward with saved mask bits; c3: conv3 (:172): fused BN backward, mask re-derived."""
    _lib, lib, cnt = _lib_and_counter()
    from convnets_amd import ops
    g = torch.Generator().manual_seed(Cin + H)
    M = N * H * H
    dy = torch.randn(N, Cout, H, H, generator=g).bfloat16()
    w = torch.randn(Cout, Cin, 1, 1, generator=g) * (1.0 / math.sqrt(Cout))
    dyg = dy.cuda().contiguous(memory_format=CL)
    desc = ops.make_desc((N, Cin, H, H), w.shape, (1, 1), (0, 0), _lib.CNK_BF16)
    _, wd = ops.PackedWeights().get(w.cuda(), desc, True)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    bnx = (torch.randn(N, H, H, Cin, generator=g) * 1.3 + 0.2).bfloat16().cuda()
    addend = torch.randn(N, H, H, Cin, generator=g).bfloat16().cuda() if style == 'c1' else None
    amask = torch.randint(0, 256, (M * Cin // 8,), generator=g, dtype=torch.uint8).cuda() if style == 'c1' else None
    kmask = torch.randint(0, 256, (M * Cin // 8,), generator=g, dtype=torch.uint8).cuda() if style == 'c1' else None
    bmean, binv = (torch.randn(Cin, generator=g) * 0.1).cuda(), (torch.rand(Cin, generator=g) + 0.5).cuda()
    bgam, bbeta = (torch.rand(Cin, generator=g) + 0.5).cuda(), (torch.randn(Cin, generator=g) * 0.3).cuda()
    nb = lib.cnk_conv2d_dgrad_bn_slices(ctypes.byref(desc))
    assert nb > 0
    bpart = torch.full((nb, Cin, 2), float('nan'), device='cuda')
    fuse = _lib.BnBwdFuse(bnx.data_ptr(), bmean.data_ptr(), binv.data_ptr(), bgam.data_ptr(), bbeta.data_ptr(),
                          kmask.data_ptr() if kmask is not None else None, bpart.data_ptr(), 1)
    dx = torch.empty(N, Cin, H, H, device='cuda', dtype=torch.bfloat16).contiguous(memory_format=CL)
    assert lib.cnk_conv2d_dgrad(ctypes.byref(desc), P(dyg), P(wd), P(dx), P(addend), P(amask), ctypes.byref(fuse), None, 0, st) == 0, lib.cnk_last_error()
    torch.cuda.synchronize()
    assert torch.isfinite(bpart).all()
    shp = (N, H, H, Cin)
    if style == 'c1':
        abits, keep = _unpack_bits(amask, shp), _unpack_bits(kmask, shp)
    else:
        abits = None
        sc = (bgam * binv).double().cpu()
        keep = ((bnx.double().cpu() * sc + (bbeta.double().cpu() - bmean.double().cpu() * sc)) > 0).double()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    v, s1, s2 = _dgrad_reference(dy, w, addend, abits, bnx, keep, bmean, binv, padding=0)
    assert rel_l2(_nhwc(dx).float().numpy(), v.float().numpy()) < 4e-3
    got = bpart.double().sum(0).cpu()
    assert ((got[:, 0] - s1).abs().max() / s1.abs().max()).item() < 5e-3
    assert ((got[:, 1] - s2).abs().max() / s2.abs().max()).item() < 5e-3


@pytest.mark.parametrize('Cin,Cout,k,H,N', [(128, 128, 3, 28, 16), (256, 512, 1, 14, 32), (64, 64, 3, 56, 8), (512, 128, 1, 28, 8)])
def test_weight_gradient_split_depth_follows_the_workspace(Cin, Cout, k, H, N):
    """cnk_conv2d_wgrad (include/cnk.h): a workspace below cnk_conv2d_workspace_bytes bounds the number of pixel-range splits
    (round 3: the side-stream weight gradients of the training step run with a share of the full depth, ops.WGRAD_SHARE).
    Full, half, and one-slab workspaces give the same gradient up to the fp32 order of the slab sum; less than a slab is refused."""
    _lib, lib, cnt = _lib_and_counter()
    from convnets_amd import ops
    g = torch.Generator().manual_seed(Cin + k)
    x = torch.randn(N, Cin, H, H, generator=g).bfloat16()
    dy = torch.randn(N, Cout, H, H, generator=g).bfloat16()
    xg, dyg = x.cuda().contiguous(memory_format=CL), dy.cuda().contiguous(memory_format=CL)
    desc = ops.make_desc(xg.shape, (Cout, Cin, k, k), (1, 1), (k // 2, k // 2), _lib.CNK_BF16)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    nws = lib.cnk_conv2d_workspace_bytes(ctypes.byref(desc), _lib.PASS_WGRAD)
    slab = (Cout * Cin * k * k * 4 + 255) // 256 * 256
    assert nws >= 4 * slab, 'the full plan splits at least four ways at this size'
    ws = torch.empty(nws, dtype=torch.uint8, device='cuda')
    out = {}
    for name, budget in (('full', nws), ('half', nws // 2 // 256 * 256), ('one', slab)):
        dw = torch.zeros(Cout, Cin, k, k, device='cuda')
        assert lib.cnk_conv2d_wgrad(ctypes.byref(desc), P(dyg), P(xg), P(dw), None, 0, P(ws), budget, st) == 0, lib.cnk_last_error()
        torch.cuda.synchronize()
        out[name] = dw
    assert lib.cnk_conv2d_wgrad(ctypes.byref(desc), P(dyg), P(xg), P(out['one']), None, 0, P(ws), slab - 256, st) != 0
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    xr = x.double().requires_grad_(False)
    wref = torch.zeros(Cout, Cin, k, k, dtype=torch.float64, requires_grad=True)
    F.conv2d(xr, wref, padding=k // 2).backward(dy.double())
    ref = wref.grad.float().numpy()
    for name in out:
        assert rel_l2(host(out[name]), ref) < 2e-5, name
    assert rel_l2(host(out['half']), host(out['full'])) < 2e-6 and rel_l2(host(out['one']), host(out['full'])) < 2e-6


def test_wgrad_share_calibration_leaves_training_state_alone():
    """ops.WGRAD_SHARE is MEASURED (GraphedTrainStep(wgrad_share='auto'), model.calibrate_wgrad_share): the measurement steps
    run on a snapshot -- parameters, Adam moments, BN statistics and the step count are what they were."""
    import convnets_amd as ca
    from convnets_amd import ops
    torch.manual_seed(21)
    s = ca.Settings(kind=26, input_size=(3, 32, 32), num_classes=10, batch_size=32, mixed_precision=True, dropout_rate=0)
    m = ca.ResNet(s).cuda()
    m.init_optimizer()
    m.train()
    x, y = torch.randn(32, 3, 32, 32, device='cuda'), torch.randint(0, 10, (32,), device='cuda')
    m.train_batch((x, y))
    torch.cuda.synchronize()
    keep = ops.WGRAD_SHARE[0]
    try:
        before = (m.optimizer.flat_p.clone(), m.optimizer.exp_avg.clone(), m.optimizer.step_count, [b.clone() for b in m.buffers()])
        t = m.calibrate_wgrad_share((x, y))
        assert set(t) == {1.0, 0.75, 0.5} and ops.WGRAD_SHARE[0] in t
        g = ca.GraphedTrainStep(m, (x, y), wgrad_share='auto')
        assert g.wgrad_share_ms is not None and ops.WGRAD_SHARE[0] in g.wgrad_share_ms
        torch.cuda.synchronize()
        assert torch.equal(before[0], m.optimizer.flat_p) and torch.equal(before[1], m.optimizer.exp_avg)
        assert before[2] == m.optimizer.step_count
        for a, b in zip(before[3], m.buffers()):
            assert torch.equal(a, b)
        l1, _ = g.replay()
        l2, _ = g.replay()
        assert float(l2) < float(l1) * 1.5 and m.optimizer.step_count == before[2] + 2
    finally:
        ops.WGRAD_SHARE[0] = keep


@pytest.mark.parametrize('N,C,H', [(128, 256, 14), (256, 512, 7), (128, 512, 7)])
def test_pingpong_full_epilogue_under_load(N, C, H):
    """Regression (round 3): the fused data gradient on the ping-pong core with EVERY CU busy (196 / 100 single-tile workgroups:
    the ResNet-152 bs 128 and ResNet-50 bs 256 shapes).  Its hand-counted s_waitcnt vmcnt used to credit the epilogue's own
    stores as still outstanding; on a busy chip store acknowledgements overtake older loads, a wait then passed with a side
    operand still in flight, and a few of the 196 tiles came out wrong -- never at the small test sizes.  Five launches each
    must equal the halo kernel bit for bit (dx) and in the sums."""
    _lib, lib, cnt = _lib_and_counter()
    from convnets_amd import ops
    g = torch.Generator().manual_seed(N + C)
    M = N * H * H
    dy = torch.randn(N, C, H, H, generator=g).bfloat16().cuda().contiguous(memory_format=CL)
    w = torch.randn(C, C, 3, 3, generator=g) * 0.02
    desc = ops.make_desc(dy.shape, w.shape, (1, 1), (1, 1), _lib.CNK_BF16)
    _, wd = ops.PackedWeights().get(w.cuda(), desc, True)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    bnx = (torch.randn(N, H, H, C, generator=g) * 1.3 + 0.2).bfloat16().cuda()
    addend = torch.randn(N, H, H, C, generator=g).bfloat16().cuda()
    amask = torch.randint(0, 256, (M * C // 8,), generator=g, dtype=torch.uint8).cuda()
    bmean, binv = (torch.randn(C, generator=g) * 0.1).cuda(), (torch.rand(C, generator=g) + 0.5).cuda()
    bgam, bbeta = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.3).cuda()

    def run(env, reps):
        outs = []
        with _lib.switches(**env):
            nb = lib.cnk_conv2d_dgrad_bn_slices(ctypes.byref(desc))
            for r in range(reps):
                bpart = torch.full((nb, C, 2), float('nan'), device='cuda')
                fuse = _lib.BnBwdFuse(bnx.data_ptr(), bmean.data_ptr(), binv.data_ptr(), bgam.data_ptr(), bbeta.data_ptr(), None, bpart.data_ptr(), 1)
                dx = torch.full_like(dy, float('nan'))
                c0, f0 = cnt(CNT_PP[0]), cnt(CNT_PP_FULL)
                assert lib.cnk_conv2d_dgrad(ctypes.byref(desc), P(dy), P(wd), P(dx), P(addend), P(amask), ctypes.byref(fuse), None, 0, st) == 0, lib.cnk_last_error()
                torch.cuda.synchronize()
                outs.append((dx, bpart.double().sum(0), cnt(CNT_PP_FULL) - f0))
        return outs
    ref = run(dict(CNK_PP3X3='0'), 1)[0]
    assert ref[2] == 0
    for dx, sums, ran in run(dict(CNK_PP3X3='1', CNK_PP_CFG=None, CNK_PP_FULL=None), 5):
        assert ran == 1, 'the unforced dispatch must put this shape on the ping-pong core'
        assert torch.equal(dx, ref[0])
        assert ((sums - ref[1]).abs().max() / ref[1].abs().max()).item() < 1e-5


@pytest.mark.parametrize('dtype', ['bf16', 'f32'])
@pytest.mark.parametrize('N,C,H,W', [(4, 64, 32, 32), (3, 64, 15, 18), (2, 128, 9, 7), (2, 24, 12, 12), (1, 8, 2, 2)])
def test_stem_pool_with_folded_bn_streaming_kernel(N, C, H, W, dtype):
    """cnk_maxpool2d_fwd_bn (conv -> BN -> ReLU -> MaxPool2d 3x3/2/1, src/resnet.py:66-67): the streaming kernel of round 3
    (nine unconditional tap loads, coefficients in registers, packed arg-max store) against torch on the CPU -- values bit
    for bit, and the arg-max bytes must name the element torch's max_pool2d returns (first maximum in scan order)."""
    _lib, lib, cnt = _lib_and_counter()
    tdt = torch.bfloat16 if dtype == 'bf16' else torch.float32
    code = _lib.CNK_BF16 if dtype == 'bf16' else _lib.CNK_F32
    g = torch.Generator().manual_seed(C + H)
    x = (torch.randn(N, H, W, C, generator=g) * 2).to(tdt)
    x[0, :2, :2] = x[0, 0, 0]                                  # ties inside a window
    scale, shift = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.5
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    xg = x.cuda()
    y = torch.empty(N, Ho, Wo, C, dtype=tdt, device='cuda')
    idx = torch.full((N, Ho, Wo, C), 255, dtype=torch.uint8, device='cuda')
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.cnk_maxpool2d_fwd_bn(P(xg), P(scale.cuda()), P(shift.cuda()), 1, P(y), P(idx), N, H, W, C, 3, 3, 2, 2, 1, 1, code, st) == 0, lib.cnk_last_error()
    torch.cuda.synchronize()
    a = torch.relu(x.float() * scale + shift).to(tdt)          # fp32 affine (one fma-free multiply-add), rounded like the BN pass stores it
    ref, ri = F.max_pool2d(a.float().permute(0, 3, 1, 2), 3, 2, 1, return_indices=True)
    assert torch.equal(y.float().cpu().permute(0, 3, 1, 2), ref)
    pos = idx.cpu().permute(0, 3, 1, 2).long()
    ho = torch.arange(Ho).view(1, 1, Ho, 1)
    wo = torch.arange(Wo).view(1, 1, 1, Wo)
    flat = (2 * ho - 1 + pos // 3) * W + (2 * wo - 1 + pos % 3)
    assert torch.equal(flat, ri)


def test_graph_replay_follows_a_learning_rate_change_without_recapture():
    """GraphedTrainStep: the scheduler's new learning rate (src/basemodel.py:69-76) reaches the captured Adam launch through
    the device record (bc1 * lr_captured / lr_now), not through a re-capture.  Two replays at lr, two at lr / 10, against the
    eager steps of an identical model: same parameters to fp32 rounding of one quotient."""
    import copy
    import convnets_amd as ca
    torch.manual_seed(21)
    s = ca.Settings(kind=18, input_size=(3, 32, 32), num_classes=10, batch_size=16, mixed_precision=True, dropout_rate=0, learning_rate=1e-3)
    x, y = torch.randn(16, 3, 32, 32, device='cuda'), torch.randint(0, 10, (16,), device='cuda')
    models = []
    for _ in range(2):
        torch.manual_seed(21)
        m = ca.ResNet(s).cuda()
        m.init_optimizer()
        m.train()
        models.append(m)
    a, b = models
    assert torch.equal(a.optimizer.flat_p, b.optimizer.flat_p)
    g = ca.GraphedTrainStep(a, (x, y))
    graph_obj = g.graph
    for lr in (1e-3, 1e-3, 1e-4, 1e-4):
        a.optimizer.param_groups[0]['lr'] = lr
        b.optimizer.param_groups[0]['lr'] = lr
        g.replay()
        b.train_batch((x, y))
    torch.cuda.synchronize()
    assert g.graph is graph_obj, 'no re-capture'
    pa, pb = a.optimizer.flat_p, b.optimizer.flat_p
    # (shown when the assertion below fails: which tensors differ, and the process-wide state the two paths share)
    from convnets_amd import ops
    off, dgr, dpa = 0, (a.optimizer.flat_g - b.optimizer.flat_g).abs(), (pa - pb).abs()
    for n, prm in a.named_parameters():
        k = prm.numel()
        if k and float(dpa[off:off + k].max()) > 0:
            print('differs: %-40s %-18s grad %.3e param %.3e' % (n, tuple(prm.shape), float(dgr[off:off + k].max()), float(dpa[off:off + k].max())))
        off += k
    print('WGRAD_SHARE', ops.WGRAD_SHARE, 'WGRAD_CHAIN', ops.WGRAD_CHAIN, 'chain', {k: (v[0] is not None, v[2]) for k, v in ops._wgrad_chain.items()},
          'join queued', ops._join_queued, 'BN_FOLD_MIN_ELEMS', ops.BN_FOLD_MIN_ELEMS)
    if float(dpa.max()) > 0:
        print('workspaces', [(k[1] & 0xffff, k[2], v.numel(), hex(v.data_ptr())) for k, v in ops._workspaces.items() if v.numel() > (4 << 20)])
        keep = ops.WGRAD_CHAIN
        for chain in (False, True):
            ops.WGRAD_CHAIN = chain
            torch.manual_seed(21)
            c = ca.ResNet(s).cuda()
            c.init_optimizer()
            c.train()
            for lr in (1e-3, 1e-3, 1e-4, 1e-4):
                c.optimizer.param_groups[0]['lr'] = lr
                c.train_batch((x, y))
                torch.cuda.synchronize()
            print('a third model, eager with a sync per step, chain %s: graph model differs by %.3e, eager model by %.3e'
                  % (chain, float((pa - c.optimizer.flat_p).abs().max()), float((pb - c.optimizer.flat_p).abs().max())))
        ops.WGRAD_CHAIN = keep
    assert ((pa - pb).abs().max() / pb.abs().max()).item() < 1e-6
    step_a = (pa - models[0].optimizer.flat_p).abs().max()
    assert a.optimizer.step_count == b.optimizer.step_count == 4


# ---- weight-stationary 64 -> 64 channel kernel (csrc/conv_pp.hip ws3x3_kernel; reference site src/resnet.py:171 at 56x56) --------
CNT_PP_WS = 12
WS_SHAPES = {
    'n3_56': (3, 64, 56, 56),          # 37 tiles, ragged last one, one tile per workgroup
    'n64_56': (64, 64, 56, 56),        # 784 tiles: three / four tiles per workgroup, the patch buffers rotate through all phases
    'n40_28': (40, 64, 28, 28),        # tiles span image boundaries
    'n50_9x13': (50, 64, 9, 13),       # odd, non-square image
}


@pytest.mark.parametrize('shape', list(WS_SHAPES))
def test_weight_stationary_64_channel_kernel(shape):
    """Forward (plain rows / BN statistics partials) and data gradient (plain / gradient addend + packed masks + fused BN
    backward, mask bits and re-derived mask) of the 64 -> 64 channel 3x3 layers on the weight-stationary kernel (unforced
    dispatch except CNK_PP_FULL=1, which also sends the fused gradients there) with a launch counter, against fp64 from the bf16-rounded operands, against the tensor the kernel itself wrote
    (statistics and BN sums) and against the r01 halo kernel (other K order: two valid bf16 roundings, not the same bits)."""
    _lib, lib, cnt = _lib_and_counter()
    from convnets_amd import ops
    N, C, H, W = WS_SHAPES[shape]
    M = N * H * W
    x, dy, w = _operands(WS_SHAPES[shape], 41)
    g = torch.Generator().manual_seed(43)
    xg, dyg = x.cuda().contiguous(memory_format=CL), dy.cuda().contiguous(memory_format=CL)
    desc = ops.make_desc(xg.shape, w.shape, (1, 1), (1, 1), _lib.CNK_BF16)
    wf, wd = ops.PackedWeights().get(w.cuda(), desc, True)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    bnx = (torch.randn(N, H, W, C, generator=g) * 1.3 + 0.2).bfloat16().cuda()
    addend = torch.randn(N, H, W, C, generator=g).bfloat16().cuda()
    amask = torch.randint(0, 256, (M * C // 8,), generator=g, dtype=torch.uint8).cuda()
    kmask = torch.randint(0, 256, (M * C // 8,), generator=g, dtype=torch.uint8).cuda()
    bmean, binv = (torch.randn(C, generator=g) * 0.1).cuda(), (torch.rand(C, generator=g) + 0.5).cuda()
    bgam, bbeta = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.3).cuda()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    out = {}
    for name, env in (('ws', dict(CNK_PP3X3=None, CNK_PP_CFG=None, CNK_PP_FULL='1', CNK_PP_WS=None)), ('halo', dict(CNK_PP3X3='0'))):
        with _lib.switches(**env):
            nsl = lib.cnk_conv2d_fwd_bn_partial_slices(ctypes.byref(desc))
            nb = lib.cnk_conv2d_dgrad_bn_slices(ctypes.byref(desc))
            if name == 'ws':
                assert nsl == (M + 63) // 64 and nb == (M + 63) // 64, 'the weight-stationary kernel does not take this shape'
            part = torch.full((nsl, C, 3), float('nan'), device='cuda')
            y, y2 = torch.empty_like(xg), torch.empty_like(xg)
            dxp = torch.empty_like(dyg)
            dxf = {}
            c0 = cnt(CNT_PP_WS)
            assert lib.cnk_conv2d_fwd(ctypes.byref(desc), P(xg), P(wf), None, P(y), P(part), None, 0, st) == 0, lib.cnk_last_error()
            assert lib.cnk_conv2d_fwd(ctypes.byref(desc), P(xg), P(wf), None, P(y2), None, None, 0, st) == 0, lib.cnk_last_error()
            assert lib.cnk_conv2d_dgrad(ctypes.byref(desc), P(dyg), P(wd), P(dxp), None, None, None, None, 0, st) == 0, lib.cnk_last_error()
            for maskmode in ('bits', 'remask'):
                bpart = torch.full((nb, C, 2), float('nan'), device='cuda')
                fuse = _lib.BnBwdFuse(bnx.data_ptr(), bmean.data_ptr(), binv.data_ptr(), bgam.data_ptr(), bbeta.data_ptr(),
                                      kmask.data_ptr() if maskmode == 'bits' else None, bpart.data_ptr(), 1)
                d = torch.empty_like(dyg)
                assert lib.cnk_conv2d_dgrad(ctypes.byref(desc), P(dyg), P(wd), P(d), P(addend), P(amask), ctypes.byref(fuse), None, 0, st) == 0, lib.cnk_last_error()
                torch.cuda.synchronize()
                assert torch.isfinite(bpart).all()
                dxf[maskmode] = (d, bpart.double().sum(0).cpu())
            torch.cuda.synchronize()
            assert cnt(CNT_PP_WS) - c0 == (5 if name == 'ws' else 0), 'the weight-stationary kernel did not run' if name == 'ws' else 'switch ignored'
            out[name] = (y, y2, part, dxp, dxf)
    y, y2, part, dxp, dxf = out['ws']
    yh, _, _, dxph, dxfh = out['halo']
    # forward
    ref = F.conv2d(x.double(), w.bfloat16().double(), padding=1)
    assert torch.equal(y, y2)
    assert rel_l2(host(y), ref.float().numpy()) < 4e-3
    assert rel_l2(host(y), host(yh)) < 3e-3
    tmean, tvar = _merge_stats(part, M)
    yd = y.double()
    assert ((tmean - yd.mean((0, 2, 3))).abs().max() / yd.abs().mean()).item() < 2e-6
    assert ((tvar - yd.var((0, 2, 3), unbiased=False)).abs() / tvar).max().item() < 2e-5
    # data gradient
    shp = (N, H, W, C)
    vplain, _, _ = _dgrad_reference(dy, w, None, None, None, None, None, None)
    assert rel_l2(_nhwc(dxp).float().numpy(), vplain.float().numpy()) < 4e-3
    assert rel_l2(host(dxp), host(dxph)) < 3e-3
    abits = _unpack_bits(amask, shp)
    for maskmode in ('bits', 'remask'):
        if maskmode == 'bits':
            keep = _unpack_bits(kmask, shp)
        else:
            sc = (bgam * binv).double().cpu()
            sf = bbeta.double().cpu() - bmean.double().cpu() * sc
            keep = ((bnx.double().cpu() * sc + sf) > 0).double()
        v, s1, s2 = _dgrad_reference(dy, w, addend, abits, bnx, keep, bmean, binv)
        d, got = dxf[maskmode]
        assert rel_l2(_nhwc(d).float().numpy(), v.float().numpy()) < 4e-3
        assert rel_l2(host(d), host(dxfh[maskmode][0])) < 3e-3
        assert ((got[:, 0] - s1).abs().max() / s1.abs().max()).item() < 5e-3
        assert ((got[:, 1] - s2).abs().max() / s2.abs().max()).item() < 5e-3
        wrote = _nhwc(d)
        assert ((got[:, 0] - wrote.sum((0, 1, 2))).abs().max() / wrote.abs().sum((0, 1, 2)).max()).item() < 1e-5
        q = (wrote * (bnx.double().cpu() - bmean.double().cpu())).sum((0, 1, 2)) * binv.double().cpu()
        assert ((got[:, 1] - q).abs().max() / q.abs().max()).item() < 1e-4


def test_weight_stationary_kernel_random_geometries():
    """Twelve seeded random (N, H, W) with 64 -> 64 channels wherever the weight-stationary kernel takes the shape: forward
    (statistics) and both data-gradient modes against the halo kernel (two valid bf16 roundings of one fp32 result:
    rel-L2 <= 3e-3, BN sums <= 5e-3) -- image sizes down to 5x5 (a tile spans ten images), odd widths, ragged last tiles."""
    _lib, lib, cnt = _lib_and_counter()
    from convnets_amd import ops
    rng = np.random.default_rng(77)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ran = 0
    for case in range(12):
        H, W = int(rng.integers(5, 60)), int(rng.integers(5, 60))
        N = int(rng.integers(max(1, 2048 // (H * W) + 1), max(2, 40000 // (H * W)) + 2))
        C = 64
        M = N * H * W
        x, dy, w = _operands((N, C, H, W), 500 + case)
        g = torch.Generator().manual_seed(900 + case)
        xg, dyg = x.cuda().contiguous(memory_format=CL), dy.cuda().contiguous(memory_format=CL)
        desc = ops.make_desc(xg.shape, w.shape, (1, 1), (1, 1), _lib.CNK_BF16)
        wf, wd = ops.PackedWeights().get(w.cuda(), desc, True)
        bnx = (torch.randn(N, H, W, C, generator=g) * 1.3 + 0.2).bfloat16().cuda()
        addend = torch.randn(N, H, W, C, generator=g).bfloat16().cuda()
        amask = torch.randint(0, 256, (M * C // 8,), generator=g, dtype=torch.uint8).cuda()
        bmean, binv = (torch.randn(C, generator=g) * 0.1).cuda(), (torch.rand(C, generator=g) + 0.5).cuda()
        bgam, bbeta = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.3).cuda()
        out = {}
        took = False
        for name, env in (('ws', dict(CNK_PP3X3=None, CNK_PP_CFG=None, CNK_PP_FULL='1', CNK_PP_WS=None)), ('halo', dict(CNK_PP3X3='0'))):
            with _lib.switches(**env):
                nsl = lib.cnk_conv2d_fwd_bn_partial_slices(ctypes.byref(desc))
                nb = lib.cnk_conv2d_dgrad_bn_slices(ctypes.byref(desc))
                part = torch.full((nsl, C, 3), float('nan'), device='cuda')
                bpart = torch.full((nb, C, 2), float('nan'), device='cuda')
                fuse = _lib.BnBwdFuse(bnx.data_ptr(), bmean.data_ptr(), binv.data_ptr(), bgam.data_ptr(), bbeta.data_ptr(),
                                      None, bpart.data_ptr(), 1)
                y, dxp, dxf = torch.empty_like(xg), torch.empty_like(dyg), torch.empty_like(dyg)
                c0 = cnt(CNT_PP_WS)
                assert lib.cnk_conv2d_fwd(ctypes.byref(desc), P(xg), P(wf), None, P(y), P(part), None, 0, st) == 0, lib.cnk_last_error()
                assert lib.cnk_conv2d_dgrad(ctypes.byref(desc), P(dyg), P(wd), P(dxp), None, None, None, None, 0, st) == 0, lib.cnk_last_error()
                assert lib.cnk_conv2d_dgrad(ctypes.byref(desc), P(dyg), P(wd), P(dxf), P(addend), P(amask), ctypes.byref(fuse), None, 0, st) == 0, lib.cnk_last_error()
                torch.cuda.synchronize()
                if name == 'ws':
                    took = cnt(CNT_PP_WS) - c0 == 3
                assert torch.isfinite(part).all() and torch.isfinite(bpart).all()
                out[name] = (y, dxp, dxf, _merge_stats(part, M), bpart.double().sum(0).cpu())
        if not took:
            continue                                   # (patch too large for the kernel's buffers, or too few tiles: halo both times)
        ran += 1
        a, b = out['ws'], out['halo']
        for k in range(3):
            assert rel_l2(host(a[k]), host(b[k])) < 3e-3, (case, N, H, W, k)
        assert ((a[3][0] - b[3][0]).abs().max() / b[3][0].abs().max().clamp_min(1e-6)).item() < 5e-3
        assert ((a[3][1] - b[3][1]).abs() / b[3][1]).max().item() < 5e-3
        assert ((a[4] - b[4]).abs().max(0).values / b[4].abs().max(0).values).max().item() < 5e-3, (case, N, H, W)
    assert ran >= 8, 'the weight-stationary kernel took only %d of 12 shapes' % ran


def test_weight_stationary_kernel_fullsize_under_load_is_repeatable():
    """bs 256, 64 -> 64 @56x56 (3136 tiles, twelve or thirteen per workgroup, every CU busy): the weight-stationary kernel's
    hand-counted vmcnt / lgkmcnt waits and its three-buffer patch ring under the conditions that exposed the ping-pong core's
    wait bug (see test_pingpong_full_epilogue_under_load).  Five launches per mode while a streaming kernel runs beside them
    on a second stream: every launch must reproduce the first bit for bit (a wait that passes early shows up as run-to-run
    differences), and the first must match the halo kernel to two valid bf16 roundings."""
    _lib, lib, cnt = _lib_and_counter()
    from convnets_amd import ops
    N, C, H = 256, 64, 56
    M = N * H * H
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, C, H, H, generator=g).bfloat16().cuda().contiguous(memory_format=CL)
    w = torch.randn(C, C, 3, 3, generator=g) * 0.04
    desc = ops.make_desc(x.shape, w.shape, (1, 1), (1, 1), _lib.CNK_BF16)
    wf, wd = ops.PackedWeights().get(w.cuda(), desc, True)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    bnx = (torch.randn(N, H, H, C, generator=g) * 1.3 + 0.2).bfloat16().cuda()
    addend = torch.randn(N, H, H, C, generator=g).bfloat16().cuda()
    amask = torch.randint(0, 256, (M * C // 8,), generator=g, dtype=torch.uint8).cuda()
    bmean, binv = (torch.randn(C, generator=g) * 0.1).cuda(), (torch.rand(C, generator=g) + 0.5).cuda()
    bgam, bbeta = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.3).cuda()
    noise_a = torch.randn(64 << 20, device='cuda')
    noise_b = torch.empty_like(noise_a)
    side = torch.cuda.Stream()

    def run(env, reps, busy):
        outs = []
        with _lib.switches(**env):
            nsl = lib.cnk_conv2d_fwd_bn_partial_slices(ctypes.byref(desc))
            nb = lib.cnk_conv2d_dgrad_bn_slices(ctypes.byref(desc))
            for r in range(reps):
                part = torch.full((nsl, C, 3), float('nan'), device='cuda')
                bpart = torch.full((nb, C, 2), float('nan'), device='cuda')
                fuse = _lib.BnBwdFuse(bnx.data_ptr(), bmean.data_ptr(), binv.data_ptr(), bgam.data_ptr(), bbeta.data_ptr(), None, bpart.data_ptr(), 1)
                y, dxp, dxf = (torch.full_like(x, float('nan')) for _ in range(3))
                if busy:
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        for _ in range(6):
                            torch.add(noise_a, 1.0, out=noise_b)          # 512 MB of HBM traffic per pass beside the convs
                c0 = cnt(CNT_PP_WS)
                assert lib.cnk_conv2d_fwd(ctypes.byref(desc), P(x), P(wf), None, P(y), P(part), None, 0, st) == 0, lib.cnk_last_error()
                assert lib.cnk_conv2d_dgrad(ctypes.byref(desc), P(x), P(wd), P(dxp), None, None, None, None, 0, st) == 0, lib.cnk_last_error()
                assert lib.cnk_conv2d_dgrad(ctypes.byref(desc), P(x), P(wd), P(dxf), P(addend), P(amask), ctypes.byref(fuse), None, 0, st) == 0, lib.cnk_last_error()
                torch.cuda.synchronize()
                outs.append((y, dxp, dxf, part, bpart, cnt(CNT_PP_WS) - c0))
        return outs
    ref = run(dict(CNK_PP3X3='0'), 1, False)[0]
    got = run(dict(CNK_PP3X3=None, CNK_PP_CFG=None, CNK_PP_FULL='1', CNK_PP_WS=None), 5, True)
    first = got[0]
    assert first[5] == 3, 'all three launches belong on the weight-stationary kernel'
    for k in range(3):
        assert torch.isfinite(first[k]).all()
        assert rel_l2(host(first[k]), host(ref[k])) < 3e-3
    assert torch.isfinite(first[3]).all() and torch.isfinite(first[4]).all()
    for o in got[1:]:
        for k in range(5):
            assert torch.equal(o[k], first[k]), 'launches of one kernel on the same operands differ (mode %d)' % k


@pytest.mark.parametrize('dtype', ['bf16', 'f32'])
@pytest.mark.parametrize('N,C,H,W,k,p,relu', [
    (4, 64, 32, 32, 3, 1, 1),          # the ResNet stem pool
    (3, 64, 15, 18, 3, 1, 1),          # odd sizes: cells without a second row / column
    (2, 24, 12, 12, 3, 1, 0),          # no ReLU, scalar-free channel count off the 64 grid
    (4, 128, 16, 16, 2, 0, 1),         # VGG's 2x2 / 2
    (64, 64, 112, 112, 3, 1, 1),       # the stem at batch 64: many slices
])
def test_bn_backward_through_maxpool_with_kept_inputs(N, C, H, W, k, p, relu, dtype):
    """cnk_maxpool2d_fwd_bn_keep + cnk_bn_bwd_maxpool_kept (round 3): the pooling pass also writes the INPUT value at every
    window's arg-max, and the backward's reduction pass then runs over the pooled gradient and those values instead of
    gathering over the whole input.  x_at_max must be exactly x at the position the arg-max byte names; y and idx must
    equal cnk_maxpool2d_fwd_bn's bit for bit; dgamma / dbeta must equal cnk_bn_bwd_maxpool's up to the fp32 summation
    order, dx up to the coefficients that follow from them."""
    _lib, lib, cnt = _lib_and_counter()
    code = _lib.CNK_F32 if dtype == 'f32' else _lib.CNK_BF16
    tdt = torch.float32 if dtype == 'f32' else torch.bfloat16
    s = 2
    g = torch.Generator().manual_seed(31 + C + H)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    x = (torch.randn(N, H, W, C, generator=g) * 1.5 + 0.3).to(tdt).cuda()
    dyp = torch.randn(N, Ho, Wo, C, generator=g).to(tdt).cuda()
    gamma = (torch.rand(C, generator=g) + 0.5).cuda()
    beta = (torch.randn(C, generator=g) * 0.2).cuda()
    M = N * H * W
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    nws = lib.cnk_bn_workspace_bytes(M, C)
    ws = torch.empty(nws // 4 + 4, dtype=torch.float32, device='cuda')
    sm, si = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
    assert lib.cnk_bn_fwd_train(P(x), P(gamma), P(beta), None, None, None, None, None, P(sm), P(si), None, 0, M, C, 1e-5, 0.1,
                                relu, code, P(ws), nws, st) == 0, lib.cnk_last_error()
    scale, shift = ctypes.c_void_p(ws.data_ptr()), ctypes.c_void_p(ws.data_ptr() + 4 * C)
    y0, y1 = (torch.empty(N, Ho, Wo, C, dtype=tdt, device='cuda') for _ in range(2))
    i0, i1 = (torch.full((N, Ho, Wo, C), 255, dtype=torch.uint8, device='cuda') for _ in range(2))
    xm = torch.full((N, Ho, Wo, C), float('nan'), dtype=tdt, device='cuda')
    assert lib.cnk_maxpool2d_fwd_bn(P(x), scale, shift, relu, P(y0), P(i0), N, H, W, C, k, k, s, s, p, p, code, st) == 0, lib.cnk_last_error()
    assert lib.cnk_maxpool2d_fwd_bn_keep(P(x), scale, shift, relu, P(y1), P(i1), P(xm), N, H, W, C, k, k, s, s, p, p, code, st) == 0, lib.cnk_last_error()
    torch.cuda.synchronize()
    assert torch.equal(y0, y1) and torch.equal(i0, i1)
    # x at the arg-max: window (ho, wo), tap byte t -> (ho*s - p + t // k, wo*s - p + t % k)
    t = i1.long()
    hh = (torch.arange(Ho, device='cuda') * s - p).view(1, Ho, 1, 1) + t // k
    wwi = (torch.arange(Wo, device='cuda') * s - p).view(1, 1, Wo, 1) + t % k
    assert (hh >= 0).all() and (hh < H).all() and (wwi >= 0).all() and (wwi < W).all()
    nn_ = torch.arange(N, device='cuda').view(N, 1, 1, 1).expand_as(t)
    cc_ = torch.arange(C, device='cuda').view(1, 1, 1, C).expand_as(t)
    assert torch.equal(xm, x[nn_, hh, wwi, cc_])
    outs = {}
    for name in ('gather', 'kept', 'kept2'):
        dx, dg, db = torch.full_like(x, 3.0), torch.full((C,), 5.0, device='cuda'), torch.full((C,), 5.0, device='cuda')
        if name == 'gather':
            rc = lib.cnk_bn_bwd_maxpool(P(dyp), P(i1), P(x), P(gamma), P(beta), P(sm), P(si), P(dx), P(dg), P(db), 0, N, H, W, C,
                                        k, k, s, s, p, p, relu, code, P(ws), nws, st)
        else:
            rc = lib.cnk_bn_bwd_maxpool_kept(P(dyp), P(i1), P(x), P(xm), P(gamma), P(beta), P(sm), P(si), P(dx), P(dg), P(db), 0,
                                             N, H, W, C, k, k, s, s, p, p, relu, code, P(ws), nws, st)
        assert rc == 0, lib.cnk_last_error()
        torch.cuda.synchronize()
        outs[name] = (dx, dg, db)
    for a, b in zip(outs['kept'], outs['kept2']):
        assert torch.equal(a, b)                                              # deterministic
    (dx0, dg0, db0), (dx1, dg1, db1) = outs['gather'], outs['kept']
    assert ((dg1 - dg0).abs().max() / dg0.abs().max()).item() < (2e-5 if dtype == 'f32' else 2e-5)
    assert ((db1 - db0).abs().max() / db0.abs().max()).item() < 2e-5
    assert rel_l2(host(dx1), host(dx0)) < (1e-5 if dtype == 'f32' else 2e-3)
