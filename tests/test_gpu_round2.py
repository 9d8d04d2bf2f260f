"""GPU parity (-m gpu), round-2 additions:
  * whole-model bf16 pinned on a WELL-CONDITIONED fixture from the reference's classes (tests/golden/models2.npz `resnet26w`),
  * element-wise comparison of the sampled gradients / post-step weights of every whole-model fixture (fp32),
  * BASELINE configs 3 and 5 architectures (VGG-16 src/vggnet.py:33, ResNet-152 src/resnet.py:35): reduced-size trace vs the
    reference fixture + one full-size property step each (ResNet-152 through the hipGraph-captured step),
  * full-size (bs 256) layer checks against the torch-CPU arithmetic for the shape classes that were missing.
"""
import math
import os

import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')
MODELS = np.load(os.path.join(G, 'models.npz'))
M2 = np.load(os.path.join(G, 'models2.npz'))
CL = torch.channels_last


def host(t):
    return t.detach().float().cpu().contiguous().numpy()


def _build(family, kind, in_size, bs, mixed, damp=-1.0, lr=0.01):
    import convnets_amd as ca
    from oracle import torch_ref as tr
    torch.manual_seed(21)
    s = ca.Settings(kind=kind, input_size=in_size, num_classes=10, batch_size=bs, batch_norm=True, dropout_rate=0,
                    mixed_precision=mixed, learning_rate=lr)
    model = ca.MODELS[family](s)
    if damp >= 0:
        tr.damp_residual_branches(model, damp)
    model = model.cuda()
    model.init_optimizer()
    model.train()
    return model


def _sampled(params, attr, n):
    from oracle import torch_ref as tr
    rows = []
    for p in params:
        t = (p.grad if attr == 'grad' else p).detach().reshape(-1)
        rows.append(np.resize(t[tr.sample_indices(t.numel(), n).to(t.device)].float().cpu().numpy(), n))
    return np.stack(rows)


def _samples_close(got, ref, tol):
    """Element-wise: every sampled element within tol of the tensor's sampled magnitude (catches a permuted, mirrored or
    sign-flipped gradient that a norm comparison cannot see)."""
    scale = np.maximum(np.abs(ref).max(axis=1, keepdims=True), 1e-12)
    return float((np.abs(got - ref) / scale).max())


FP32_CASES = [('resnet26', 'resnet', 26, (3, 32, 32)), ('resnet18', 'resnet', 18, (3, 32, 32)), ('vgg11', 'vgg', 11, (3, 32, 32))]


@pytest.mark.parametrize('name,family,kind,in_size', FP32_CASES)
def test_model_gradient_samples_elementwise_fp32(name, family, kind, in_size):
    """tests/golden/make_golden.py stores 16 evenly spaced elements of every parameter's gradient and post-step value
    (grad_samples / post_samples) from the reference's own step; the HIP fp32 path must reproduce them element by element."""
    g = lambda k: MODELS[f'{name}/{k}']
    bs = len(g('targets'))
    model = _build(family, kind, in_size, bs, mixed=False)
    x, y = torch.from_numpy(g('x')), torch.from_numpy(g('targets'))
    model.train_batch((x, y))
    torch.cuda.synchronize()
    n = g('grad_samples').shape[1]
    gref, pref = g('grad_samples'), g('post_samples')
    got_g = _sampled(model.parameters(), 'grad', n)
    err = np.abs(got_g - gref) / np.maximum(np.abs(gref).max(axis=1, keepdims=True), 1e-12)
    eg, out_el = float(err.max()), int((err > 2e-3).sum())
    out_t = int(sum(rel_l2(a, b) > 2e-3 for a, b in zip(got_g, gref)))
    # Adam's first step moves a weight by lr*g/(|g|+eps) ~ lr*sign(g): an element whose gradient is at rounding level may
    # go either way, so the post-step value is compared where the reference gradient is clearly non-zero
    got_p = _sampled(model.parameters(), 'value', n)
    solid = np.abs(gref) > np.maximum(1e-3 * np.abs(gref).max(axis=1, keepdims=True), 1e-6)
    ep = float((np.abs(got_p - pref)[solid] / np.maximum(np.abs(pref), 1e-2)[solid]).max())
    print('%s fp32: %d of %d sampled gradient elements beyond 2e-3 of their tensor (worst %.2e), %d of %d tensors beyond 2e-3 L2-rel; '
          'post-step weight element %.2e (%d of %d elements with a solid gradient)'
          % (name, out_el, err.size, eg, out_t, len(gref), ep, int(solid.sum()), solid.size))
    # A ReLU input within fp32 rounding of zero (every batch has some in the big early tensors: tests/golden/make_golden.py
    # relu_margin keeps them out of the SMALL late-stage tensors only, where one flip would move every upstream gradient)
    # lands on either side in two correct fp32 evaluations; one flip moves ONE channel's BN dgamma / dbeta by ~1/sqrt(count)
    # of its value (measured 2e-2) and nothing else visibly.  So: at most 3 of the ~1000 sampled elements and 2 tensors
    # outside the bound, nothing beyond 0.1 -- a wrong kernel moves whole tensors.
    assert out_el <= 3 and out_t <= 2 and eg < 0.1
    # the Adam step of an element whose gradient flipped sign differs by 2*lr: same allowance
    bad_p = int(((np.abs(got_p - pref) / np.maximum(np.abs(pref), 1e-2))[solid] > 1e-3).sum())
    assert bad_p <= 3 and solid.sum() > 0.5 * solid.size


@pytest.mark.parametrize('name,family,kind,in_size', [('resnet26w', 'resnet', 26, (3, 64, 64)), ('vgg16r', 'vgg', 16, (3, 32, 32)),
                                                      ('resnet152r', 'resnet', 152, (3, 32, 32))])
def test_models2_trace_fp32(name, family, kind, in_size):
    """One reference step of ResNet-26 (well-conditioned variant), VGG-16 and ResNet-152 (reduced input) on the HIP fp32
    path: logits, loss, correct count, gradient norms and samples, post-step weights."""
    g = lambda k: M2[f'{name}/{k}']
    bs = len(g('targets'))
    model = _build(family, kind, in_size, bs, mixed=False, damp=float(g('damp')))
    assert list(model.state_dict().keys()) == list(g('sd_keys'))
    assert sum(p.numel() for p in model.parameters()) == int(g('n_params'))
    x, y = torch.from_numpy(g('x')), torch.from_numpy(g('targets'))
    loss, correct = model.train_batch((x, y))
    torch.cuda.synchronize()
    # VGG-16 / ResNet-152 at 32x32, bs 8 end in 1x1 maps whose BN normalises over 8 values: fp32 summation-order noise is
    # amplified there (the ResNet-152 fixture is damped like resnet26w; undamped, single gradient tensors differ by 20 %
    # between two fp32 implementations); the well-conditioned ResNet-26 has 512 samples per tensor, a worst-of-45,000 statistic
    t_loss, t_norm, t_el = {'resnet26w': (1e-4, 2e-3, 2e-2), 'vgg16r': (1e-4, 5e-3, 5e-2), 'resnet152r': (2e-4, 5e-3, 2e-1)}[name]      # 155 convs deep: measured 1.1e-1 / 4.5e-2 on its worst (small) tensor
    assert abs(float(loss) - float(g('loss'))) <= t_loss * abs(float(g('loss')))
    assert int(correct) == int(g('correct'))
    gn = np.array([float(p.grad.double().norm()) for p in model.parameters()])
    assert rel_l2(gn, g('grad_norms')) < t_norm
    n = g('grad_samples').shape[1]
    got = _sampled(model.parameters(), 'grad', n)
    ref = g('grad_samples')
    err = np.abs(got - ref) / np.maximum(np.abs(ref).max(axis=1, keepdims=True), 1e-12)
    per_tensor = np.array([rel_l2(a, b) for a, b in zip(got, ref)])
    out_el, out_t = float((err > t_el).mean()), int((per_tensor > t_el).sum())
    print('%s fp32: loss rel %.2e, gradient norms %.2e, sampled gradients: %.3f %% of %d elements beyond %.0e of their tensor (worst '
          '%.2e), %d of %d tensors beyond %.0e L2-rel (worst %.2e)'
          % (name, abs(float(loss) - float(g('loss'))) / abs(float(g('loss'))), rel_l2(gn, g('grad_norms')), 100 * out_el, err.size,
             t_el, err.max(), out_t, len(per_tensor), t_el, per_tensor.max()))
    # Element-wise with room for ReLU-mask flips: these nets hold millions of ReLU inputs, a few of them within fp32
    # rounding (1e-6 of the mean magnitude) of zero, and two correct fp32 evaluations (the reference on torch-CPU, this path,
    # or this path with another summation order in one statistics kernel) put those on different sides.  One flip moves ONE
    # channel's BN dgamma/dbeta -- a sum over the few hundred elements of that channel -- by ~1/sqrt(count) of its value
    # (measured: a single element 1.8e-1 off, its tensor 3e-2, everything else < 1e-4) and nothing else visibly.  A wrong
    # kernel moves whole tensors.  So: all but 0.1 % of the sampled elements and all but two tensors inside the bound, no
    # element beyond 0.3, no tensor beyond 0.1 -- except resnet152r, whose bound is that wide already.
    assert out_el <= 1e-3 and out_t <= 2
    assert err.max() < max(0.3, t_el) and per_tensor.max() < max(0.1, t_el)
    pn = np.array([float(p.detach().double().norm()) for p in model.parameters()])
    assert rel_l2(pn, g('post_norms')) < 1e-4


def test_wellconditioned_resnet26_bf16_vs_emulation_and_fp32():
    """bf16 at the whole-model level.  Fixture `resnet26w`: the reference's ResNet-26 at 64x64, bs 32 (no BN reduces over
    fewer than 128 values) with the residual branches' last BN damped x0.1 -- a random-init deep BN ResNet is chaotic
    (every rounding that flips a ReLU mask moves the gradient by ~sqrt(flipped fraction) per layer: the oracle's own
    bf16 emulation is 0.72 L2 away from fp32 on the undamped net at this size, 0.12 on the damped one, 0.08-0.10 with the
    branches switched off entirely), so the honest bounds are
        HIP bf16 vs the oracle's bf16 emulation (same rounding points):  gradient L2-rel <= 8e-2 (measured 6.0e-2), loss <= 2e-3
        HIP bf16 vs the fp32 reference fixture:                          <= what the emulation itself is from fp32, + 3e-2
    and the second is also capped at 1.6e-1 absolutely."""
    from oracle import torch_ref as tr
    from oracle.bf16_emu import emulate_bf16
    g = lambda k: M2[f'resnet26w/{k}']
    bs, in_size, damp = len(g('targets')), (3, 64, 64), float(g('damp'))
    x, y = torch.from_numpy(g('x')), torch.from_numpy(g('targets'))
    torch.set_num_threads(min(16, os.cpu_count() or 1))

    def oracle(emulate):
        torch.manual_seed(21)
        m = tr.MODELS['resnet'](tr.RefSettings(26, in_size, 10, batch_size=bs, dropout_rate=0))
        tr.damp_residual_branches(m, damp)
        if emulate:
            emulate_bf16(m, round_weights=lambda conv: True)
        m.train()
        out = m(x)
        loss = m.cost_function(out, y)
        loss.backward()
        return out.detach(), float(loss.detach()), [p.grad.detach().double() for p in m.parameters()]
    out_f, loss_f, g_f = oracle(False)
    assert abs(loss_f - float(g('loss'))) <= 1e-5 * abs(loss_f)            # the oracle IS the fixture's computation
    out_e, loss_e, g_e = oracle(True)
    cat = lambda gs: torch.cat([t.reshape(-1) for t in gs])
    emu_vs_f32 = float((cat(g_e) - cat(g_f)).norm() / cat(g_f).norm())
    model = _build('resnet', 26, in_size, bs, mixed=True, damp=damp)
    model.optimizer.zero_grad()
    out = model(x.cuda())
    loss = model.cost_function(out, y.cuda())
    loss.backward()
    torch.cuda.synchronize()
    g_h = [p.grad.detach().cpu().double() for p in model.parameters()]
    hip_vs_emu = float((cat(g_h) - cat(g_e)).norm() / cat(g_e).norm())
    hip_vs_f32 = float((cat(g_h) - cat(g_f)).norm() / cat(g_f).norm())
    # the same against the committed fixture (dense samples of the reference's fp32 gradients)
    n = g('grad_samples').shape[1]
    smp = _sampled(model.parameters(), 'grad', n).astype(np.float64)
    hip_vs_fixture = float(np.linalg.norm(smp - g('grad_samples')) / np.linalg.norm(g('grad_samples')))
    worst = max(float((a - b).norm() / b.norm()) for a, b in zip(g_h, g_e))
    print('resnet26w bf16: HIP vs emulation grads %.3e (worst tensor %.3e) logits %.3e loss %.3e | HIP vs fp32 %.3e (sampled fixture %.3e) | '
          'emulation vs fp32 %.3e' % (hip_vs_emu, worst, float((out.detach().float().cpu() - out_e).norm() / out_e.norm()),
                                      abs(float(loss) - loss_e) / abs(loss_e), hip_vs_f32, hip_vs_fixture, emu_vs_f32))
    assert abs(float(loss) - loss_e) <= 2e-3 * abs(loss_e)
    # measured 6.0e-2 (r02, MI355X): the emulation shares the kernels' rounding POINTS, not their summation order (split-K
    # slabs, fused residual-gradient sums), and even the damped net amplifies that residue
    assert hip_vs_emu <= 8e-2
    assert hip_vs_f32 <= emu_vs_f32 + 3e-2 and hip_vs_f32 <= 1.6e-1
    assert hip_vs_fixture <= emu_vs_f32 + 5e-2


@pytest.mark.parametrize('name,family,kind', [('vgg16r', 'vgg', 16), ('resnet152r', 'resnet', 152)])
def test_baseline_architectures_reduced_bf16(name, family, kind):
    """VGG-16 / ResNet-152 on the bf16 path at the fixture's reduced size: loss within 3e-2 of the reference's fp32 step
    (whole-model loss rel <= 2e-2 after one step; these nets end in 1x1 maps normalised over 8 samples)."""
    g = lambda k: M2[f'{name}/{k}']
    bs = len(g('targets'))
    model = _build(family, kind, (3, 32, 32), bs, mixed=True, damp=float(g('damp')))
    loss, correct = model.train_batch((torch.from_numpy(g('x')), torch.from_numpy(g('targets'))))
    torch.cuda.synchronize()
    tol = 3e-2
    print('%s bf16: loss %.4f vs fp32 reference %.4f' % (name, float(loss), float(g('loss'))))
    assert abs(float(loss) - float(g('loss'))) <= tol * abs(float(g('loss')))
    assert bool(torch.isfinite(model.optimizer.flat_g).all()) and bool(torch.isfinite(model.optimizer.flat_p).all())


def test_fullsize_vgg16_fp32_step_properties():
    """BASELINE config 3: VGG-16 (reference variant: BN + GAP head), 224x224, bs 64, fp32.  Size-independent properties of a
    real step: initial loss 64*ln(1000), finite gradients and statistics, positive BN variances, the loss goes down."""
    import convnets_amd as ca
    torch.manual_seed(21)
    s = ca.Settings(kind=16, input_size=(3, 224, 224), num_classes=1000, batch_size=64, mixed_precision=False, learning_rate=1e-3)
    model = ca.VGGNet(s).cuda()
    assert sum(p.numel() for p in model.parameters()) == int(MODELS['vgg16/n_params'])
    model.init_optimizer()
    model.train()
    gen = torch.Generator(device='cuda').manual_seed(21)
    x = torch.randn(64, 3, 224, 224, generator=gen, device='cuda')
    y = torch.randint(0, 1000, (64,), generator=gen, device='cuda')
    before = model.optimizer.flat_p.clone()
    loss1, correct = model.train_batch((x, y))
    torch.cuda.synchronize()
    assert abs(float(loss1) - 64 * math.log(1000)) < 0.08 * 64 * math.log(1000)      # dropout 0.5 on a 4096-wide head
    assert 0 <= int(correct) <= 64
    gflat = model.optimizer.flat_g
    assert bool(torch.isfinite(gflat).all()) and float(gflat.abs().max()) > 0
    for k, b in model.named_buffers():
        assert bool(torch.isfinite(b.float()).all()), k
        if k.endswith('running_var'):
            assert float(b.min()) > 0, k
    assert float((model.optimizer.flat_p - before).abs().max()) > 5e-4
    model.train_batch((x, y))
    loss3, _ = model.train_batch((x, y))
    assert float(loss3) < float(loss1)
    del model
    torch.cuda.empty_cache()


def test_fullsize_resnet152_bf16_graphed_step_properties():
    """BASELINE config 5: ResNet-152, 224x224, bs 128, bf16, the step captured into a hipGraph (GraphedTrainStep) and
    replayed: initial loss 128*ln(1000), finite state, the loss goes down, replays advance the optimizer."""
    import convnets_amd as ca
    torch.manual_seed(21)
    s = ca.Settings(kind=152, input_size=(3, 224, 224), num_classes=1000, batch_size=128, mixed_precision=True, learning_rate=1e-3)
    model = ca.ResNet(s).cuda()
    assert sum(p.numel() for p in model.parameters()) == int(MODELS['resnet152/n_params'])
    model.init_optimizer()
    model.train()
    gen = torch.Generator(device='cuda').manual_seed(21)
    x = torch.randn(128, 3, 224, 224, generator=gen, device='cuda')
    y = torch.randint(0, 1000, (128,), generator=gen, device='cuda')
    step = ca.GraphedTrainStep(model, (x, y))
    assert model.optimizer.step_count == 0                     # constructing the graph does not advance training
    losses = []
    for _ in range(3):
        loss, correct = step.replay()
        losses.append(float(loss))
    torch.cuda.synchronize()
    assert model.optimizer.step_count == 3
    assert abs(losses[0] - 128 * math.log(1000)) < 0.06 * 128 * math.log(1000)
    assert losses[2] < losses[0]
    assert bool(torch.isfinite(model.optimizer.flat_g).all()) and bool(torch.isfinite(model.optimizer.flat_p).all())
    for k, b in model.named_buffers():
        if k.endswith('running_var'):
            assert float(b.min()) > 0 and bool(torch.isfinite(b).all()), k
    del step, model
    torch.cuda.empty_cache()


# (name, N, Cin, H, Cout, k, stride, pad): BASELINE-size shape classes not covered by test_fullsize_layer_vs_cpu_reference
FULLSIZE = [('c1x1s2_256_512_56', 256, 256, 56, 512, 1, 2, 0),
            ('c3x3s2_128_128_56', 256, 128, 56, 128, 3, 2, 1),
            ('stem7x7_3_64_224', 256, 3, 224, 64, 7, 2, 3),
            ('c1x1_2048_512_7', 256, 2048, 7, 512, 1, 1, 0)]


@pytest.mark.parametrize('name,N,Cin,H,Cout,k,stride,pad', FULLSIZE)
def test_fullsize_layers_vs_cpu_reference(name, N, Cin, H, Cout, k, stride, pad):
    """fp32 path at bs 256 against torch-CPU on the same tensors (y, dx, dw): index overflow / tiling / split-K errors of
    the strided 1x1 (stride-parity classes), the strided 3x3, the N=256 stem (row-patch kernels) and a 2048-channel layer."""
    import torch.nn.functional as F
    from convnets_amd import ops
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * (1.0 / math.sqrt(Cin * k * k))
    Ho = (H + 2 * pad - k) // stride + 1
    dy = torch.randn(N, Cout, Ho, Ho, generator=g)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    xr, wr = x.clone().requires_grad_(Cin > 3), w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, stride=stride, padding=pad)
    yr.backward(dy)
    xt = x.cuda().contiguous(memory_format=CL).requires_grad_(Cin > 3)
    wt = w.cuda().requires_grad_(True)
    y = ops.Conv2dFn.apply(xt, wt, None, (stride, stride), (pad, pad), ops.PackedWeights())
    y.backward(dy.cuda().contiguous(memory_format=CL))
    torch.cuda.synchronize()
    assert rel_l2(host(y), yr.detach().numpy()) < 1e-5
    if Cin > 3:                                              # the stem has no data gradient (first layer)
        assert rel_l2(host(xt.grad), xr.grad.numpy()) < 1e-5
    assert rel_l2(host(wt.grad), wr.grad.numpy()) < 5e-5     # up to 3.2 M-term fp32 reductions
    # bf16 path on the same tensors: within bf16 rounding of the fp32 result
    xb = x.cuda().bfloat16().contiguous(memory_format=CL).requires_grad_(Cin > 3)
    wb = w.cuda().requires_grad_(True)
    yb = ops.Conv2dFn.apply(xb, wb, None, (stride, stride), (pad, pad), ops.PackedWeights())
    yb.backward(dy.cuda().bfloat16().contiguous(memory_format=CL))
    torch.cuda.synchronize()
    assert rel_l2(host(yb), yr.detach().numpy()) < 1e-2
    if Cin > 3:
        assert rel_l2(host(xb.grad), xr.grad.numpy()) < 1e-2
    assert rel_l2(host(wb.grad), wr.grad.numpy()) < 1e-2


def test_graph_capture_with_a_live_eager_loss():
    """a caller that still holds the non-detached loss / outputs of an eager step when GraphedTrainStep is
    built keeps that step's autograd graph (and its AccumulateGrad nodes, created on the eager stream) alive.  The
    capture must still succeed (warm-up and capture share one side stream) and replay exactly like a fresh capture."""
    import convnets_amd as ca
    from oracle import torch_ref as tr
    x, y = tr.synthetic_batch(8, (3, 32, 32), 10, seed=5)
    x, y = x.cuda(), y.cuda()
    model = _build('resnet', 18, (3, 32, 32), 8, mixed=True, lr=1e-3)
    model.optimizer.zero_grad()
    out = model(x)                                  # an eager forward whose graph stays alive ...
    held = model.cost_function(out, y)              # ... through this non-detached loss
    held.backward(retain_graph=True)
    model.optimizer.zero_grad()
    snap = model.optimizer.flat_p.clone()
    step = ca.GraphedTrainStep(model, (x, y))
    assert torch.equal(model.optimizer.flat_p, snap)
    l1 = float(step.replay()[0])
    l2 = float(step.replay()[0])
    ref = _build('resnet', 18, (3, 32, 32), 8, mixed=True, lr=1e-3)
    r1 = float(ref.train_batch((x, y))[0])
    r2 = float(ref.train_batch((x, y))[0])
    torch.cuda.synchronize()
    assert (l1, l2) == (r1, r2)
    assert held.requires_grad                       # still alive, never replayed


def test_pool_fusion_marker_follows_module_surgery():
    """the BN+ReLU+MaxPool fusion is keyed on a marker set at the first forward; replacing the pool module
    afterwards (or deep-copying the block) must not leave the old pool applied inside the ConvBlock."""
    import copy
    import torch.nn as nn
    import convnets_amd as ca
    from convnets_amd import layers
    x = torch.randn(4, 3, 32, 32, device='cuda')

    def first_pool(model):
        kids = list(model.features.children())
        return next(i for i, m in enumerate(kids) if isinstance(m, layers.MaxPool2d))
    a = _build('vgg', 11, (3, 32, 32), 4, mixed=False)
    a(x)                                             # links the pools
    i = first_pool(a)
    assert a.features[i - 1].__dict__.get('_next_pool') is not None
    a.features[i] = nn.Identity()                    # surgery AFTER the first forward: no pooling there any more
    b = _build('vgg', 11, (3, 32, 32), 4, mixed=False)
    b.features[first_pool(b)] = nn.Identity()        # the same surgery BEFORE any forward
    ya, yb = a(x), b(x)
    torch.cuda.synchronize()
    assert ya.shape == yb.shape and torch.equal(ya, yb)
    assert a.features[i - 1].__dict__.get('_next_pool') is None          # the stale marker was dropped
    # a deep copy of a linked block used on its own behaves like a plain conv -> BN -> ReLU block
    c = _build('vgg', 11, (3, 32, 32), 4, mixed=False)
    c(x)
    j = first_pool(c)
    blk = copy.deepcopy(c.features[j - 1])
    inp = torch.randn(4, blk[0].in_channels, 16, 16, device='cuda').contiguous(memory_format=CL)
    out = blk(inp)
    assert out.shape[-1] == 16                       # not pooled


@pytest.mark.parametrize('Cin,Cout,k,H', [(128, 128, 3, 28), (64, 64, 3, 56), (256, 512, 1, 14), (192, 128, 3, 6)])
def test_wgrad_bitwise_reproducible_and_vs_fp64(Cin, Cout, k, H):
    """Weight gradients (the all-taps 3x3 kernel of conv_wgrad3.hip, the one-tap split-K kernel, both through the fixed-order
    slab sum): two launches on the same operands agree bit for bit, and the result matches torch-CPU in fp64."""
    import ctypes
    import torch.nn.functional as F
    from convnets_amd import _lib, ops
    lib = _lib.load()
    N = 32
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, Cin, H, H, generator=g)
    dy = torch.randn(N, Cout, H, H, generator=g)
    xb = x.cuda().bfloat16().contiguous(memory_format=CL)
    dyb = dy.cuda().bfloat16().contiguous(memory_format=CL)
    desc = ops.make_desc(xb.shape, (Cout, Cin, k, k), (1, 1), (k // 2, k // 2), _lib.CNK_BF16)
    nws = lib.cnk_conv2d_workspace_bytes(ctypes.byref(desc), 2)
    ws = torch.empty(max(nws, 16), dtype=torch.uint8, device='cuda')
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    outs = []
    for _ in range(2):
        dw = torch.full((Cout, Cin, k, k), 7.0, device='cuda')          # overwritten (accumulate = 0)
        assert lib.cnk_conv2d_wgrad(ctypes.byref(desc), P(dyb), P(xb), P(dw), None, 0, P(ws), nws, st) == 0, lib.cnk_last_error()
        torch.cuda.synchronize()
        outs.append(dw.clone())
    assert torch.equal(outs[0], outs[1])
    ref = torch.nn.grad.conv2d_weight(xb.double().cpu(), (Cout, Cin, k, k), dyb.double().cpu(), padding=k // 2)
    assert rel_l2(host(outs[0]), ref.numpy()) < 2e-6
    acc = torch.ones(Cout, Cin, k, k, device='cuda')
    assert lib.cnk_conv2d_wgrad(ctypes.byref(desc), P(dyb), P(xb), P(acc), None, 1, P(ws), nws, st) == 0
    torch.cuda.synchronize()
    assert rel_l2(host(acc), ref.numpy() + 1.0) < 2e-6


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('N,C,H,W,k,s,p,relu', [
    (8, 64, 32, 32, 3, 2, 1, 1),      # the ResNet stem pool (resnet.py:88-89)
    (3, 24, 15, 13, 3, 2, 1, 1),      # odd sizes, ragged last windows, 3 vector columns
    (4, 128, 16, 16, 2, 2, 0, 1),     # VGG stage end (vggnet.py)
    (2, 6, 9, 11, 3, 2, 0, 0),        # scalar channel path, no padding (rows/columns no window covers), no ReLU
    (2, 16, 12, 12, 3, 3, 1, 1),      # one window per element at most
    (64, 64, 112, 112, 3, 2, 1, 1),   # many slices: exercises the XCD-ordered slice walk and the 4-wide final sums
])
def test_bn_backward_through_maxpool_gather(N, C, H, W, k, s, p, relu, dtype):
    """cnk_bn_bwd_maxpool (BN backward whose incoming gradient is gathered from the pooled gradient and the arg-max bytes)
    against the two-call pair cnk_maxpool2d_bwd + cnk_bn_bwd on the same operands -- fp32: same window order, equal up to
    the order of the slice sums; bf16: only the rounding of the intermediate the pair stores differs -- and against torch
    autograd of  max_pool2d(relu(batch_norm(x)))  in fp64 on the CPU."""
    import ctypes
    import torch.nn.functional as F
    from convnets_amd import _lib
    lib = _lib.load()
    code = _lib.CNK_F32 if dtype == 'f32' else _lib.CNK_BF16
    tdt = torch.float32 if dtype == 'f32' else torch.bfloat16
    g = torch.Generator().manual_seed(11)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    x = (torch.randn(N, H, W, C, generator=g) * 1.5 + 0.3).to(tdt).cuda()
    dyp = torch.randn(N, Ho, Wo, C, generator=g).to(tdt).cuda()
    gamma = (torch.rand(C, generator=g) + 0.5).cuda()
    beta = (torch.randn(C, generator=g) * 0.2).cuda()
    M = N * H * W
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    nws = lib.cnk_bn_workspace_bytes(M, C)
    ws = torch.empty(nws // 4 + 4, dtype=torch.float32, device='cuda')
    sm, si = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
    assert lib.cnk_bn_fwd_train(P(x), P(gamma), P(beta), None, None, None, None, None, P(sm), P(si), None, 0, M, C, 1e-5, 0.1,
                                relu, code, P(ws), nws, st) == 0, lib.cnk_last_error()
    scale, shift = ctypes.c_void_p(ws.data_ptr()), ctypes.c_void_p(ws.data_ptr() + 4 * C)
    y = torch.empty(N, Ho, Wo, C, dtype=tdt, device='cuda')
    idx = torch.empty(N, Ho, Wo, C, dtype=torch.uint8, device='cuda')
    assert lib.cnk_maxpool2d_fwd_bn(P(x), scale, shift, relu, P(y), P(idx), N, H, W, C, k, k, s, s, p, p, code, st) == 0, lib.cnk_last_error()
    # the pair
    da = torch.empty_like(x)
    dx_pair, dg_pair, db_pair = torch.empty_like(x), torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
    assert lib.cnk_maxpool2d_bwd(P(dyp), P(idx), P(da), N, H, W, C, k, k, s, s, p, p, code, st) == 0, lib.cnk_last_error()
    assert lib.cnk_bn_bwd(P(da), P(x), None, None, P(gamma), P(beta), P(sm), P(si), P(dx_pair), None, P(dg_pair), P(db_pair), 0,
                          None, 0, M, C, relu, code, P(ws), nws, st) == 0, lib.cnk_last_error()
    # the gathered form, twice (determinism), then accumulating
    outs = []
    for _ in range(2):
        dx, dg, db = torch.full_like(x, 3.0), torch.full((C,), 5.0, device='cuda'), torch.full((C,), 5.0, device='cuda')
        assert lib.cnk_bn_bwd_maxpool(P(dyp), P(idx), P(x), P(gamma), P(beta), P(sm), P(si), P(dx), P(dg), P(db), 0, N, H, W, C,
                                      k, k, s, s, p, p, relu, code, P(ws), nws, st) == 0, lib.cnk_last_error()
        torch.cuda.synchronize()
        outs.append((dx, dg, db))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    dx, dg, db = outs[0]
    if dtype == 'f32':
        # the 2 x 2-cell kernels of stride-2 pools slice the tensor differently from cnk_bn_bwd: same values up to the
        # fp32 order of the per-slice sums; the element-wise kernels share its slices and agree bit for bit
        assert rel_l2(host(dx), host(dx_pair)) < 2e-6
        assert rel_l2(host(dg), host(dg_pair)) < 2e-6 and rel_l2(host(db), host(db_pair)) < 2e-6
        with _lib.switches(CNK_BN_POOL_ELEMENTWISE='1'):
            ex, eg, eb = torch.empty_like(x), torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
            assert lib.cnk_bn_bwd_maxpool(P(dyp), P(idx), P(x), P(gamma), P(beta), P(sm), P(si), P(ex), P(eg), P(eb), 0, N, H, W,
                                          C, k, k, s, s, p, p, relu, code, P(ws), nws, st) == 0, lib.cnk_last_error()
            torch.cuda.synchronize()
        assert torch.equal(ex, dx_pair) and torch.equal(eg, dg_pair) and torch.equal(eb, db_pair)
    else:
        assert rel_l2(host(dx), host(dx_pair)) < 6e-3
        assert rel_l2(host(dg), host(dg_pair)) < 3e-3 and rel_l2(host(db), host(db_pair)) < 3e-3
    acc_g, acc_b = torch.ones(C, device='cuda'), torch.ones(C, device='cuda')
    assert lib.cnk_bn_bwd_maxpool(P(dyp), P(idx), P(x), P(gamma), P(beta), P(sm), P(si), P(dx), P(acc_g), P(acc_b), 1, N, H, W, C,
                                  k, k, s, s, p, p, relu, code, P(ws), nws, st) == 0, lib.cnk_last_error()
    torch.cuda.synchronize()
    assert torch.equal(acc_g, dg + 1.0) and torch.equal(acc_b, db + 1.0)
    if M * C > 4e6:
        return                                       # the fp64 CPU autograd below is for the small cases
    xr = x.double().cpu().permute(0, 3, 1, 2).requires_grad_(True)
    gr, br = gamma.double().cpu().requires_grad_(True), beta.double().cpu().requires_grad_(True)
    a = F.batch_norm(xr, None, None, gr, br, training=True, eps=1e-5)
    if dtype == 'bf16':                              # the pooled values are the bf16-rounded BN outputs
        a = a + (a.detach().bfloat16().double() - a.detach())
    if relu:
        a = F.relu(a)
    out = F.max_pool2d(a, k, s, p)
    out.backward(dyp.double().cpu().permute(0, 3, 1, 2))
    tol = 1e-5 if dtype == 'f32' else 8e-3
    assert rel_l2(host(dx), xr.grad.permute(0, 2, 3, 1).numpy()) < tol
    assert rel_l2(host(dg), gr.grad.numpy()) < tol and rel_l2(host(db), br.grad.numpy()) < tol


def test_bn_backward_through_maxpool_refuses_wide_overlap():
    """More than 2 x 2 windows over an element (3x3 / stride 1): the entry point refuses instead of dropping windows."""
    import ctypes
    from convnets_amd import _lib
    lib = _lib.load()
    t = torch.zeros(1 * 8 * 8 * 8, device='cuda')
    P = lambda v: ctypes.c_void_p(v.data_ptr())
    nws = lib.cnk_bn_workspace_bytes(64, 8)
    ws = torch.empty(nws // 4 + 4, device='cuda')
    rc = lib.cnk_bn_bwd_maxpool(P(t), P(t), P(t), P(t), P(t), P(t), P(t), P(t), P(t), P(t), 0, 1, 8, 8, 8, 3, 3, 1, 1, 1, 1, 1,
                                _lib.CNK_F32, P(ws), nws, None)
    assert rc != 0 and b'2 x 2' in lib.cnk_last_error()


@pytest.mark.parametrize('dtype', ['bf16', 'f32'])
@pytest.mark.parametrize('N,Cin,H,W,Cout,k,s,p', [
    (6, 3, 64, 64, 64, 7, 2, 3),       # the ResNet stem (resnet.py:88): ring of 9 input rows, 2 new rows per output row
    (5, 3, 33, 32, 64, 7, 2, 3),       # odd height: last windows hang over the bottom edge; rows straddle images
    (4, 3, 40, 40, 64, 3, 1, 1),       # VGG's first conv (vggnet.py): ring of 4, 1 new row
    (3, 3, 67, 64, 48, 7, 4, 3),       # stride 4: ring of 11, 4 new rows (AlexNet's 11x11 exceeds the K' budget: other kernel)
    (3, 4, 31, 32, 64, 5, 2, 2),       # 4 input channels, 5x5 / 2
    (4, 1, 28, 32, 32, 5, 1, 2),       # single-channel input, Cout < 64
    (32, 3, 224, 224, 64, 7, 2, 3),    # bench geometry: 14 output rows per workgroup, incremental refills throughout
])
def test_first_layer_conv_forward_and_bn_partials(N, Cin, H, W, Cout, k, s, p, dtype):
    """The row-patch forward (conv_rowpatch.hip: rolling ring of input rows, next rows prefetched under the MFMA loop) against
    torch-CPU conv2d on the same (rounded) operands, with the BN statistics partials it emits per workgroup merged (Chan) and
    compared with the statistics of the tensor it wrote."""
    import ctypes
    import torch.nn.functional as F
    from convnets_amd import _lib, ops
    lib = _lib.load()
    code = _lib.CNK_F32 if dtype == 'f32' else _lib.CNK_BF16
    tdt = torch.float32 if dtype == 'f32' else torch.bfloat16
    g = torch.Generator().manual_seed(17)
    x = (torch.randn(N, Cin, H, W, generator=g) + 0.4).to(tdt)
    w = torch.randn(Cout, Cin, k, k, generator=g) * 0.1
    bias = torch.randn(Cout, generator=g)
    xg = x.cuda().contiguous(memory_format=CL)
    desc = ops.make_desc(xg.shape, w.shape, (s, s), (p, p), code)
    pk = ops.PackedWeights()
    wf, _ = pk.get(w.cuda(), desc, False)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    y = torch.empty(N, Cout, Ho, Wo, dtype=tdt, device='cuda').contiguous(memory_format=CL)
    nsl = lib.cnk_conv2d_fwd_bn_partial_slices(ctypes.byref(desc))
    assert (nsl > 0) == (Cout == 64)                         # the statistics ride on the 64-channel copy-out
    part = torch.full((max(nsl, 1), Cout, 3), float('nan'), device='cuda')
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    bg = bias.cuda()
    assert lib.cnk_conv2d_fwd(ctypes.byref(desc), P(xg), P(wf), P(bg), P(y), P(part) if nsl else None, None, 0, st) == 0, lib.cnk_last_error()
    y2 = torch.empty_like(y)
    assert lib.cnk_conv2d_fwd(ctypes.byref(desc), P(xg), P(wf), P(bg), P(y2), None, None, 0, st) == 0, lib.cnk_last_error()
    torch.cuda.synchronize()
    assert torch.equal(y, y2)                                # the statistics epilogue does not change the output
    if N * H * W <= 2e5:
        wr = w.to(tdt).double()
        ref = F.conv2d(x.double(), wr, bias.double(), stride=s, padding=p)
        assert rel_l2(host(y), ref.float().numpy()) < (1e-5 if dtype == 'f32' else 4e-3)
    if not nsl:
        assert lib.cnk_conv2d_fwd(ctypes.byref(desc), P(xg), P(wf), P(bg), P(y), P(part), None, 0, st) != 0      # refused, not ignored
        return
    pt = part.double().cpu()
    assert torch.isfinite(pt).all()
    n, mean, m2 = pt[:, :, 0], pt[:, :, 1], pt[:, :, 2]
    tn = n.sum(0)
    assert torch.all(tn == N * Ho * Wo)
    tmean = (n * mean).sum(0) / tn
    tvar = (m2.sum(0) + (n * (mean - tmean) ** 2).sum(0)) / tn
    yd = y.double().cpu()
    assert ((tmean - yd.mean((0, 2, 3))).abs().max() / yd.abs().mean()).item() < 2e-6
    assert ((tvar - yd.var((0, 2, 3), unbiased=False)).abs() / tvar).max().item() < 2e-5


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('shape', [(5, 3, 32, 32), (2, 1, 28, 28), (3, 4, 6, 10), (2, 2, 8, 8), (2, 3, 5, 7), (3, 5, 8, 8),
                                   (64, 3, 224, 224)])
def test_input_layout_handoff(shape, dtype):
    """cnk_nchw_to_nhwc (the NCHW fp32 batch -> NHWC `dtype` hand-off, basemodel.py:185-188): the four-pixels-per-thread
    kernel for <= 4 planes with H*W % 4 == 0, and the element-wise one otherwise (odd H*W, 5 planes): bit-exact."""
    from convnets_amd import ops
    tdt = torch.float32 if dtype == 'f32' else torch.bfloat16
    g = torch.Generator().manual_seed(9)
    x = torch.randn(shape, generator=g)
    out = ops.nchw_to_nhwc(x.cuda(), tdt)
    torch.cuda.synchronize()
    assert out.shape == x.shape and out.dtype == tdt and out.is_contiguous(memory_format=CL)
    assert torch.equal(out.cpu(), x.to(tdt))


@pytest.mark.parametrize('N,Cin,Cout,H,bias', [
    (16, 1024, 256, 14, False),     # 13 row tiles of 256 (the last one ragged: 3136 rows), 256 x 256 and 256 x 128 tiles
    (8, 2048, 512, 7, True),        # 392 rows: two row tiles, the second 136 rows; bias
    (4, 64, 128, 9, False),         # the shortest K the kernel takes (two 32-channel steps: the ring never fills)
    (2, 96, 384, 5, True),          # K = 3 steps, Nout = 3 x 128 (not a multiple of 256: only the 256 x 128 tile fits)
    (256, 1024, 256, 14, False),    # bench size: what the dispatch picks by itself
])
def test_dense_1x1_ring_gemm_matches_the_implicit_gemm_kernel(N, Cin, Cout, H, bias):
    """conv_gemm1x1.hip (256 x 128 / 256 x 256 tiles, four-stage LDS-DMA ring) against igemm_gather_kernel on the same
    operands -- forward with bias and BN statistics partials, plain data gradient: the outputs are bit-identical (same MFMA
    order along K), the statistics agree after merging -- and against torch fp32 on the CPU for the small cases."""
    import ctypes
    import torch.nn.functional as F
    from convnets_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(23)
    x = torch.randn(N, Cin, H, H, generator=g).bfloat16()
    dy = torch.randn(N, Cout, H, H, generator=g).bfloat16()
    w = torch.randn(Cout, Cin, 1, 1, generator=g) * 0.05
    b = torch.randn(Cout, generator=g).cuda() if bias else None
    xg, dyg = x.cuda().contiguous(memory_format=CL), dy.cuda().contiguous(memory_format=CL)
    desc = ops.make_desc(xg.shape, w.shape, (1, 1), (0, 0), _lib.CNK_BF16)
    wf, wd = ops.PackedWeights().get(w.cuda(), desc, True)
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    M = N * H * H
    res, fused = {}, {}
    bnx = (torch.randn(N, H, H, Cin, generator=g) * 1.3 + 0.2).bfloat16().cuda()
    addend = torch.randn(N, H, H, Cin, generator=g).bfloat16().cuda()
    amask = torch.randint(0, 256, (N * H * H * Cin // 8,), generator=g, dtype=torch.uint8).cuda()
    bmean, binv = (torch.randn(Cin, generator=g) * 0.1).cuda(), (torch.rand(Cin, generator=g) + 0.5).cuda()
    bgam, bbeta = (torch.rand(Cin, generator=g) + 0.5).cuda(), (torch.randn(Cin, generator=g) * 0.3).cuda()
    for mode in ('0', '2', '4', '1'):
        with _lib.switches(CNK_GEMM1X1=mode):
            nsl = lib.cnk_conv2d_fwd_bn_partial_slices(ctypes.byref(desc))
            assert nsl > 0
            part = torch.full((nsl, Cout, 3), float('nan'), device='cuda')
            y, dx = torch.empty_like(dyg), torch.empty_like(xg)
            assert lib.cnk_conv2d_fwd(ctypes.byref(desc), P(xg), P(wf), P(b), P(y), P(part), None, 0, st) == 0, lib.cnk_last_error()
            assert lib.cnk_conv2d_dgrad(ctypes.byref(desc), P(dyg), P(wd), P(dx), None, None, None, None, 0, st) == 0, lib.cnk_last_error()
            # the data gradient as the training step issues it: + gradient addend gated by a packed ReLU mask, and the
            # first pass of the backward of the BN + ReLU layer that produced x (mask re-derived from bnx)
            nb = lib.cnk_conv2d_dgrad_bn_slices(ctypes.byref(desc))
            assert nb > 0
            bpart = torch.full((nb, Cin, 2), float('nan'), device='cuda')
            fuse = _lib.BnBwdFuse(bnx.data_ptr(), bmean.data_ptr(), binv.data_ptr(), bgam.data_ptr(), bbeta.data_ptr(), None, bpart.data_ptr(), 1)
            dxf = torch.empty_like(xg)
            assert lib.cnk_conv2d_dgrad(ctypes.byref(desc), P(dyg), P(wd), P(dxf), P(addend), P(amask), ctypes.byref(fuse), None, 0, st) == 0, lib.cnk_last_error()
            torch.cuda.synchronize()
            assert torch.isfinite(bpart).all()
            fused[mode] = (dxf, bpart.double().sum(0))
            pt = part.double()
            assert torch.isfinite(pt).all()
            n, mean, m2 = pt[:, :, 0], pt[:, :, 1], pt[:, :, 2]
            tn = n.sum(0)
            assert torch.all(tn == M)
            tmean = (n * mean).sum(0) / tn
            tvar = (m2.sum(0) + (n * (mean - tmean) ** 2).sum(0)) / tn
            res[mode] = (y, dx, nsl, tmean, tvar)
    assert res['2'][2] == (M + 255) // 256 and res['0'][2] == (M + 127) // 128      # one record per row tile of either kernel
    for mode in ('2', '4', '1'):
        assert torch.equal(res[mode][0], res['0'][0]) and torch.equal(res[mode][1], res['0'][1])
        assert torch.equal(fused[mode][0], fused['0'][0])            # masked gradient: same element-wise arithmetic
        assert ((fused[mode][1] - fused['0'][1]).abs().max() / fused['0'][1].abs().max()).item() < 1e-5      # sum dx', sum dx' (x - mean) invstd
        assert ((res[mode][3] - res['0'][3]).abs().max() / res['0'][0].float().abs().mean()).item() < 1e-6
        assert ((res[mode][4] - res['0'][4]).abs() / res['0'][4]).max().item() < 1e-5
    yd = res['2'][0].double()
    assert ((res['2'][3] - yd.mean((0, 2, 3))).abs().max() / yd.abs().mean()).item() < 2e-6      # statistics of the tensor it wrote
    assert ((res['2'][4] - yd.var((0, 2, 3), unbiased=False)).abs() / res['2'][4]).max().item() < 2e-5
    if M * Cin * Cout <= 2e9:
        ref = F.conv2d(x.double(), w.bfloat16().double(), None if b is None else b.double().cpu())
        assert rel_l2(host(res['2'][0]), ref.float().numpy()) < 4e-3
        refdx = F.conv_transpose2d(dy.double(), w.bfloat16().double())
        assert rel_l2(host(res['2'][1]), refdx.float().numpy()) < 4e-3
