"""Generates tests/golden/*.npz from the REFERENCE's own classes (run in the build container only).

    python tests/golden/make_golden.py          # needs /root/reference (read-only)

The reference (milan-pavlovic-ai/ConvNets) is pure Python over torch.nn and has no tests or golden
vectors of its own, so the vectors that pin parity are produced here by importing its
modules from /root/reference/src and running the layers its own factory builds
(basemodel.py:966-1042), its residual blocks (resnet.py:119-196) and whole models for one
training step (basemodel.py:245-288) on torch-CPU.  Three packages the reference imports but the
hot path never touches (torchsummary, torchvision, seaborn) are absent from the image and are
replaced by empty module objects; ReduceLROnPlateau(verbose=...) (basemodel.py:78) is tolerated.
Only DATA is written (inputs, outputs, sampled gradients, norms) -- never reference source text.

Provenance recorded in each file: torch version/config string, seed.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/src'


def import_reference():
    for name, attrs in [('torchsummary', ['summary']), ('torchvision', []),
                        ('torchvision.datasets', []), ('torchvision.transforms', []),
                        ('seaborn', ['set'])]:
        m = types.ModuleType(name)
        for a in attrs:
            setattr(m, a, lambda *args, **kw: None)
        sys.modules.setdefault(name, m)
    sys.modules['torchvision'].datasets = sys.modules['torchvision.datasets']
    sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import basemodel
    import densenet
    import inceptionnet_v1
    import mobilenet_v1
    import resnet
    import resnext
    import se_resnet
    import senet
    import shufflenet_v1
    import sk_resnet
    import sknet
    import squeezenet
    import settings
    import vggnet
    from torch.optim.lr_scheduler import ReduceLROnPlateau as _RLP

    class _RLPCompat(_RLP):
        def __init__(self, *a, verbose=None, **k):
            super().__init__(*a, **k)

    basemodel.ReduceLROnPlateau = _RLPCompat
    basemodel.mobilenet_v1 = mobilenet_v1          # carried along for make_models / make_dwblocks
    basemodel.resnext = resnext
    basemodel.inceptionnet_v1 = inceptionnet_v1
    basemodel.squeezenet = squeezenet
    basemodel.densenet = densenet
    basemodel.se_resnet = se_resnet
    basemodel.senet = senet
    basemodel.shufflenet_v1 = shufflenet_v1
    basemodel.sknet = sknet
    basemodel.sk_resnet = sk_resnet
    return settings, basemodel, resnet, vggnet


def meta():
    return dict(torch_version=np.array(torch.__version__), seed=np.array(21))


# (name, N, Cin, H, W, Cout, k, stride, pad, batch_norm) -- batch_norm=False => conv has a bias
CONV_CASES = [
    ('c3x3s1', 2, 64, 14, 14, 64, 3, 1, 1, True),
    ('c3x3s2', 2, 64, 15, 15, 128, 3, 2, 1, True),
    ('c3x3s2e', 2, 64, 16, 16, 64, 3, 2, 1, True),
    ('c1x1s1', 2, 64, 8, 8, 256, 1, 1, 0, True),
    ('c1x1s2', 2, 128, 14, 14, 256, 1, 2, 0, True),
    ('c1x1s2o', 2, 64, 7, 7, 64, 1, 2, 0, True),
    ('stem7x7', 2, 3, 32, 32, 64, 7, 2, 3, True),
    ('vggc1', 2, 3, 16, 16, 64, 3, 1, 1, True),
    ('cn3x3b', 2, 1, 28, 28, 32, 3, 1, 0, False),
    ('cn5x5b', 2, 32, 13, 13, 64, 5, 2, 1, False),
    ('lenet5x5', 2, 6, 14, 14, 16, 5, 1, 0, True),
    ('alex11', 1, 3, 67, 67, 64, 11, 4, 2, True),
    ('alex5', 1, 64, 7, 7, 64, 5, 1, 2, True),
    ('c3x3_192', 1, 192, 6, 6, 128, 3, 1, 1, True),
]

# (name, N, C, H, W, k, stride, pad, activation)
BLOCK_CASES = [
    ('blk3x3', 4, 64, 10, 10, 64, 3, 1, 1, True),
    ('blk1x1na', 4, 64, 6, 6, 128, 1, 1, 0, False),
    ('blkstem', 2, 3, 20, 20, 64, 7, 2, 3, True),
]

POOL_CASES = [('mp3s2p1', 2, 64, 16, 16, 3, 2, 1), ('mp2s2', 2, 64, 14, 14, 2, 2, 0),
              ('mp3s2', 2, 64, 13, 13, 3, 2, 0), ('mp2s2odd', 2, 8, 13, 13, 2, 2, 0)]


def make_ops(settings, basemodel):
    out = meta()

    def probe(batch_norm, in_size):
        s = settings.Settings(kind=0, input_size=in_size, num_classes=10, batch_norm=batch_norm,
                              mixed_precision=False, dropout_rate=0)
        return basemodel.MultiClassBaseModel(s)

    g = torch.Generator().manual_seed(21)
    for name, N, Ci, H, W, Co, k, st, p, bn in CONV_CASES:
        net = probe(bn, (Ci, H, W))
        layer = net.conv2d(num_filters=Co, kernel_size=k, stride=st, padding=p)   # reference factory
        with torch.no_grad():
            layer.weight.copy_(torch.randn(layer.weight.shape, generator=g) * 0.1)
            if layer.bias is not None:
                layer.bias.copy_(torch.randn(Co, generator=g))
        x = torch.randn(N, Ci, H, W, generator=g, requires_grad=True)
        y = layer(x)
        assert (net.in_channels, net.height, net.width) == tuple(y.shape[1:])
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy)
        out[f'conv/{name}/cfg'] = np.array([N, Ci, H, W, Co, k, st, p, int(layer.bias is not None)])
        out[f'conv/{name}/x'] = x.detach().numpy()
        out[f'conv/{name}/w'] = layer.weight.detach().numpy()
        out[f'conv/{name}/dy'] = dy.numpy()
        out[f'conv/{name}/y'] = y.detach().numpy()
        out[f'conv/{name}/dx'] = x.grad.numpy()
        out[f'conv/{name}/dw'] = layer.weight.grad.numpy()
        if layer.bias is not None:
            out[f'conv/{name}/b'] = layer.bias.detach().numpy()
            out[f'conv/{name}/db'] = layer.bias.grad.numpy()

    for name, N, Ci, H, W, Co, k, st, p, act in BLOCK_CASES:
        net = probe(True, (Ci, H, W))
        blk = net.conv2d_block(num_filters=Co, kernel_size=k, stride=st, padding=p, activation=act)
        conv, bn = blk[0], blk[1]
        with torch.no_grad():
            conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.1)
            bn.weight.copy_(torch.rand(Co, generator=g) + 0.5)
            bn.bias.copy_(torch.randn(Co, generator=g) * 0.2)
        blk.train()
        x = torch.randn(N, Ci, H, W, generator=g, requires_grad=True)
        y = blk(x)
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy)
        out[f'block/{name}/cfg'] = np.array([N, Ci, H, W, Co, k, st, p, int(act)])
        for key, val in [('x', x.detach()), ('w', conv.weight.detach()), ('gamma', bn.weight.detach()),
                         ('beta', bn.bias.detach()), ('dy', dy), ('y', y.detach()), ('dx', x.grad),
                         ('dw', conv.weight.grad), ('dgamma', bn.weight.grad), ('dbeta', bn.bias.grad),
                         ('running_mean', bn.running_mean), ('running_var', bn.running_var)]:
            out[f'block/{name}/{key}'] = val.numpy().copy()
        # eval-mode forward with the running statistics just produced (valid_model path)
        blk.eval()
        with torch.no_grad():
            out[f'block/{name}/y_eval'] = blk(x.detach()).numpy()

    for name, N, C, H, W, k, st, p in POOL_CASES:
        net = probe(True, (C, H, W))
        layer = net.maxpool2d(kernel_size=k, stride=st, padding=p)
        x = torch.randn(N, C, H, W, generator=g)
        x = torch.relu(x).requires_grad_(True) if name == 'mp3s2p1' else x.requires_grad_(True)
        y = layer(x)
        assert (net.height, net.width) == tuple(y.shape[2:])
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy)
        out[f'pool/{name}/cfg'] = np.array([N, C, H, W, k, st, p])
        out[f'pool/{name}/x'] = x.detach().numpy()
        out[f'pool/{name}/dy'] = dy.numpy()
        out[f'pool/{name}/y'] = y.detach().numpy()
        out[f'pool/{name}/dx'] = x.grad.numpy()

    # global average pool through the reference factory
    net = probe(True, (32, 7, 7))
    layer = net.adapt_avgpool2d(output_size=1)
    x = torch.randn(3, 32, 7, 7, generator=g, requires_grad=True)
    y = layer(x)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    out['gap/x'], out['gap/dy'] = x.detach().numpy(), dy.numpy()
    out['gap/y'], out['gap/dx'] = y.detach().numpy(), x.grad.numpy()
    np.savez_compressed(os.path.join(HERE, 'ops.npz'), **out)
    print('ops.npz', len(out), 'arrays')


def make_resblocks(settings, resnet):
    """ResBottleneck / ResBlock (resnet.py:119-196) with and without the projection shortcut."""
    out = meta()
    g = torch.Generator().manual_seed(22)
    # (name, block, Cin, H, num_filters, expansion, stride)
    cases = [('bott_proj_s2', 'bottleneck', 64, 8, 32, 4, 2), ('bott_id', 'bottleneck', 128, 6, 32, 4, 1),
             ('bott_proj_s1', 'bottleneck', 64, 6, 64, 4, 1), ('basic_proj_s2', 'basic', 64, 8, 128, 1, 2),
             ('basic_id', 'basic', 64, 6, 64, 1, 1)]
    for name, kind, Ci, H, nf, exp, stride in cases:
        s = settings.Settings(kind=50, input_size=(Ci, H, H), num_classes=10, mixed_precision=False,
                              dropout_rate=0)
        net = resnet.ResNet.__new__(resnet.ResNet)
        import basemodel
        basemodel.MultiClassBaseModel.__init__(net, s)
        block_type = resnet.ResBottleneck if kind == 'bottleneck' else resnet.ResBlock
        blk = net.residual_block(block_type, nf, exp, stride)     # reference wiring :85-98
        with torch.no_grad():
            for m in blk.modules():
                if isinstance(m, nn.Conv2d):
                    m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / m.weight[0].numel()) ** 0.5)
                elif isinstance(m, nn.BatchNorm2d):
                    m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                    m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.2)
        blk.train()
        N = 4
        x = torch.randn(N, Ci, H, H, generator=g, requires_grad=True)
        y = blk(x)
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy)
        out[f'{name}/cfg'] = np.array([N, Ci, H, nf, exp, stride, int(kind == 'bottleneck')])
        out[f'{name}/x'], out[f'{name}/dy'] = x.detach().numpy(), dy.numpy()
        out[f'{name}/y'], out[f'{name}/dx'] = y.detach().numpy(), x.grad.numpy()
        for k, v in blk.state_dict().items():
            out[f'{name}/sd/{k}'] = v.numpy().copy()          # weights before + running stats after
        for k, p in blk.named_parameters():
            out[f'{name}/grad/{k}'] = p.grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'resblocks.npz'), **out)
    print('resblocks.npz', len(out), 'arrays')


def sample(t, n=16):
    f = t.detach().reshape(-1)
    idx = torch.linspace(0, f.numel() - 1, min(n, f.numel())).long()
    return f[idx].numpy().copy()


def relu_margin(model, x, small=32768):
    """Smallest |ReLU input| of one training-mode forward, relative to that tensor's mean |input|, over the ReLU inputs
    with at most `small` elements.  A pre-activation within fp32 rounding of zero makes the step's gradients depend on the
    summation order of whoever evaluates it (the mask flips; in a 16 x 256 x 2 x 2 tensor ONE flip moves the layer's
    gradients by 1e-2), so such a batch pins nothing element-wise: model_case() draws another batch instead."""
    import torch.nn.functional as F
    worst = [float('inf')]
    orig = F.relu

    def spy(inp, inplace=False):
        if inp.numel() <= small:
            a = inp.detach().abs()
            worst[0] = min(worst[0], float(a.min() / a.mean()))
        return orig(inp, inplace=False)
    F.relu = spy
    try:
        state = {k: v.clone() for k, v in model.state_dict().items()}
        with torch.no_grad():
            model(x)
        model.load_state_dict(state)                     # the probe must not advance the BN running statistics
    finally:
        F.relu = orig
    return worst[0]


MODEL_CASES = None


def model_cases(basemodel, resnet, vggnet):
    # batch 16: the last ResNet stage is 1x1 spatial at 32x32 input, so BN there sees only `batch`
    # samples per channel; with fewer the normalised activations are too ill-conditioned to compare
    return [('resnet26', resnet.ResNet, 26, (3, 32, 32), True, 16),
            ('resnet18', resnet.ResNet, 18, (3, 32, 32), True, 16),
            ('vgg11', vggnet.VGGNet, 11, (3, 32, 32), True, 16),
            ('convnet', basemodel.ConvNet, 0, (1, 28, 28), False, 8),
            ('mobilenet', basemodel.mobilenet_v1.MobileNetV1, 1, (3, 32, 32), True, 16),
            ('resnext26', basemodel.resnext.ResNeXt, 26, (3, 32, 32), True, 16),
            ('inception', basemodel.inceptionnet_v1.InceptionNetV1, 1, (3, 32, 32), True, 16),
            ('squeezenet', basemodel.squeezenet.SqueezeNet, 1.1, (3, 64, 64), True, 16),
            ('densenet121', basemodel.densenet.DenseNet, 121, (3, 32, 32), True, 16),
            ('seresnet26', basemodel.se_resnet.SEResNet, 26, (3, 32, 32), True, 16),
            ('senet26', basemodel.senet.SENet, 26, (3, 32, 32), True, 16),
            ('shufflenet_g4', basemodel.shufflenet_v1.ShuffleNetV1, 'g4', (3, 32, 32), True, 16),
            ('sknet26', basemodel.sknet.SKNet, 26, (3, 32, 32), True, 16),
            ('skresnet26', basemodel.sk_resnet.SKResNet, 26, (3, 32, 32), True, 16)]


def model_case(settings, out, name, cls, kind, in_size, bn, bs, data_seeds=(21,), min_margin=0.0):
    """One whole-model one-step trace into out[name/...].  data_seeds: batches to try in order; the first whose
    relu_margin() reaches min_margin is used (recorded as name/data_seed, name/relu_margin)."""
    s = settings.Settings(kind=kind, input_size=in_size, num_classes=10, batch_size=bs,
                          batch_norm=bn, mixed_precision=False, dropout_rate=0, seed=21)
    for data_seed in data_seeds:
        torch.manual_seed(21)                 # weights are re-drawn from this seed by the tests
        model = cls(s)
        model.init_optimizer()
        g = torch.Generator().manual_seed(data_seed)
        x = torch.randn(bs, *in_size, generator=g)
        y = torch.randint(0, 10, (bs,), generator=g)
        model.train()
        margin = relu_margin(model, x)
        if margin >= min_margin:
            break
        print('%-14s data seed %d: smallest |ReLU input| / mean %.2e, next batch' % (name, data_seed, margin))
    else:
        raise RuntimeError('%s: no batch with ReLU margin >= %g among seeds %s' % (name, min_margin, data_seeds))
    out[f'{name}/data_seed'], out[f'{name}/relu_margin'] = np.array(data_seed), np.array(margin)
    out[f'{name}/x'], out[f'{name}/targets'] = x.numpy(), y.numpy()
    out[f'{name}/param_names'] = np.array([k for k, _ in model.named_parameters()])
    out[f'{name}/init_norms'] = np.array([float(p.detach().double().norm()) for p in model.parameters()])
    # the reference's own step: basemodel.py:247 zero_grad, :275 training_step, :278 backward, :284 step
    model.optimizer.zero_grad()
    model.setting.device.move = lambda b: b            # CPU: DeviceMngr.move is the identity here
    outputs = model(x)
    loss = model.cost_function(outputs, y)
    correct = model.score(outputs, y)
    loss.backward()
    out[f'{name}/logits'] = outputs.detach().numpy()
    out[f'{name}/loss'] = np.array(float(loss))
    out[f'{name}/correct'] = np.array(correct)
    out[f'{name}/grad_norms'] = np.array([float(p.grad.double().norm()) for p in model.parameters()])
    out[f'{name}/grad_samples'] = np.stack([np.resize(sample(p.grad), 16) for p in model.parameters()])
    model.optimizer.step()
    out[f'{name}/post_norms'] = np.array([float(p.detach().double().norm()) for p in model.parameters()])
    out[f'{name}/post_samples'] = np.stack([np.resize(sample(p), 16) for p in model.parameters()])
    bufs = [(k, b) for k, b in model.named_buffers() if 'running' in k]
    out[f'{name}/buf_names'] = np.array([k for k, _ in bufs])
    out[f'{name}/buf_norms'] = np.array([float(b.double().norm()) for _, b in bufs])
    out[f'{name}/n_params'] = np.array(sum(p.numel() for p in model.parameters()))
    out[f'{name}/sd_keys'] = np.array(list(model.state_dict().keys()))
    # second step's loss pins the optimizer update through the whole network
    model.optimizer.zero_grad()
    loss2 = model.cost_function(model(x), y)
    out[f'{name}/loss2'] = np.array(float(loss2))
    print('%-14s data seed %d, smallest |ReLU input| / mean %.2e' % (name, data_seed, margin))


# Cases whose seed-21 batch has a ReLU input within fp32 rounding of zero in a small late-stage tensor (relu_margin):
# they take the first later batch that keeps 5e-6 of the mean magnitude clear in every ReLU input of <= 32768 elements
# (fp32 evaluations of these nets agree to ~1e-6 of it).  resnet18: 1.5e-6 at the output ReLU of the first 256-channel
# block (16 x 256 x 2 x 2), flipped by any evaluation whose rounding differs by 13 ulp.
REDRAWN = {k: dict(data_seeds=tuple(range(21, 1300, 100)), min_margin=5e-6) for k in ('resnet26', 'resnet18')}   # resnet26
# keeps seed 21 (1.2e-5).  vgg11 is NOT in the list: its tail is 512 x 1 x 1 (BN over the 16 samples of the batch), where this
# path and torch-CPU differ by ~1e-5 of the magnitude, so no margin a batch offers protects it (seed 121, margin 9.6e-6, flipped
# and moved 25 of 30 gradient tensors by 1-2e-2); it stays on the seed-21 batch, which both evaluate alike.


def remake_model_case(settings, basemodel, resnet, vggnet, name):
    """Regenerates ONE case of models.npz in place (the other arrays are carried over byte for byte)."""
    path = os.path.join(HERE, 'models.npz')
    old = np.load(path)
    out = {k: old[k] for k in old.files if not k.startswith(name + '/')}
    case = next(c for c in model_cases(basemodel, resnet, vggnet) if c[0] == name)
    model_case(settings, out, *case, **REDRAWN.get(name, {}))
    np.savez_compressed(path, **out)
    print('models.npz: case %s rewritten (%d arrays)' % (name, len(out)))


def make_models(settings, basemodel, resnet, vggnet):
    """Whole-model one-step traces: (logits, loss, correct, per-parameter grad norm + samples,
    post-Adam-step parameter norm + samples, BN running-stat norms) on the reference classes."""
    out = meta()
    for case in model_cases(basemodel, resnet, vggnet):
        model_case(settings, out, *case, **REDRAWN.get(case[0], {}))
    # architecture facts of the BASELINE configs (no weights): state_dict keys + parameter counts
    for name, cls, kind in [('resnet50', resnet.ResNet, 50), ('resnet152', resnet.ResNet, 152),
                            ('vgg16', vggnet.VGGNet, 16)]:
        s = settings.Settings(kind=kind, input_size=(3, 224, 224), num_classes=1000,
                              mixed_precision=False, init_params=False)
        m = cls(s)
        out[f'{name}/n_params'] = np.array(sum(p.numel() for p in m.parameters()))
        out[f'{name}/sd_keys'] = np.array(list(m.state_dict().keys()))
        out[f'{name}/sd_shapes'] = np.array([str(tuple(v.shape)) for v in m.state_dict().values()])
    np.savez_compressed(os.path.join(HERE, 'models.npz'), **out)
    print('models.npz', len(out), 'arrays')


def make_dwblocks(settings, basemodel):
    """Depthwise-separable blocks built by the reference's own factory (mobilenet_v1.Conv2dBlockDW on a
    MultiClassBaseModel): forward output, input gradient, every parameter gradient, running statistics."""
    out = meta()
    DW = basemodel.mobilenet_v1.Conv2dBlockDW
    for name, C, F_, H, stride, bn in [('dw64s1', 64, 128, 14, 1, True), ('dw128s2', 128, 256, 13, 2, True),
                                       ('dw32nobn', 32, 64, 9, 1, False)]:
        s = settings.Settings(kind=0, input_size=(C, H, H), num_classes=10, batch_size=4, batch_norm=bn,
                              mixed_precision=False, dropout_rate=0, seed=21)
        net = basemodel.MultiClassBaseModel(s)
        torch.manual_seed(21)
        blk = DW(net, num_filters=F_, kernel_size=3, stride=stride, padding=1)
        net.block = blk
        net.init_params()
        g = torch.Generator().manual_seed(5)
        with torch.no_grad():
            for m in blk.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                    m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.2)
                if isinstance(m, nn.Conv2d) and m.bias is not None:
                    m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
        x = torch.randn(4, C, H, H, generator=g).requires_grad_(True)
        blk.train()
        y = blk(x)
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy)
        out[f'{name}/cfg'] = np.array([C, F_, H, stride, int(bn)])
        out[f'{name}/x'], out[f'{name}/y'], out[f'{name}/dy'], out[f'{name}/dx'] = \
            x.detach().numpy(), y.detach().numpy(), dy.numpy(), x.grad.numpy()
        for k, p in blk.named_parameters():
            out[f'{name}/param/{k}'] = p.detach().numpy()
            out[f'{name}/grad/{k}'] = p.grad.numpy()
        for k, b in blk.named_buffers():
            if 'running' in k:
                out[f'{name}/buf/{k}'] = b.numpy()
    np.savez_compressed(os.path.join(HERE, 'dwblocks.npz'), **out)
    print('dwblocks.npz', len(out), 'arrays')


def make_affine():
    """RandomAffine(degrees=15, shear=15, scale=(0.75, 1.25)) of src/mngrdata.py:148 on PIL images, as torchvision
    applies it: Image.transform(size, AFFINE, matrix, NEAREST, fillcolor=0) -- the expected outputs below come from
    Pillow itself; the matrix from the restatement of torchvision's _get_inverse_affine_matrix (torchvision is not
    installed here), stored next to them."""
    from PIL import Image
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import transforms_ref as tf
    rng = np.random.RandomState(21)
    out = meta()
    out['pillow_version'] = np.array(__import__('PIL').__version__)
    sizes = [(32, 32)] * 10 + [(24, 40), (40, 24), (17, 33), (64, 64), (7, 5), (224, 224)]
    for i, (H, W) in enumerate(sizes):
        img = rng.randint(0, 256, (H, W, 3)).astype(np.uint8)
        angle, scale, shear = rng.uniform(-15, 15), rng.uniform(0.75, 1.25), rng.uniform(-15, 15)
        m = tf.tv_inverse_affine_matrix([W * 0.5, H * 0.5], angle, [0.0, 0.0], scale, [shear, 0.0])
        ref = np.asarray(Image.fromarray(img).transform((W, H), Image.AFFINE, m, Image.NEAREST, fillcolor=0))
        out[f'case{i}/img'], out[f'case{i}/draw'], out[f'case{i}/matrix'] = img, np.array([angle, scale, shear]), np.array(m)
        out[f'case{i}/out'] = ref
    out['n_cases'] = np.array(len(sizes))
    np.savez_compressed(os.path.join(HERE, 'affine.npz'), **out)
    print('affine.npz', len(out), 'arrays')


def damp_residual_branches(model, factor):
    """Scales gamma of the LAST BatchNorm of every residual branch (resnet.py:130-135 / :169-175).  A randomly
    initialised deep BN ResNet is in the chaotic phase: any perturbation of a pre-activation that flips a ReLU mask
    changes its gradients by ~sqrt(flipped fraction) per layer, so bf16 rounding alone moves whole-model gradients by
    0.3-0.8 L2 (oracle/bf16_emu.py vs fp32, any input size or batch -- measured 0.72 for ResNet-26 at 64x64, bs 32).
    Damped branches (a trained network's regime: near-identity blocks) bring that to ~0.1, where a whole-model bf16
    comparison means something.  Returns the scaled modules."""
    out = []
    with torch.no_grad():
        for m in model.modules():
            seq = getattr(m, 'res_bottleneck', None) or getattr(m, 'res_block', None)
            if seq is not None:
                bn = [c for c in seq[-1].modules() if isinstance(c, nn.BatchNorm2d)][-1]
                bn.weight.mul_(factor)
                out.append(bn)
    return out


def make_models2(settings, resnet, vggnet):
    """Round-2 whole-model traces (models2.npz; models.npz stays byte-identical):
      resnet26w  -- ResNet-26 bottleneck, 64x64, bs 32, residual branches damped x0.1: the well-conditioned fixture
                    for the bf16 path (every BN reduces over >= 128 values; dense gradient samples)
      vgg16r / resnet152r -- the BASELINE configs 3 and 5 architectures (src/vggnet.py:33, src/resnet.py:35) at reduced
                    input size and batch, one reference step each (ResNet-152 with the same x0.1 damping: undamped, its 50
                    blocks amplify fp32 summation-order noise to 20 % in single gradient tensors)."""
    out = meta()
    cases = [('resnet26w', resnet.ResNet, 26, (3, 64, 64), 32, 0.1, 512),
             ('vgg16r', vggnet.VGGNet, 16, (3, 32, 32), 8, None, 64),
             ('resnet152r', resnet.ResNet, 152, (3, 32, 32), 8, 0.1, 32)]
    for name, cls, kind, in_size, bs, damp, nsamp in cases:
        s = settings.Settings(kind=kind, input_size=in_size, num_classes=10, batch_size=bs, batch_norm=True,
                              mixed_precision=False, dropout_rate=0, seed=21)
        torch.manual_seed(21)
        model = cls(s)
        if damp is not None:
            damp_residual_branches(model, damp)
        model.init_optimizer()
        g = torch.Generator().manual_seed(21)
        x = torch.randn(bs, *in_size, generator=g)
        y = torch.randint(0, 10, (bs,), generator=g)
        model.train()
        out[f'{name}/x'], out[f'{name}/targets'] = x.numpy(), y.numpy()
        out[f'{name}/damp'] = np.array(-1.0 if damp is None else damp)
        out[f'{name}/init_norms'] = np.array([float(p.detach().double().norm()) for p in model.parameters()])
        model.optimizer.zero_grad()
        outputs = model(x)
        loss = model.cost_function(outputs, y)
        correct = model.score(outputs, y)
        loss.backward()
        out[f'{name}/logits'] = outputs.detach().numpy()
        out[f'{name}/loss'] = np.array(float(loss))
        out[f'{name}/correct'] = np.array(correct)
        out[f'{name}/grad_norms'] = np.array([float(p.grad.double().norm()) for p in model.parameters()])
        out[f'{name}/grad_samples'] = np.stack([np.resize(sample(p.grad, nsamp), nsamp) for p in model.parameters()])
        model.optimizer.step()
        out[f'{name}/post_norms'] = np.array([float(p.detach().double().norm()) for p in model.parameters()])
        out[f'{name}/post_samples'] = np.stack([np.resize(sample(p, nsamp), nsamp) for p in model.parameters()])
        out[f'{name}/n_params'] = np.array(sum(p.numel() for p in model.parameters()))
        out[f'{name}/sd_keys'] = np.array(list(model.state_dict().keys()))
        print(name, 'loss', float(loss), 'params', int(out[f'{name}/n_params']))
    np.savez_compressed(os.path.join(HERE, 'models2.npz'), **out)
    print('models2.npz', len(out), 'arrays')


if __name__ == '__main__':
    torch.set_num_threads(8)
    settings, basemodel, resnet, vggnet = import_reference()
    if len(sys.argv) > 1 and sys.argv[1] == 'models2':
        make_models2(settings, resnet, vggnet)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[1] == 'models-case':      # python make_golden.py models-case resnet18
        remake_model_case(settings, basemodel, resnet, vggnet, sys.argv[2])
        sys.exit(0)
    make_ops(settings, basemodel)
    make_resblocks(settings, resnet)
    make_models(settings, basemodel, resnet, vggnet)
    make_dwblocks(settings, basemodel)
    make_affine()
    make_models2(settings, resnet, vggnet)
