"""CPU suite: pins the oracle (oracle/cpu_ops.c and oracle/torch_ref.py) against the golden vectors
that tests/golden/make_golden.py produced from the reference's own classes."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_l2
from oracle import cpu_ops as orc
from oracle import torch_ref as tr

G = os.path.join(os.path.dirname(__file__), 'golden')
OPS = np.load(os.path.join(G, 'ops.npz'))
MODELS = np.load(os.path.join(G, 'models.npz'))
TOL = 2e-6      # fp32 oracle (double accumulation) vs torch-CPU fp32


def _cases(prefix):
    return sorted({k.split('/')[1] for k in OPS.files if k.startswith(prefix + '/')})


@pytest.mark.parametrize('name', _cases('conv'))
def test_c_oracle_conv(name):
    g = lambda k: OPS[f'conv/{name}/{k}']
    N, Ci, H, W, Co, k, st, p, has_b = g('cfg')
    b = g('b') if has_b else None
    assert rel_l2(orc.conv2d_fwd(g('x'), g('w'), b, int(st), int(p)), g('y')) < TOL
    assert rel_l2(orc.conv2d_dgrad(g('dy'), g('w'), (int(H), int(W)), int(st), int(p)), g('dx')) < TOL
    if has_b:
        dw, db = orc.conv2d_wgrad(g('dy'), g('x'), int(k), int(st), int(p), with_bias=True)
        assert rel_l2(db, g('db')) < TOL
    else:
        dw = orc.conv2d_wgrad(g('dy'), g('x'), int(k), int(st), int(p))
    assert rel_l2(dw, g('dw')) < 5e-6


@pytest.mark.parametrize('name', _cases('block'))
def test_c_oracle_conv_bn_relu_block(name):
    g = lambda k: OPS[f'block/{name}/{k}']
    N, Ci, H, W, Co, k, st, p, act = g('cfg')
    c = orc.conv2d_fwd(g('x'), g('w'), None, int(st), int(p))
    rm, rv = np.zeros(Co, np.float32), np.ones(Co, np.float32)
    z, sm, si = orc.bn_fwd_train(c, g('gamma'), g('beta'), rm, rv)
    y = orc.relu_fwd(z) if act else z
    assert rel_l2(y, g('y')) < 5e-6
    assert rel_l2(rm, g('running_mean')) < 5e-6 and rel_l2(rv, g('running_var')) < 5e-6
    dz = orc.relu_bwd(g('dy'), y) if act else g('dy')
    dc, dg, db = orc.bn_bwd(dz, c, g('gamma'), sm, si)
    assert rel_l2(dg, g('dgamma')) < 2e-5 and rel_l2(db, g('dbeta')) < 2e-5
    assert rel_l2(orc.conv2d_wgrad(dc, g('x'), int(k), int(st), int(p)), g('dw')) < 5e-5
    if name != 'blkstem':
        assert rel_l2(orc.conv2d_dgrad(dc, g('w'), (int(H), int(W)), int(st), int(p)), g('dx')) < 5e-5
    ye = orc.bn_fwd_eval(c, g('gamma'), g('beta'), rm, rv)
    assert rel_l2(orc.relu_fwd(ye) if act else ye, g('y_eval')) < 5e-6


@pytest.mark.parametrize('name', _cases('pool'))
def test_c_oracle_maxpool(name):
    g = lambda k: OPS[f'pool/{name}/{k}']
    N, C, H, W, k, st, p = g('cfg')
    y, idx = orc.maxpool_fwd(g('x'), int(k), int(st), int(p))
    assert np.array_equal(y, g('y'))
    assert np.array_equal(orc.maxpool_bwd(g('dy'), idx, (int(H), int(W))), g('dx'))


def test_c_oracle_gap():
    assert rel_l2(orc.gap_fwd(OPS['gap/x']), OPS['gap/y']) < TOL
    assert rel_l2(orc.gap_bwd(OPS['gap/dy'], (7, 7)), OPS['gap/dx']) < TOL


def test_c_oracle_edge_cases():
    # 1x1 image, kernel == padded input, single channel, stride larger than the kernel
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1, 1, 1, 1)).astype(np.float32)
    w = rng.standard_normal((2, 1, 3, 3)).astype(np.float32)
    y = orc.conv2d_fwd(x, w, None, 1, 1)
    assert y.shape == (1, 2, 1, 1) and np.allclose(y[0, :, 0, 0], w[:, 0, 1, 1] * x[0, 0, 0, 0])
    x = rng.standard_normal((2, 3, 7, 7)).astype(np.float32)
    w = rng.standard_normal((4, 3, 2, 2)).astype(np.float32)
    ref = torch.nn.functional.conv2d(torch.from_numpy(x), torch.from_numpy(w), stride=3).numpy()
    assert rel_l2(orc.conv2d_fwd(x, w, None, 3, 0), ref) < TOL


MODEL_CASES = [('resnet26', 'resnet', 26, (3, 32, 32), True), ('resnet18', 'resnet', 18, (3, 32, 32), True),
               ('vgg11', 'vgg', 11, (3, 32, 32), True), ('convnet', 'convnet', 0, (1, 28, 28), False),
               ('mobilenet', 'mobilenet', 1, (3, 32, 32), True), ('resnext26', 'resnext', 26, (3, 32, 32), True),
               ('inception', 'inception', 1, (3, 32, 32), True), ('squeezenet', 'squeezenet', 1.1, (3, 64, 64), True),
               ('densenet121', 'densenet', 121, (3, 32, 32), True), ('seresnet26', 'seresnet', 26, (3, 32, 32), True),
               ('senet26', 'senet', 26, (3, 32, 32), True), ('shufflenet_g4', 'shufflenet', 'g4', (3, 32, 32), True),
               ('sknet26', 'sknet', 26, (3, 32, 32), True), ('skresnet26', 'skresnet', 26, (3, 32, 32), True)]


@pytest.mark.parametrize('name,family,kind,in_size,bn', MODEL_CASES)
def test_torch_restatement_one_step_matches_reference_trace(name, family, kind, in_size, bn):
    """The torch.nn restatement re-draws the reference's weights from the same seed (same module
    registration order => same RNG stream) and must reproduce the reference's one-step trace."""
    g = lambda k: MODELS[f'{name}/{k}']
    s = tr.RefSettings(kind, in_size, 10, batch_size=len(g('targets')), batch_norm=bn, dropout_rate=0)
    torch.manual_seed(21)
    m = tr.MODELS[family](s)
    assert list(m.state_dict().keys()) == list(g('sd_keys'))
    assert [k for k, _ in m.named_parameters()] == list(g('param_names'))
    assert sum(p.numel() for p in m.parameters()) == int(g('n_params'))
    norms = np.array([float(p.detach().double().norm()) for p in m.parameters()])
    np.testing.assert_allclose(norms, g('init_norms'), rtol=1e-6)
    m.init_optimizer()
    x, y = torch.from_numpy(g('x')), torch.from_numpy(g('targets'))
    m.train()
    out = m(x)
    np.testing.assert_allclose(out.detach().numpy(), g('logits'), rtol=1e-4, atol=1e-5)
    loss, correct = tr.ref_training_step(m, x, y)
    assert abs(loss - float(g('loss'))) <= 1e-5 * abs(float(g('loss')))
    assert correct == int(g('correct'))
    post = np.array([float(p.detach().double().norm()) for p in m.parameters()])
    np.testing.assert_allclose(post, g('post_norms'), rtol=1e-4)
    loss2 = float(m.cost_function(m(x), y))
    assert abs(loss2 - float(g('loss2'))) <= 2e-3 * abs(float(g('loss2')))


@pytest.mark.parametrize('name,family,kind', [('resnet50', 'resnet', 50), ('resnet152', 'resnet', 152), ('vgg16', 'vgg', 16)])
def test_baseline_architectures_match_reference(name, family, kind):
    s = tr.RefSettings(kind, (3, 224, 224), 1000, init_params=False)
    m = tr.MODELS[family](s)
    assert list(m.state_dict().keys()) == list(MODELS[f'{name}/sd_keys'])
    assert [str(tuple(v.shape)) for v in m.state_dict().values()] == list(MODELS[f'{name}/sd_shapes'])
    assert sum(p.numel() for p in m.parameters()) == int(MODELS[f'{name}/n_params'])


DWB = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'dwblocks.npz'))


@pytest.mark.parametrize('name', sorted({k.split('/')[0] for k in DWB.files if '/' in k}))
def test_torch_restatement_depthwise_block_matches_reference(name):
    """oracle RefConv2dBlockDW (mobilenet_v1.py:92-125) with the reference's parameters reproduces the
    reference block's output, input gradient, parameter gradients and running statistics bit for bit."""
    C, F_, H, stride, bn = (int(v) for v in DWB[f'{name}/cfg'])
    s = tr.RefSettings(0, (C, H, H), 10, batch_size=4, batch_norm=bool(bn), dropout_rate=0)
    net = tr.RefBaseModel(s)
    blk = tr.RefConv2dBlockDW(net, num_filters=F_, kernel_size=3, stride=stride, padding=1)
    with torch.no_grad():
        for k, p in blk.named_parameters():
            p.copy_(torch.from_numpy(DWB[f'{name}/param/{k}']))
    x = torch.from_numpy(DWB[f'{name}/x']).requires_grad_(True)
    blk.train()
    y = blk(x)
    y.backward(torch.from_numpy(DWB[f'{name}/dy']))
    assert np.array_equal(y.detach().numpy(), DWB[f'{name}/y'])
    assert np.array_equal(x.grad.numpy(), DWB[f'{name}/dx'])
    for k, p in blk.named_parameters():
        assert np.array_equal(p.grad.numpy(), DWB[f'{name}/grad/{k}']), k
    for k, b in blk.named_buffers():
        if 'running' in k:
            assert np.array_equal(b.numpy(), DWB[f'{name}/buf/{k}']), k


def test_affine_restatement_matches_pillow_golden():
    """oracle/transforms_ref.py (torchvision's inverse affine matrix + Pillow's fixed-point nearest-neighbour
    transform, the path of transforms.RandomAffine at src/mngrdata.py:148) against outputs produced by Pillow."""
    from oracle import transforms_ref as tf
    A = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'affine.npz'))
    for i in range(int(A['n_cases'])):
        img, (angle, scale, shear) = A[f'case{i}/img'], A[f'case{i}/draw']
        H, W = img.shape[:2]
        m = tf.tv_inverse_affine_matrix([W * 0.5, H * 0.5], angle, [0.0, 0.0], scale, [shear, 0.0])
        assert np.allclose(m, A[f'case{i}/matrix'], rtol=0, atol=0)
        assert np.array_equal(tf.random_affine(img, angle, scale, shear), A[f'case{i}/out']), i
