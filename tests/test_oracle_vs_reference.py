"""CPU suite, build container only (skipped where /root/reference is absent, e.g. the GPU box):
proves the oracle's torch.nn restatement is the reference -- identical state_dict keys and
BITWISE-identical logits and gradients against the imported reference classes on the same
seed-drawn weights (both reach the same ATen/oneDNN kernels)."""
import os
import sys

import pytest
import torch

REF = '/root/reference/src'
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree not present')


@pytest.fixture(scope='module')
def ref_modules():
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'golden'))
    import make_golden
    return make_golden.import_reference()


@pytest.mark.parametrize('family,kind,in_size,bn', [('resnet', 26, (3, 32, 32), True), ('resnet', 18, (3, 32, 32), True),
                                                    ('vgg', 11, (3, 32, 32), True), ('convnet', 0, (1, 28, 28), False)])
def test_restatement_is_bitwise_the_reference(ref_modules, family, kind, in_size, bn):
    settings, basemodel, resnet, vggnet = ref_modules
    from oracle import torch_ref as tr
    cls = {'resnet': resnet.ResNet, 'vgg': vggnet.VGGNet, 'convnet': basemodel.ConvNet}[family]
    s = settings.Settings(kind=kind, input_size=in_size, num_classes=10, batch_size=8, batch_norm=bn,
                          mixed_precision=False, dropout_rate=0, seed=21)
    torch.manual_seed(21)
    ref = cls(s)
    torch.manual_seed(21)
    mine = tr.MODELS[family](tr.RefSettings(kind, in_size, 10, batch_size=8, batch_norm=bn, dropout_rate=0))
    assert list(ref.state_dict().keys()) == list(mine.state_dict().keys())
    for a, b in zip(ref.state_dict().values(), mine.state_dict().values()):
        assert torch.equal(a, b)
    x, y = tr.synthetic_batch(8, in_size, 10, seed=21)
    ref.train()
    mine.train()
    o1, o2 = ref(x), mine(x)
    assert torch.equal(o1, o2)
    ref.cost_function(o1, y).backward()
    mine.cost_function(o2, y).backward()
    for p, q in zip(ref.parameters(), mine.parameters()):
        assert torch.equal(p.grad, q.grad)


def test_product_factory_mirrors_reference_factory_signatures(ref_modules):
    import inspect
    settings, basemodel, resnet, vggnet = ref_modules
    import convnets_amd as ca
    for name in ('conv2d', 'conv2d_block', 'maxpool2d', 'adapt_avgpool2d', 'save_conv_outshape',
                 'save_adapt_outshape', 'num_flat_features', 'init_params', 'init_optimizer', 'score',
                 'gradient_clipping', 'forward_step', 'training_step', 'validation_step', 'train_model',
                 'valid_model'):
        a = inspect.signature(getattr(basemodel.MultiClassBaseModel, name))
        b = inspect.signature(getattr(ca.MultiClassBaseModel, name))
        assert list(a.parameters) == list(b.parameters), name
    assert ca.ResNet.config == resnet.ResNet.config and ca.VGGNet.config == vggnet.VGGNet.config
