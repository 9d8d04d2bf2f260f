"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the reference's hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
Nothing under convnets_amd/ imports it; the product path fails loudly without the HIP library.

What it restates (all citations are into /root/reference/src):
  * the layer factory + shape tracker        basemodel.py:725-816, 966-1067
  * init_params                              basemodel.py:85-108  (ConvNet variant :1110-1129)
  * ResNet / ResBlock / ResBottleneck        resnet.py:29-36, 53-117, 119-196
  * VGGNet                                   vggnet.py:30-35, 55-103
  * ConvNet                                  basemodel.py:1070-1108
  * the fp32 training-step sequence          basemodel.py:245-288 (zero_grad, forward, CE(sum),
                                             backward, Adam step; clipping is off by default)
The arithmetic itself lives in third-party torch (reference pins torch==1.6.0+cu101,
requirements.txt:18; here torch 2.10 CPU / oneDNN): this restatement calls the same stock
torch.nn layers the reference's factory returns, so on identical weights it is bit-identical to
the reference classes.  Pinning: tests/test_oracle_vs_reference.py (runs only where
/root/reference exists) proves state_dict-key equality and bitwise-equal logits/grads against the
imported reference classes, and tests/golden/*.npz (made by tests/golden/make_golden.py from the
reference classes) pins it on the GPU box where the reference cannot travel.

LeNet-5 and AlexNet do not exist in the reference; they are written here in the
reference's template_net.py style on the same factory, and their oracle is this file itself.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class RefSettings:
    """Attribute bag with the defaults of settings.py:177-222 that the hot path reads."""

    def __init__(self, kind, input_size, num_classes, batch_size=256, batch_norm=True,
                 learning_rate=0.01, weight_decay=1e-4, dropout_rate=0.5, init_params=True,
                 mixed_precision=False, seed=21):
        self.kind = kind
        self.input_size = tuple(input_size)
        self.num_classes = num_classes
        self.batch_size = batch_size
        self.batch_norm = batch_norm
        self.learning_rate = learning_rate
        self.weight_decay = weight_decay
        self.dropout_rate = dropout_rate
        self.init_params = init_params
        self.mixed_precision = mixed_precision
        self.seed = seed


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class RefBaseModel(nn.Module):
    """basemodel.py:28-56 (state), :725-816 (shape tracker), :966-1042 (factories)."""

    def __init__(self, setting):
        super().__init__()
        self.setting = setting
        self.in_channels, self.height, self.width = setting.input_size
        self.cost_function = nn.CrossEntropyLoss(reduction='sum')   # basemodel.py:46
        self.optimizer = None

    # basemodel.py:58-67 (Adam with L2 weight decay; scheduler/scaler are not on the step path)
    def init_optimizer(self):
        self.optimizer = torch.optim.Adam(self.parameters(), lr=self.setting.learning_rate,
                                          weight_decay=self.setting.weight_decay)

    # basemodel.py:85-108
    def init_params(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, mean=0, std=0.01)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    # basemodel.py:725-781 (formula :770-771)
    def save_conv_outshape(self, layer):
        kh, kw = _pair(layer.kernel_size)
        sh, sw = _pair(layer.stride)
        ph, pw = _pair(layer.padding)
        dh, dw = _pair(layer.dilation) if hasattr(layer, 'dilation') else (1, 1)
        self.height = int(np.floor(1 + (self.height + 2 * ph - dh * (kh - 1) - 1) / sh))
        self.width = int(np.floor(1 + (self.width + 2 * pw - dw * (kw - 1) - 1) / sw))
        if isinstance(layer, nn.Conv2d):
            self.in_channels = layer.out_channels
        return self.in_channels, self.height, self.width

    # basemodel.py:783-810
    def save_adapt_outshape(self, layer):
        self.height, self.width = _pair(layer.output_size)
        return self.in_channels, self.height, self.width

    def num_flat_features(self):                                   # basemodel.py:812-816
        return self.in_channels * self.height * self.width

    def conv2d(self, in_channels=None, num_filters=None, set_output=True, **kwargs):   # :966-979
        if in_channels is None:
            in_channels = self.in_channels
        layer = nn.Conv2d(in_channels, num_filters, **kwargs, bias=not self.setting.batch_norm)
        if set_output:
            self.save_conv_outshape(layer)
        return layer

    def conv2d_depthwise(self, in_channels=None, num_filters=None, set_output=True, **kwargs):   # :981-997
        if in_channels is None:
            in_channels = self.in_channels
        if num_filters is None:
            num_filters = self.in_channels
        layer = nn.Conv2d(in_channels, num_filters, groups=in_channels, **kwargs, bias=not self.setting.batch_norm)
        if set_output:
            self.save_conv_outshape(layer)
        return layer

    def conv2d_block(self, in_channels=None, num_filters=None, set_output=True, activation=True,
                     **kwargs):                                                        # :999-1015
        layers = [self.conv2d(in_channels, num_filters, set_output, **kwargs)]
        if self.setting.batch_norm:
            layers += [nn.BatchNorm2d(num_features=num_filters)]
        if activation:
            layers += [nn.ReLU(inplace=True)]
        return nn.Sequential(*layers)

    def maxpool2d(self, set_output=True, **kwargs):                                    # :1017-1024
        layer = nn.MaxPool2d(**kwargs)
        if set_output:
            self.save_conv_outshape(layer)
        return layer

    def avgpool2d(self, set_output=True, **kwargs):                                    # :1026-1033
        layer = nn.AvgPool2d(**kwargs)
        if set_output:
            self.save_conv_outshape(layer)
        return layer

    def adapt_avgpool2d(self, set_output=True, **kwargs):                              # :1035-1042
        layer = nn.AdaptiveAvgPool2d(**kwargs)
        if set_output:
            self.save_adapt_outshape(layer)
        return layer

    def forward(self, x):                                          # resnet.py:110-117 et al.
        x = self.features(x)
        x = torch.flatten(x, start_dim=1)
        return self.classifier(x)


class RefResBlock(nn.Module):                                      # resnet.py:119-156
    def __init__(self, network, num_filters, expansion, dim_synch=None, **kwargs):
        super().__init__()
        self.dim_synch = dim_synch
        self.res_block = nn.Sequential(
            network.conv2d_block(num_filters=num_filters, kernel_size=3, padding=1, **kwargs),
            network.conv2d_block(num_filters=num_filters * expansion, activation=False,
                                 kernel_size=3, padding=1))

    def forward(self, x):
        identity = x
        output = self.res_block(x)
        if self.dim_synch is not None:
            identity = self.dim_synch(identity)
        output += identity
        return F.relu(output)


class RefResBottleneck(nn.Module):                                 # resnet.py:158-196
    def __init__(self, network, num_filters, expansion, dim_synch=None, **kwargs):
        super().__init__()
        self.dim_synch = dim_synch
        self.res_bottleneck = nn.Sequential(
            network.conv2d_block(num_filters=num_filters, kernel_size=1),
            network.conv2d_block(num_filters=num_filters, kernel_size=3, padding=1, **kwargs),
            network.conv2d_block(num_filters=num_filters * expansion, activation=False,
                                 kernel_size=1))

    def forward(self, x):
        identity = x
        output = self.res_bottleneck(x)
        if self.dim_synch is not None:
            identity = self.dim_synch(identity)
        output += identity
        return F.relu(output)


class RefResNet(RefBaseModel):                                     # resnet.py:17-117
    config = {
        '18': ('basic', [(64, 2, 1), (128, 2, 2), (256, 2, 2), (512, 2, 2)]),
        '34': ('basic', [(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]),
        '26': ('bottleneck', [(64, 2, 1), (128, 2, 2), (256, 2, 2), (512, 2, 2)]),
        '50': ('bottleneck', [(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]),
        '101': ('bottleneck', [(64, 3, 1), (128, 4, 2), (256, 23, 2), (512, 3, 2)]),
        '152': ('bottleneck', [(64, 3, 1), (128, 8, 2), (256, 36, 2), (512, 3, 2)]),
    }

    def __init__(self, setting):
        super().__init__(setting)
        kind, cfg = self.config[str(setting.kind)]
        block_type = RefResBottleneck if kind == 'bottleneck' else RefResBlock
        expansion = 4 if kind == 'bottleneck' else 1
        layers = [self.conv2d_block(num_filters=64, kernel_size=7, stride=2, padding=3),
                  self.maxpool2d(kernel_size=3, stride=2, padding=1)]
        for num_filters, num_repeat, stride in cfg:
            layers += [self.residual_block(block_type, num_filters, expansion, stride)]
            for _ in range(1, num_repeat):
                layers += [self.residual_block(block_type, num_filters, expansion)]
        layers += [self.adapt_avgpool2d(output_size=1)]
        self.features = nn.Sequential(*layers)
        self.classifier = nn.Sequential(nn.Dropout(p=setting.dropout_rate),
                                        nn.Linear(self.num_flat_features(), setting.num_classes))
        if setting.init_params:
            self.init_params()

    def residual_block(self, block_type, num_filters, expansion, stride=1):   # resnet.py:85-98
        if stride != 1 or self.in_channels != num_filters * expansion:
            dim_synch = self.conv2d_block(self.in_channels, num_filters * expansion,
                                          set_output=False, activation=False, kernel_size=1,
                                          stride=stride)
        else:
            dim_synch = None
        return block_type(self, num_filters=num_filters, expansion=expansion, dim_synch=dim_synch,
                          stride=stride)


class RefVGGNet(RefBaseModel):                                     # vggnet.py:17-103
    config = {
        '11': [64, 'M', 128, 'M', 256, 256, 'M', 512, 512, 'M', 512, 512, 'M'],
        '13': [64, 64, 'M', 128, 128, 'M', 256, 256, 'M', 512, 512, 'M', 512, 512, 'M'],
        '16': [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M',
               512, 512, 512, 'M'],
        '19': [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M',
               512, 512, 512, 512, 'M'],
    }

    def __init__(self, setting):
        super().__init__(setting)
        layers = []
        for element in self.config[str(setting.kind)]:
            if element == 'M':
                layers += [self.maxpool2d(kernel_size=2, stride=2)]
            else:
                layers += [self.conv2d_block(num_filters=element, kernel_size=3, padding=1)]
        layers += [self.adapt_avgpool2d(output_size=1)]
        self.features = nn.Sequential(*layers)
        p = setting.dropout_rate
        self.classifier = nn.Sequential(
            nn.Dropout(p=p), nn.Linear(self.num_flat_features(), 4096), nn.ReLU(inplace=True),
            nn.Dropout(p=p), nn.Linear(4096, 4096), nn.ReLU(inplace=True),
            nn.Dropout(p=p), nn.Linear(4096, setting.num_classes))
        if setting.init_params:
            self.init_params()


class RefConvNet(RefBaseModel):                                    # basemodel.py:1070-1108
    def __init__(self, setting):
        super().__init__(setting)
        self.features = nn.Sequential(
            self.conv2d_block(num_filters=32, kernel_size=3),
            self.maxpool2d(kernel_size=2, stride=2),
            self.conv2d_block(num_filters=64, kernel_size=5, stride=2, padding=1),
            self.maxpool2d(kernel_size=2, stride=2))
        self.classifier = nn.Sequential(
            nn.Linear(self.num_flat_features(), 2048), nn.ReLU(inplace=True),
            nn.Dropout(p=setting.dropout_rate), nn.Linear(2048, setting.num_classes))
        if setting.init_params:
            self.init_params()


class RefLeNet5(RefBaseModel):
    """LeNet-5 (BASELINE.json config 1) in template_net.py:17-66 style; not in the reference (F5).
    conv5x5(6) pad 2 -> pool2 -> conv5x5(16) -> pool2 -> 400-120-84-classes, ReLU + max-pool as
    the reference's factory offers (its avgpool/tanh-free vocabulary)."""

    def __init__(self, setting):
        super().__init__(setting)
        self.features = nn.Sequential(
            self.conv2d_block(num_filters=6, kernel_size=5, padding=2),
            self.maxpool2d(kernel_size=2, stride=2),
            self.conv2d_block(num_filters=16, kernel_size=5),
            self.maxpool2d(kernel_size=2, stride=2))
        self.classifier = nn.Sequential(
            nn.Linear(self.num_flat_features(), 120), nn.ReLU(inplace=True),
            nn.Linear(120, 84), nn.ReLU(inplace=True),
            nn.Dropout(p=setting.dropout_rate), nn.Linear(84, setting.num_classes))
        if setting.init_params:
            self.init_params()


class RefAlexNet(RefBaseModel):
    """AlexNet (BASELINE.json config 2: 11x11/4, 5x5 p2, three 3x3 p1) in the reference's
    template style with its GAP-free 6x6 flatten; single-tower channel plan 64/192/384/256/256."""

    def __init__(self, setting):
        super().__init__(setting)
        self.features = nn.Sequential(
            self.conv2d_block(num_filters=64, kernel_size=11, stride=4, padding=2),
            self.maxpool2d(kernel_size=3, stride=2),
            self.conv2d_block(num_filters=192, kernel_size=5, padding=2),
            self.maxpool2d(kernel_size=3, stride=2),
            self.conv2d_block(num_filters=384, kernel_size=3, padding=1),
            self.conv2d_block(num_filters=256, kernel_size=3, padding=1),
            self.conv2d_block(num_filters=256, kernel_size=3, padding=1),
            self.maxpool2d(kernel_size=3, stride=2))
        p = setting.dropout_rate
        self.classifier = nn.Sequential(
            nn.Dropout(p=p), nn.Linear(self.num_flat_features(), 4096), nn.ReLU(inplace=True),
            nn.Dropout(p=p), nn.Linear(4096, 4096), nn.ReLU(inplace=True),
            nn.Linear(4096, setting.num_classes))
        if setting.init_params:
            self.init_params()


def ref_training_step(model, inputs, targets):
    """One iteration of the loop body basemodel.py:245-288, fp32 branch (:273-284):
    zero_grad -> forward -> CE(sum) -> score -> backward -> (clipping off) -> optimizer.step.
    Returns (loss, correct) like training_step() :204-215 / score() :111-128."""
    model.train()
    model.optimizer.zero_grad()
    outputs = model(inputs)
    loss = model.cost_function(outputs, targets)
    with torch.no_grad():
        correct = int((outputs.argmax(dim=1) == targets).sum())
    loss.backward()
    model.optimizer.step()
    return float(loss.detach()), correct


def damp_residual_branches(model, factor):
    """Scales gamma of the last BatchNorm of every residual branch (resnet.py:130-135 / :169-175), as
    tests/golden/make_golden.py does on the reference's classes for the well-conditioned fixture `resnet26w`
    (module names are the reference's, so this works on the oracle's and on the product's models alike)."""
    with torch.no_grad():
        for m in model.modules():
            seq = getattr(m, 'res_bottleneck', None)
            if seq is None:
                seq = getattr(m, 'res_block', None)
            if seq is not None:
                bn = [c for c in seq[-1].modules() if isinstance(c, nn.BatchNorm2d)][-1]
                bn.weight.mul_(factor)
    return model


def sample_indices(numel, n):
    """The element positions tests/golden/make_golden.py:sample() stores of a tensor."""
    return torch.linspace(0, numel - 1, min(n, numel)).long()


def synthetic_batch(batch_size, input_size, num_classes, seed=21):
    """x ~ N(0,1) like the reference's own timing tensor (basemodel.py:646); uniform targets."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    x = torch.randn(batch_size, *input_size, generator=g)
    y = torch.randint(0, num_classes, (batch_size,), generator=g)
    return x, y


class RefConv2dBlock(nn.Sequential):                               # basemodel.py:1045-1067
    def __init__(self, network, in_channels=None, num_filters=None, set_output=True, activation=True, **kwargs):
        super().__init__()
        self.add_module('conv', network.conv2d(in_channels, num_filters, set_output, **kwargs))
        if network.setting.batch_norm:
            self.add_module('bn', nn.BatchNorm2d(num_features=num_filters))
        if activation:
            self.add_module('relu', nn.ReLU(inplace=True))


class RefConv2dBlockDW(nn.Sequential):                             # mobilenet_v1.py:92-125
    def __init__(self, network, in_channels=None, num_filters=None, set_output=True, activation=True, **kwargs):
        super().__init__()
        self.add_module('conv_dw', network.conv2d_depthwise(in_channels, in_channels, set_output, **kwargs))
        if network.setting.batch_norm:
            if in_channels is None:
                in_channels = network.in_channels
            self.add_module('bn_dw', nn.BatchNorm2d(num_features=in_channels))
        if activation:
            self.add_module('relu_dw', nn.ReLU(inplace=True))
        self.add_module('conv_pw', network.conv2d(in_channels, num_filters, set_output, kernel_size=1))
        if network.setting.batch_norm:
            self.add_module('bn_pw', nn.BatchNorm2d(num_features=num_filters))
        if activation:
            self.add_module('relu_pw', nn.ReLU(inplace=True))


class RefMobileNetV1(RefBaseModel):                                # mobilenet_v1.py:16-90
    def __init__(self, setting):
        super().__init__(setting)
        cfg = [(64, 1), (128, 2), (128, 1), (256, 2), (256, 1), (512, 2), (512, 1), (512, 1), (512, 1),
               (512, 1), (512, 1), (1024, 2), (1024, 1)]
        mods = [RefConv2dBlock(self, num_filters=32, kernel_size=3, stride=2, padding=1)]
        for filters, stride in cfg:
            mods.append(RefConv2dBlockDW(self, num_filters=filters, kernel_size=3, stride=stride, padding=1))
        mods.append(self.adapt_avgpool2d(output_size=1))
        self.features = nn.Sequential(*mods)
        self.classifier = nn.Sequential(nn.Dropout(p=setting.dropout_rate),
                                        nn.Linear(self.num_flat_features(), setting.num_classes))
        if setting.init_params:
            self.init_params()


class RefResNeXt(RefBaseModel):                                    # resnext.py:16-118 (bottleneck kinds)
    config = {
        '26': [(128, 2, 1), (256, 2, 2), (512, 2, 2), (1024, 2, 2)],
        '50': [(128, 3, 1), (256, 4, 2), (512, 6, 2), (1024, 3, 2)],
        '101': [(128, 3, 1), (256, 4, 2), (512, 23, 2), (1024, 3, 2)],
        '152': [(128, 3, 1), (256, 8, 2), (512, 36, 2), (1024, 3, 2)],
    }

    def __init__(self, setting):
        super().__init__(setting)
        mods = [self.conv2d_block(num_filters=64, kernel_size=7, stride=2, padding=3),
                self.maxpool2d(kernel_size=3, stride=2, padding=1)]
        for num_filters, num_repeat, stride in self.config[str(setting.kind)]:
            mods += [self.residual_block(num_filters, 2, stride)]
            for _ in range(1, num_repeat):
                mods += [self.residual_block(num_filters, 2)]
        self.features = nn.Sequential(*mods)                         # no global pool (resnext.py:51-79)
        self.classifier = nn.Sequential(nn.Dropout(p=setting.dropout_rate),
                                        nn.Linear(self.num_flat_features(), setting.num_classes))
        if setting.init_params:
            self.init_params()

    def residual_block(self, num_filters, expansion, stride=1, cardinality=32):     # resnext.py:81-96
        if stride != 1 or self.in_channels != num_filters * expansion:
            dim_synch = self.conv2d_block(self.in_channels, num_filters * expansion, set_output=False,
                                          activation=False, kernel_size=1, stride=stride)
        else:
            dim_synch = None
        return RefResBottleneck(self, num_filters=num_filters, expansion=expansion, dim_synch=dim_synch,
                                stride=stride, groups=cardinality)


class RefInceptionBlock(nn.Module):                                # inceptionnet_v1.py:99-143
    def __init__(self, network, c1, c3r, c3, c5r, c5, pp):
        super().__init__()
        self.branch1 = network.conv2d_block(network.in_channels, c1, set_output=False, kernel_size=1)
        self.branch2 = nn.Sequential(
            network.conv2d_block(network.in_channels, c3r, set_output=False, kernel_size=1),
            network.conv2d_block(c3r, c3, set_output=False, kernel_size=3, padding=1))
        self.branch3 = nn.Sequential(
            network.conv2d_block(network.in_channels, c5r, set_output=False, kernel_size=1),
            network.conv2d_block(c5r, c5, set_output=False, kernel_size=5, padding=2))
        self.branch4 = nn.Sequential(
            network.maxpool2d(set_output=True, kernel_size=3, stride=1, padding=1),
            network.conv2d_block(num_filters=pp, set_output=True, kernel_size=1))
        network.in_channels = c1 + c3 + c5 + pp

    def forward(self, x):
        return torch.cat([self.branch1(x), self.branch2(x), self.branch3(x), self.branch4(x)], dim=1)


class RefInceptionNetV1(RefBaseModel):                             # inceptionnet_v1.py:16-97
    def __init__(self, setting):
        super().__init__(setting)
        B = RefInceptionBlock
        mods = [self.conv2d_block(num_filters=64, kernel_size=7, stride=2, padding=3),
                self.maxpool2d(kernel_size=3, stride=2, padding=1),
                self.conv2d_block(num_filters=64, kernel_size=1),
                self.conv2d_block(num_filters=192, kernel_size=3, padding=1),
                self.maxpool2d(kernel_size=3, stride=2, padding=1),
                B(self, 64, 96, 128, 16, 32, 32), B(self, 128, 128, 192, 32, 96, 64),
                self.maxpool2d(kernel_size=3, stride=2, padding=1),
                B(self, 192, 96, 208, 16, 48, 64), B(self, 160, 112, 224, 24, 64, 64),
                B(self, 128, 128, 256, 24, 64, 64), B(self, 112, 144, 288, 32, 64, 64),
                B(self, 256, 160, 320, 32, 128, 128),
                self.maxpool2d(kernel_size=3, stride=2, padding=1),
                B(self, 256, 160, 320, 32, 128, 128), B(self, 384, 192, 384, 48, 128, 128),
                self.adapt_avgpool2d(output_size=1)]
        self.features = nn.Sequential(*mods)
        self.classifier = nn.Sequential(nn.Dropout(p=setting.dropout_rate),
                                        nn.Linear(self.num_flat_features(), setting.num_classes))
        if setting.init_params:
            self.init_params()


class RefFire(nn.Module):                                          # squeezenet.py:122-152
    def __init__(self, network, squeeze_num, expand_1x1_num, expand_3x3_num):
        super().__init__()
        self.squeeze = network.conv2d_block(num_filters=squeeze_num, kernel_size=1)
        self.expand_1x1 = network.conv2d_block(network.in_channels, expand_1x1_num, set_output=False, kernel_size=1)
        self.expand_3x3 = network.conv2d_block(network.in_channels, expand_3x3_num, set_output=True, kernel_size=3,
                                               padding=1)
        network.in_channels = expand_1x1_num + expand_3x3_num

    def forward(self, x):
        x = self.squeeze(x)
        return torch.cat([self.expand_1x1(x), self.expand_3x3(x)], dim=1)


class RefSqueezeNet(RefBaseModel):                                 # squeezenet.py:16-120
    config = {
        '1.0': [('conv', 96, 7, 2), ('maxpool', 3, 2), ('fire', 16, 64, 64), ('fire', 16, 64, 64), ('fire', 32, 128, 128),
                ('maxpool', 3, 2), ('fire', 32, 128, 128), ('fire', 48, 192, 192), ('fire', 48, 192, 192),
                ('fire', 64, 256, 256), ('maxpool', 3, 2), ('fire', 64, 256, 256)],
        '1.1': [('conv', 64, 3, 2), ('maxpool', 3, 2), ('fire', 16, 64, 64), ('fire', 16, 64, 64), ('maxpool', 3, 2),
                ('fire', 32, 128, 128), ('fire', 32, 128, 128), ('maxpool', 3, 2), ('fire', 48, 192, 192),
                ('fire', 48, 192, 192), ('fire', 64, 256, 256), ('fire', 64, 256, 256)],
    }

    def __init__(self, setting):
        super().__init__(setting)
        mods = []
        for cfg in self.config[str(setting.kind)]:
            if cfg[0] == 'fire':
                mods.append(RefFire(self, *cfg[1:]))
            elif cfg[0] == 'maxpool':
                mods.append(self.maxpool2d(kernel_size=cfg[1], stride=cfg[2]))
            else:
                mods.append(self.conv2d_block(num_filters=cfg[1], kernel_size=cfg[2], stride=cfg[3]))
        self.features = nn.Sequential(*mods)
        self.classifier = nn.Sequential(nn.Dropout(p=setting.dropout_rate),
                                        self.conv2d_block(num_filters=setting.num_classes, kernel_size=1),
                                        self.adapt_avgpool2d(output_size=1))
        if setting.init_params:
            self.init_params()

    def forward(self, x):
        return torch.flatten(self.classifier(self.features(x)), start_dim=1)


class RefDenseLayer(nn.Module):                                    # densenet.py:105-140
    def __init__(self, network, growth_rate, bottleneck_factor):
        super().__init__()
        p = network.setting.dropout_rate
        self.bottleneck = nn.Sequential(nn.BatchNorm2d(network.in_channels), nn.ReLU(inplace=True), nn.Dropout(p=p),
                                        network.conv2d(num_filters=bottleneck_factor * growth_rate, kernel_size=1))
        self.conv_block = nn.Sequential(nn.BatchNorm2d(network.in_channels), nn.ReLU(inplace=True), nn.Dropout(p=p),
                                        network.conv2d(num_filters=growth_rate, kernel_size=3, padding=1))

    def forward(self, x):
        return torch.cat([x, self.conv_block(self.bottleneck(x))], dim=1)


class RefDenseBlock(nn.Module):                                    # densenet.py:142-169
    def __init__(self, network, dense_block_size, growth_rate, bottleneck_factor):
        super().__init__()
        mods = []
        num_input_filters = network.in_channels
        for i in range(dense_block_size):
            mods.append(RefDenseLayer(network, growth_rate, bottleneck_factor))
            network.in_channels += num_input_filters + i * growth_rate
        self.dense_block = nn.Sequential(*mods)

    def forward(self, x):
        return self.dense_block(x)


class RefTransitionBlock(nn.Sequential):                           # densenet.py:171-188
    def __init__(self, network, compression_factor):
        super().__init__()
        self.add_module('batch_norm', nn.BatchNorm2d(network.in_channels))
        self.add_module('relu', nn.ReLU(inplace=True))
        self.add_module('conv', network.conv2d(num_filters=network.in_channels // compression_factor, kernel_size=1))
        self.add_module('avg_pool', network.avgpool2d(kernel_size=2, stride=2))


class RefDenseNet(RefBaseModel):                                   # densenet.py:16-103
    config = {'121': (32, [6, 12, 24, 16], 64), '169': (32, [6, 12, 32, 32], 64), '201': (32, [6, 12, 48, 32], 64),
              '264': (32, [6, 12, 64, 48], 64), '161': (48, [6, 12, 36, 24], 96)}

    def __init__(self, setting):
        super().__init__(setting)
        growth_rate, blocks, init_features = self.config[str(setting.kind)]
        mods = [self.conv2d_block(num_filters=init_features, kernel_size=7, stride=2, padding=3),
                self.maxpool2d(kernel_size=3, stride=2, padding=1)]
        for i, size in enumerate(blocks):
            mods.append(RefDenseBlock(self, size, growth_rate, bottleneck_factor=4))
            if i != len(blocks) - 1:
                mods.append(RefTransitionBlock(self, compression_factor=2))
        mods += [nn.BatchNorm2d(num_features=self.in_channels), nn.ReLU(inplace=True), self.adapt_avgpool2d(output_size=1)]
        self.features = nn.Sequential(*mods)
        self.classifier = nn.Sequential(nn.Dropout(p=setting.dropout_rate),
                                        nn.Linear(self.num_flat_features(), setting.num_classes))
        if setting.init_params:
            self.init_params()


class RefSEUnit(nn.Module):                                        # senet.py:215-254 == se_resnet.py:216-255
    def __init__(self, network, reduction):
        super().__init__()
        c = network.in_channels
        self.squeeze = network.adapt_avgpool2d(set_output=False, output_size=1)
        self.excitation = nn.Sequential(nn.Linear(c, c // reduction, bias=False), nn.ReLU(inplace=True),
                                        nn.Linear(c // reduction, c, bias=False), nn.Sigmoid())

    def forward(self, x):
        n, c = x.shape[0], x.shape[1]
        gate = self.excitation(self.squeeze(x).view(n, c)).view(n, c, 1, 1)
        return x * gate.expand_as(x)


class _RefSEBlockBase(nn.Module):
    def _fwd(self, seq, x):
        identity = x
        output = seq(x)
        if self.dim_synch is not None:
            identity = self.dim_synch(identity)
        output += identity
        return F.relu(output)


class RefSEResBlock(_RefSEBlockBase):                              # se_resnet.py:135-173
    def __init__(self, network, num_filters, expansion, reduction, dim_synch=None, **kwargs):
        super().__init__()
        self.dim_synch = dim_synch
        self.res_block = nn.Sequential(
            RefConv2dBlock(network, num_filters=num_filters, kernel_size=3, padding=1, **kwargs),
            RefConv2dBlock(network, num_filters=num_filters * expansion, activation=False, kernel_size=3, padding=1),
            RefSEUnit(network, reduction))

    def forward(self, x):
        return self._fwd(self.res_block, x)


class RefSEResBottleneck(_RefSEBlockBase):                         # se_resnet.py:175-214
    def __init__(self, network, num_filters, expansion, reduction, dim_synch=None, **kwargs):
        super().__init__()
        self.dim_synch = dim_synch
        self.res_bottleneck = nn.Sequential(
            RefConv2dBlock(network, num_filters=num_filters, kernel_size=1),
            RefConv2dBlock(network, num_filters=num_filters, kernel_size=3, padding=1, **kwargs),
            RefConv2dBlock(network, num_filters=num_filters * expansion, activation=False, kernel_size=1),
            RefSEUnit(network, reduction))

    def forward(self, x):
        return self._fwd(self.res_bottleneck, x)


class RefSEBottleneck(_RefSEBlockBase):                            # senet.py:175-213
    def __init__(self, network, num_filters, expansion, reduction, dim_synch=None, **kwargs):
        super().__init__()
        self.dim_synch = dim_synch
        self.se_bottleneck = nn.Sequential(
            RefConv2dBlock(network, num_filters=num_filters, kernel_size=1),
            RefConv2dBlock(network, num_filters=num_filters, kernel_size=3, padding=1, **kwargs),
            RefConv2dBlock(network, num_filters=num_filters * expansion, activation=False, kernel_size=1),
            RefSEUnit(network, reduction))

    def forward(self, x):
        return self._fwd(self.se_bottleneck, x)


class RefSEResNet(RefBaseModel):                                   # se_resnet.py:17-133
    config = RefResNet.config

    def __init__(self, setting):
        super().__init__(setting)
        kind, cfg = self.config[str(setting.kind)]
        block_type = RefSEResBottleneck if kind == 'bottleneck' else RefSEResBlock
        expansion = 4 if kind == 'bottleneck' else 1
        mods = [self.conv2d_block(num_filters=64, kernel_size=7, stride=2, padding=3),
                self.maxpool2d(kernel_size=3, stride=2, padding=1)]
        for num_filters, num_repeat, stride in cfg:
            mods.append(self.se_residual_block(block_type, num_filters, expansion, 16, stride))
            for _ in range(1, num_repeat):
                mods.append(self.se_residual_block(block_type, num_filters, expansion, 16))
        mods.append(self.adapt_avgpool2d(output_size=1))
        self.features = nn.Sequential(*mods)
        self.classifier = nn.Sequential(nn.Dropout(p=setting.dropout_rate),
                                        nn.Linear(self.num_flat_features(), setting.num_classes))
        if setting.init_params:
            self.init_params()

    def se_residual_block(self, block_type, num_filters, expansion, reduction, stride=1):
        if stride != 1 or self.in_channels != num_filters * expansion:
            dim_synch = self.conv2d_block(self.in_channels, num_filters * expansion, set_output=False,
                                          activation=False, kernel_size=1, stride=stride)
        else:
            dim_synch = None
        return block_type(self, num_filters=num_filters, expansion=expansion, reduction=reduction,
                          dim_synch=dim_synch, stride=stride)


class RefSENet(RefBaseModel):                                      # senet.py:17-133
    config = RefResNeXt.config

    def __init__(self, setting):
        super().__init__(setting)
        mods = [self.conv2d_block(num_filters=64, kernel_size=7, stride=2, padding=3),
                self.maxpool2d(kernel_size=3, stride=2, padding=1)]
        for num_filters, num_repeat, stride in self.config[str(setting.kind)]:
            mods.append(self.se_block(num_filters, 2, 16, stride))
            for _ in range(1, num_repeat):
                mods.append(self.se_block(num_filters, 2, 16))
        mods.append(self.adapt_avgpool2d(output_size=1))
        self.features = nn.Sequential(*mods)
        self.classifier = nn.Sequential(nn.Dropout(p=setting.dropout_rate),
                                        nn.Linear(self.num_flat_features(), setting.num_classes))
        if setting.init_params:
            self.init_params()

    def se_block(self, num_filters, expansion, reduction, stride=1, cardinality=32):
        if stride != 1 or self.in_channels != num_filters * expansion:
            input_channels = self.in_channels
            dim_synch = self.conv2d_block(self.in_channels, num_filters * expansion, set_output=True,
                                          activation=False, kernel_size=1, stride=stride)
            self.in_channels = input_channels
        else:
            dim_synch = None
        return RefSEBottleneck(self, num_filters=num_filters, expansion=expansion, reduction=reduction,
                               dim_synch=dim_synch, stride=stride, groups=cardinality)


class RefShuffleUnit(nn.Module):                                   # shufflenet_v1.py:99-187
    def __init__(self, network, num_output_channels, groups, stride, downsample, first_conv):
        super().__init__()
        self.groups = groups
        self.downsample = downsample
        bottleneck = num_output_channels // 4
        ident = network.in_channels
        if downsample:
            num_output_channels -= ident
        self.conv1x1_group_compress = RefConv2dBlock(network, num_filters=bottleneck, kernel_size=1,
                                                     groups=1 if first_conv else groups)
        self.conv3x3_depthwise = RefConv2dBlock(network, num_filters=bottleneck, activation=False, kernel_size=3,
                                                stride=stride, padding=1, groups=bottleneck)
        self.conv1x1_group_expand = RefConv2dBlock(network, num_filters=num_output_channels, activation=False,
                                                   kernel_size=1, groups=groups)
        if downsample:
            network.in_channels += ident
        self.avg_pool = network.avgpool2d(set_output=False, kernel_size=3, stride=2, padding=1)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        identity = self.avg_pool(x) if self.downsample else x
        out = self.conv1x1_group_compress(x)
        n, c, h, w = out.shape
        out = out.view(n, self.groups, c // self.groups, h, w).transpose(1, 2).contiguous().view(n, -1, h, w)
        out = self.conv1x1_group_expand(self.conv3x3_depthwise(out))
        if self.downsample:
            out = torch.cat([identity, out], dim=1)
        else:
            out += identity
        return self.relu(out)


class RefShuffleNetV1(RefBaseModel):                               # shufflenet_v1.py:16-97
    config = {
        'g1': [(2, 1, 144), (1, 3, 144), (2, 1, 288), (1, 7, 288), (2, 1, 576), (1, 3, 576)],
        'g2': [(2, 1, 200), (1, 3, 200), (2, 1, 400), (1, 7, 400), (2, 1, 800), (1, 3, 800)],
        'g3': [(2, 1, 240), (1, 3, 240), (2, 1, 480), (1, 7, 480), (2, 1, 960), (1, 3, 960)],
        'g4': [(2, 1, 272), (1, 3, 272), (2, 1, 544), (1, 7, 544), (2, 1, 1088), (1, 3, 1088)],
        'g8': [(2, 1, 384), (1, 3, 384), (2, 1, 768), (1, 7, 768), (2, 1, 1536), (1, 3, 1536)],
    }

    def __init__(self, setting):
        super().__init__(setting)
        groups = int(str(setting.kind)[1:])
        mods = [self.conv2d_block(num_filters=24, kernel_size=3, stride=2, padding=1),
                self.maxpool2d(kernel_size=3, stride=2, padding=1)]
        for i, (stride, repeat, out_ch) in enumerate(self.config[str(setting.kind)]):
            for j in range(repeat):
                mods.append(RefShuffleUnit(self, out_ch, groups, stride, stride == 2, i == 0 and j == 0))
        mods.append(self.adapt_avgpool2d(output_size=1))
        self.features = nn.Sequential(*mods)
        self.classifier = nn.Sequential(nn.Dropout(p=setting.dropout_rate),
                                        nn.Linear(self.num_flat_features(), setting.num_classes))
        if setting.init_params:
            self.init_params()


class RefSKConv(nn.Module):                                        # sknet.py:172-240 == sk_resnet.py:171-239
    def __init__(self, network, num_paths=2, groups=32, reduction=16, min_descriptor=32, **kwargs):
        super().__init__()
        self.num_paths = num_paths
        self.num_features = network.in_channels
        self.kernels = nn.ModuleList([])
        self.attentions = nn.ModuleList([])
        width = max(network.in_channels // reduction, min_descriptor)
        for i in range(num_paths):
            self.kernels.append(RefConv2dBlock(network, num_filters=network.in_channels, set_output=False,
                                               kernel_size=3, padding=1 + i, dilation=1 + i, groups=groups, **kwargs))
            self.attentions.append(nn.Sequential(nn.Conv2d(width, network.in_channels, kernel_size=1)))
        self.global_avgpool = network.adapt_avgpool2d(set_output=False, output_size=1)
        self.descriptor = RefConv2dBlock(network, num_filters=width, set_output=False, kernel_size=1)
        self.softmax = nn.Softmax(dim=1)

    def forward(self, x):
        maps = torch.stack([path(x) for path in self.kernels], dim=1)          # [N, P, C, H, W]
        squeezed = self.descriptor(self.global_avgpool(maps.sum(dim=1)))
        logits = torch.stack([att(squeezed) for att in self.attentions], dim=1)   # [N, P, C, 1, 1]
        return (maps * self.softmax(logits)).sum(dim=1)


class RefSKBlock(nn.Module):                                       # sknet.py:133-170, sk_resnet.py:131-169
    def __init__(self, network, num_filters, expansion, dim_synch=None, seq_name='sk_bottleneck', **kwargs):
        super().__init__()
        self.dim_synch = dim_synch
        self.seq_name = seq_name                                   # 'sk_bottleneck' in sknet.py, 'res_bottleneck' in sk_resnet.py
        setattr(self, seq_name, nn.Sequential(
            RefConv2dBlock(network, num_filters=num_filters, kernel_size=1),
            RefSKConv(network, **kwargs),
            RefConv2dBlock(network, num_filters=num_filters * expansion, activation=False, kernel_size=1)))

    def forward(self, x):
        identity = x
        output = getattr(self, self.seq_name)(x)
        if self.dim_synch is not None:
            identity = self.dim_synch(identity)
        output += identity
        return F.relu(output)


class _RefSKModel(RefBaseModel):
    def __init__(self, setting):
        super().__init__(setting)
        mods = [self.conv2d_block(num_filters=64, kernel_size=7, stride=2, padding=3),
                self.maxpool2d(kernel_size=3, stride=2, padding=1)]
        for num_filters, num_repeat, stride in self.config[str(setting.kind)]:
            mods.append(self.make_block(num_filters, stride))
            for _ in range(1, num_repeat):
                mods.append(self.make_block(num_filters, 1))
        mods.append(self.adapt_avgpool2d(output_size=1))
        self.features = nn.Sequential(*mods)
        self.classifier = nn.Sequential(nn.Dropout(p=setting.dropout_rate),
                                        nn.Linear(self.num_flat_features(), setting.num_classes))
        if setting.init_params:
            self.init_params()


class RefSKNet(_RefSKModel):                                       # sknet.py:17-131
    config = RefResNeXt.config

    def make_block(self, num_filters, stride):                     # sk_block, :86-112 (expansion 2, cardinality 32)
        if stride != 1 or self.in_channels != num_filters * 2:
            keep = self.in_channels
            dim_synch = self.conv2d_block(self.in_channels, num_filters * 2, set_output=True, activation=False,
                                          kernel_size=1, stride=stride)
            self.in_channels = keep
        else:
            dim_synch = None
        return RefSKBlock(self, num_filters=num_filters, expansion=2, dim_synch=dim_synch, stride=stride, groups=32)


class RefSKResNet(_RefSKModel):                                    # sk_resnet.py:17-129
    config = {k: v[1] for k, v in RefResNet.config.items() if v[0] == 'bottleneck'}

    def make_block(self, num_filters, stride):                     # sk_residual_block, :87-110 (expansion 4)
        if stride != 1 or self.in_channels != num_filters * 4:
            dim_synch = self.conv2d_block(self.in_channels, num_filters * 4, set_output=False, activation=False,
                                          kernel_size=1, stride=stride)
        else:
            dim_synch = None
        return RefSKBlock(self, num_filters=num_filters, expansion=4, dim_synch=dim_synch, seq_name='res_bottleneck',
                          stride=stride)


MODELS = {'resnet': RefResNet, 'vgg': RefVGGNet, 'convnet': RefConvNet, 'lenet5': RefLeNet5,
          'alexnet': RefAlexNet, 'mobilenet': RefMobileNetV1, 'resnext': RefResNeXt,
          'inception': RefInceptionNetV1, 'squeezenet': RefSqueezeNet,
          'densenet': RefDenseNet, 'seresnet': RefSEResNet, 'senet': RefSENet,
          'shufflenet': RefShuffleNetV1, 'sknet': RefSKNet, 'skresnet': RefSKResNet}
