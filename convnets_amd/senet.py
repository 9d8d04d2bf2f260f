"""Squeeze-and-excitation families as the reference builds them: SE-ResNet (src/se_resnet.py:17-214: the ResNet
skeleton, every block followed by an SEUnit before the shortcut add) and SENet (src/senet.py:17-214: the
ResNeXt-style bottleneck with a 32-group 3x3 conv, expansion 2, plus the SEUnit).  Same module names (res_block /
res_bottleneck / se_bottleneck with conv.bn.relu-named Conv2dBlocks) => same state_dict keys.  SURVEY.md 8f n4."""
import torch.nn as nn

from . import layers, ops
from .basemodel import MultiClassBaseModel


def _run(block_seq, x):
    for m in block_seq:
        x = m(x)
    return x


class _SEBase(nn.Module):
    def _forward(self, seq, x):
        identity = x
        output = _run(seq, x)
        if self.dim_synch is not None:
            identity = self.dim_synch(identity)
        return ops.AddReLUFn.apply(output, identity)          # output += identity; F.relu(output)


class SEResBlock(_SEBase):                                   # src/se_resnet.py:135-173
    def __init__(self, network, num_filters, expansion, reduction, dim_synch=None, **kwargs):
        super().__init__()
        self.dim_synch = dim_synch
        B = layers.Conv2dBlock
        self.res_block = nn.Sequential(
            B(network, num_filters=num_filters, kernel_size=3, padding=1, **kwargs),
            B(network, num_filters=num_filters * expansion, activation=False, kernel_size=3, padding=1),
            layers.SEUnit(network, reduction))

    def forward(self, x):
        return self._forward(self.res_block, x)


class SEResBottleneck(_SEBase):                              # src/se_resnet.py:175-214
    def __init__(self, network, num_filters, expansion, reduction, dim_synch=None, **kwargs):
        super().__init__()
        self.dim_synch = dim_synch
        B = layers.Conv2dBlock
        self.res_bottleneck = nn.Sequential(
            B(network, num_filters=num_filters, kernel_size=1),
            B(network, num_filters=num_filters, kernel_size=3, padding=1, **kwargs),
            B(network, num_filters=num_filters * expansion, activation=False, kernel_size=1),
            layers.SEUnit(network, reduction))

    def forward(self, x):
        return self._forward(self.res_bottleneck, x)


class SEBottleneck(_SEBase):                                 # src/senet.py:175-213
    def __init__(self, network, num_filters, expansion, reduction, dim_synch=None, **kwargs):
        super().__init__()
        self.dim_synch = dim_synch
        B = layers.Conv2dBlock
        self.se_bottleneck = nn.Sequential(
            B(network, num_filters=num_filters, kernel_size=1),
            B(network, num_filters=num_filters, kernel_size=3, padding=1, **kwargs),
            B(network, num_filters=num_filters * expansion, activation=False, kernel_size=1),
            layers.SEUnit(network, reduction))

    def forward(self, x):
        return self._forward(self.se_bottleneck, x)


class _SEModel(MultiClassBaseModel):
    def __init__(self, setting):
        super().__init__(setting)
        self.features = self.make_feature_layers()
        self.classifier = nn.Sequential(layers.Dropout(p=self.setting.dropout_rate),
                                        layers.Linear(self.num_flat_features(), self.setting.num_classes))
        if self.setting.init_params:
            self.init_params()


class SEResNet(_SEModel):                                    # src/se_resnet.py:17-133
    config = {
        '18': ('basic', [(64, 2, 1), (128, 2, 2), (256, 2, 2), (512, 2, 2)]),
        '34': ('basic', [(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]),
        '26': ('bottleneck', [(64, 2, 1), (128, 2, 2), (256, 2, 2), (512, 2, 2)]),
        '50': ('bottleneck', [(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]),
        '101': ('bottleneck', [(64, 3, 1), (128, 4, 2), (256, 23, 2), (512, 3, 2)]),
        '152': ('bottleneck', [(64, 3, 1), (128, 8, 2), (256, 36, 2), (512, 3, 2)]),
    }

    def make_feature_layers(self):
        kind, config = SEResNet.config[str(self.setting.kind)]
        block_type = SEResBottleneck if kind == 'bottleneck' else SEResBlock
        expansion = 4 if kind == 'bottleneck' else 1
        mods = [self.conv2d_block(num_filters=64, kernel_size=7, stride=2, padding=3),
                self.maxpool2d(kernel_size=3, stride=2, padding=1)]
        for num_filters, num_repeat, stride in config:
            mods.append(self.se_residual_block(block_type, num_filters, expansion, 16, stride))
            for _ in range(1, num_repeat):
                mods.append(self.se_residual_block(block_type, num_filters, expansion, 16))
        mods.append(self.adapt_avgpool2d(output_size=1))
        return nn.Sequential(*mods)

    def se_residual_block(self, block_type, num_filters, expansion, reduction, stride=1):   # :89-114
        if stride != 1 or self.in_channels != num_filters * expansion:
            dim_synch = self.conv2d_block(self.in_channels, num_filters * expansion, set_output=False,
                                          activation=False, kernel_size=1, stride=stride)
        else:
            dim_synch = None
        return block_type(self, num_filters=num_filters, expansion=expansion, reduction=reduction,
                          dim_synch=dim_synch, stride=stride)


class SENet(_SEModel):                                       # src/senet.py:17-133
    config = {
        '26': [(128, 2, 1), (256, 2, 2), (512, 2, 2), (1024, 2, 2)],
        '50': [(128, 3, 1), (256, 4, 2), (512, 6, 2), (1024, 3, 2)],
        '101': [(128, 3, 1), (256, 4, 2), (512, 23, 2), (1024, 3, 2)],
        '152': [(128, 3, 1), (256, 8, 2), (512, 36, 2), (1024, 3, 2)],
    }

    def make_feature_layers(self):
        mods = [self.conv2d_block(num_filters=64, kernel_size=7, stride=2, padding=3),
                self.maxpool2d(kernel_size=3, stride=2, padding=1)]
        for num_filters, num_repeat, stride in SENet.config[str(self.setting.kind)]:
            mods.append(self.se_block(num_filters, 2, 16, stride))
            for _ in range(1, num_repeat):
                mods.append(self.se_block(num_filters, 2, 16))
        mods.append(self.adapt_avgpool2d(output_size=1))
        return nn.Sequential(*mods)

    def se_block(self, num_filters, expansion, reduction, stride=1, cardinality=32):        # :87-113
        if stride != 1 or self.in_channels != num_filters * expansion:
            input_channels = self.in_channels
            dim_synch = self.conv2d_block(self.in_channels, num_filters * expansion, set_output=True,
                                          activation=False, kernel_size=1, stride=stride)
            self.in_channels = input_channels          # (the reference restores the channels, not the spatial size)
        else:
            dim_synch = None
        return SEBottleneck(self, num_filters=num_filters, expansion=expansion, reduction=reduction,
                            dim_synch=dim_synch, stride=stride, groups=cardinality)
