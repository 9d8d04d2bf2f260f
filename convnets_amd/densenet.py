"""DenseNet-121/169/201/264/161 as the reference builds it (src/densenet.py:16-131): pre-activation dense
layers (BN -> ReLU -> Dropout -> 1x1 conv -> BN -> ReLU -> Dropout -> 3x3 conv, output concatenated to the
input), transition blocks (BN -> ReLU -> 1x1 conv -> 2x2 average pool).  Same containers and child order =>
same state_dict keys (features.2.dense_block.0.bottleneck.0.weight ...).  SURVEY.md 8f n4.
Each BN + ReLU pair is one fused kernel pass; the concat is torch.cat on NHWC tensors."""
import torch
import torch.nn as nn

from . import layers
from .basemodel import MultiClassBaseModel


def _bn_relu_drop_conv(seq, x):
    """children: 0 BatchNorm2d, 1 ReLU, 2 Dropout, 3 Conv2d"""
    x = seq[0](x, relu=True) if isinstance(seq[0], layers.BatchNorm2d) else seq[1](seq[0](x))
    return seq[3](seq[2](x))


class DenseLayer(nn.Module):                                 # src/densenet.py:105-140
    def __init__(self, network, growth_rate, bottleneck_factor):
        super().__init__()
        p = network.setting.dropout_rate
        self.bottleneck = nn.Sequential(
            layers.BatchNorm2d(network.in_channels), layers.ReLU(inplace=True), layers.Dropout(p=p),
            network.conv2d(num_filters=bottleneck_factor * growth_rate, kernel_size=1))
        self.conv_block = nn.Sequential(
            layers.BatchNorm2d(network.in_channels), layers.ReLU(inplace=True), layers.Dropout(p=p),
            network.conv2d(num_filters=growth_rate, kernel_size=3, padding=1))

    def forward(self, x):
        out = _bn_relu_drop_conv(self.conv_block, _bn_relu_drop_conv(self.bottleneck, x))
        return torch.cat([x, out], dim=1)


class DenseBlock(nn.Module):                                 # src/densenet.py:142-169
    def __init__(self, network, dense_block_size, growth_rate, bottleneck_factor):
        super().__init__()
        mods = []
        num_input_filters = network.in_channels
        for i in range(dense_block_size):
            mods.append(DenseLayer(network, growth_rate, bottleneck_factor))
            network.in_channels += num_input_filters + i * growth_rate
        self.dense_block = nn.Sequential(*mods)

    def forward(self, x):
        return self.dense_block(x)


class TransitionBlock(nn.Sequential):                        # src/densenet.py:171-188
    def __init__(self, network, compression_factor):
        super().__init__()
        self.add_module('batch_norm', layers.BatchNorm2d(network.in_channels))
        self.add_module('relu', layers.ReLU(inplace=True))
        self.add_module('conv', network.conv2d(num_filters=network.in_channels // compression_factor, kernel_size=1))
        self.add_module('avg_pool', network.avgpool2d(kernel_size=2, stride=2))

    def forward(self, x):
        return self.avg_pool(self.conv(self.batch_norm(x, relu=True)))


class DenseNet(MultiClassBaseModel):
    config = {'121': (32, [6, 12, 24, 16], 64), '169': (32, [6, 12, 32, 32], 64), '201': (32, [6, 12, 48, 32], 64),
              '264': (32, [6, 12, 64, 48], 64), '161': (48, [6, 12, 36, 24], 96)}

    def __init__(self, setting):
        super().__init__(setting)
        self.features = self.make_feature_layers()
        self.classifier = self.make_classifier_layers()
        if self.setting.init_params:
            self.init_params()

    def make_feature_layers(self):                           # src/densenet.py:56-84
        growth_rate, dense_blocks_list, num_init_features = DenseNet.config[str(self.setting.kind)]
        mods = [self.conv2d_block(num_filters=num_init_features, kernel_size=7, stride=2, padding=3),
                self.maxpool2d(kernel_size=3, stride=2, padding=1)]
        for i, dense_block_size in enumerate(dense_blocks_list):
            mods.append(DenseBlock(self, dense_block_size, growth_rate, bottleneck_factor=4))
            if i != len(dense_blocks_list) - 1:
                mods.append(TransitionBlock(self, compression_factor=2))
        mods += [layers.BatchNorm2d(num_features=self.in_channels), layers.ReLU(inplace=True),
                 self.adapt_avgpool2d(output_size=1)]
        return _DenseFeatures(*mods)

    def make_classifier_layers(self):                        # src/densenet.py:86-94
        return nn.Sequential(layers.Dropout(p=self.setting.dropout_rate),
                             layers.Linear(self.num_flat_features(), self.setting.num_classes))


class _DenseFeatures(nn.Sequential):
    """nn.Sequential whose forward fuses a BatchNorm2d child with the ReLU child that follows it."""

    def forward(self, x):
        mods = list(self)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, layers.BatchNorm2d) and i + 1 < len(mods) and isinstance(mods[i + 1], layers.ReLU):
                x = m(x, relu=True)
                i += 2
            else:
                x = m(x)
                i += 1
        return x
