"""Selective-kernel families as the reference builds them: SKNet (src/sknet.py:17-170: ResNeXt-style widths,
expansion 2, SKConv with cardinality 32 in place of the grouped 3x3) and SK-ResNet (src/sk_resnet.py:17-169: the
ResNet bottleneck widths, expansion 4, SKConv with its default 32 groups).  Module names sk_bottleneck (SKNet) / res_bottleneck (SK-ResNet) /
kernels / attentions / descriptor as in the reference => same state_dict keys.  SURVEY.md 8f n4."""
import torch.nn as nn

from . import layers, ops
from .basemodel import MultiClassBaseModel


class _SKBlock(nn.Module):
    seq_name = None

    def __init__(self, network, num_filters, expansion, dim_synch=None, **kwargs):
        super().__init__()
        self.dim_synch = dim_synch
        B = layers.Conv2dBlock
        setattr(self, self.seq_name, nn.Sequential(
            B(network, num_filters=num_filters, kernel_size=1),
            layers.SKConv(network, **kwargs),
            B(network, num_filters=num_filters * expansion, activation=False, kernel_size=1)))

    def forward(self, x):
        seq = getattr(self, self.seq_name)
        identity = x if self.dim_synch is None else self.dim_synch(x)
        output = seq[1](seq[0](x))
        # output += identity; F.relu(output): fused into the last BN's apply pass
        return seq[2](output, residual=identity, relu_after_add=True)


class SKBottleneck(_SKBlock):                                # src/sknet.py:133-170
    seq_name = 'sk_bottleneck'


class SKResBottleneck(_SKBlock):                             # src/sk_resnet.py:131-169
    seq_name = 'res_bottleneck'


class _SKModel(MultiClassBaseModel):
    def __init__(self, setting):
        super().__init__(setting)
        self.features = self.make_feature_layers()
        self.classifier = nn.Sequential(layers.Dropout(p=self.setting.dropout_rate),
                                        layers.Linear(self.num_flat_features(), self.setting.num_classes))
        if self.setting.init_params:
            self.init_params()

    def make_feature_layers(self):
        mods = [self.conv2d_block(num_filters=64, kernel_size=7, stride=2, padding=3),
                self.maxpool2d(kernel_size=3, stride=2, padding=1)]
        for num_filters, num_repeat, stride in self.config[str(self.setting.kind)]:
            mods.append(self.make_block(num_filters, stride))
            for _ in range(1, num_repeat):
                mods.append(self.make_block(num_filters))
        mods.append(self.adapt_avgpool2d(output_size=1))
        return nn.Sequential(*mods)


class SKNet(_SKModel):                                       # src/sknet.py:17-131
    config = {
        '26': [(128, 2, 1), (256, 2, 2), (512, 2, 2), (1024, 2, 2)],
        '50': [(128, 3, 1), (256, 4, 2), (512, 6, 2), (1024, 3, 2)],
        '101': [(128, 3, 1), (256, 4, 2), (512, 23, 2), (1024, 3, 2)],
        '152': [(128, 3, 1), (256, 8, 2), (512, 36, 2), (1024, 3, 2)],
    }

    def make_block(self, num_filters, stride=1):
        return self.sk_block(SKBottleneck, num_filters, 2, stride)

    def sk_block(self, block_type, num_filters, expansion, stride=1, cardinality=32):       # :86-112
        if stride != 1 or self.in_channels != num_filters * expansion:
            input_channels = self.in_channels
            dim_synch = self.conv2d_block(self.in_channels, num_filters * expansion, set_output=True,
                                          activation=False, kernel_size=1, stride=stride)
            self.in_channels = input_channels          # (the reference restores the channels, not the spatial size)
        else:
            dim_synch = None
        return block_type(self, num_filters=num_filters, expansion=expansion, dim_synch=dim_synch, stride=stride,
                          groups=cardinality)


class SKResNet(_SKModel):                                    # src/sk_resnet.py:17-129
    config = {
        '26': [(64, 2, 1), (128, 2, 2), (256, 2, 2), (512, 2, 2)],
        '50': [(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)],
        '101': [(64, 3, 1), (128, 4, 2), (256, 23, 2), (512, 3, 2)],
        '152': [(64, 3, 1), (128, 8, 2), (256, 36, 2), (512, 3, 2)],
    }

    def make_block(self, num_filters, stride=1):
        return self.sk_residual_block(SKResBottleneck, num_filters, 4, stride)

    def sk_residual_block(self, block_type, num_filters, expansion, stride=1):              # :87-110
        if stride != 1 or self.in_channels != num_filters * expansion:
            dim_synch = self.conv2d_block(self.in_channels, num_filters * expansion, set_output=False,
                                          activation=False, kernel_size=1, stride=stride)
        else:
            dim_synch = None
        return block_type(self, num_filters=num_filters, expansion=expansion, dim_synch=dim_synch, stride=stride)
