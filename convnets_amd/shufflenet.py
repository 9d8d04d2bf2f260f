"""ShuffleNet-v1 (g1/g2/g3/g4/g8) as the reference builds it (src/shufflenet_v1.py:16-187): 3x3/2 stem with 24
channels, shuffle units = grouped 1x1 compress -> channel shuffle -> depthwise 3x3 -> grouped 1x1 expand, shortcut
by add (stride 1) or by concat with a 3x3/2 average pool (stride 2).  Same module names => same state_dict keys.
SURVEY.md 8f n4.  The grouped 1x1 convs (Cin != Cout) run as the dense conv of their block-diagonal weight."""
import torch
import torch.nn as nn

from . import layers, ops
from .basemodel import MultiClassBaseModel


class ShuffleUnit(nn.Module):                                # src/shufflenet_v1.py:99-187
    def __init__(self, network, num_output_channels, groups, stride, downsample, first_conv):
        super().__init__()
        self.groups = groups
        self.groups_first_conv1x1 = 1 if first_conv else groups
        self.downsample = downsample
        bottleneck_channels = num_output_channels // 4
        num_channels_identity = network.in_channels
        if self.downsample:
            num_output_channels -= num_channels_identity
        B = layers.Conv2dBlock
        self.conv1x1_group_compress = B(network, num_filters=bottleneck_channels, kernel_size=1,
                                        groups=self.groups_first_conv1x1)
        self.conv3x3_depthwise = B(network, num_filters=bottleneck_channels, activation=False, kernel_size=3,
                                   stride=stride, padding=1, groups=bottleneck_channels)
        self.conv1x1_group_expand = B(network, num_filters=num_output_channels, activation=False, kernel_size=1,
                                      groups=self.groups)
        if self.downsample:
            network.in_channels += num_channels_identity
        self.avg_pool = network.avgpool2d(set_output=False, kernel_size=3, stride=2, padding=1)
        self.relu = layers.ReLU(inplace=True)

    def forward(self, x):
        if self.downsample:
            identity = self.avg_pool(x)
            output = self.conv1x1_group_compress(x)
            output = ops.ChannelShuffleFn.apply(output, self.groups)
            output = self.conv1x1_group_expand(self.conv3x3_depthwise(output))
            return self.relu(torch.cat([identity, output], dim=1))
        # stride-1 unit: `output + x` and the ReLU run inside the last BN's apply pass, and the two gradients of x
        # (compress conv, shortcut) are summed inside the compress conv's dgrad epilogue (ops.GradLink) -- unless
        # that conv takes the fp32 detour (channel counts off the 8-grid), which does not carry a link
        first = self.conv1x1_group_compress.conv
        link = ops.GradLink(2) if (self.training and torch.is_grad_enabled() and x.requires_grad
                                   and not first._f32_detour and hasattr(self.conv1x1_group_compress, 'bn')) else None
        output = self.conv1x1_group_compress(x, in_link=link)
        output = ops.ChannelShuffleFn.apply(output, self.groups)
        output = self.conv3x3_depthwise(output)
        return self.conv1x1_group_expand(output, residual=x, relu_after_add=True, res_link=link)


class ShuffleNetV1(MultiClassBaseModel):
    config = {
        'g1': [(2, 1, 144), (1, 3, 144), (2, 1, 288), (1, 7, 288), (2, 1, 576), (1, 3, 576)],
        'g2': [(2, 1, 200), (1, 3, 200), (2, 1, 400), (1, 7, 400), (2, 1, 800), (1, 3, 800)],
        'g3': [(2, 1, 240), (1, 3, 240), (2, 1, 480), (1, 7, 480), (2, 1, 960), (1, 3, 960)],
        'g4': [(2, 1, 272), (1, 3, 272), (2, 1, 544), (1, 7, 544), (2, 1, 1088), (1, 3, 1088)],
        'g8': [(2, 1, 384), (1, 3, 384), (2, 1, 768), (1, 7, 768), (2, 1, 1536), (1, 3, 1536)],
    }

    def __init__(self, setting):
        super().__init__(setting)
        self.features = self.make_feature_layers()
        self.classifier = nn.Sequential(layers.Dropout(p=self.setting.dropout_rate),
                                        layers.Linear(self.num_flat_features(), self.setting.num_classes))
        if self.setting.init_params:
            self.init_params()

    def make_feature_layers(self):                           # src/shufflenet_v1.py:52-78
        config = ShuffleNetV1.config[str(self.setting.kind)]
        groups = int(str(self.setting.kind)[1:])
        mods = [self.conv2d_block(num_filters=24, kernel_size=3, stride=2, padding=1),
                self.maxpool2d(kernel_size=3, stride=2, padding=1)]
        for i, (stride, repeat, num_output_channels) in enumerate(config):
            for j in range(repeat):
                mods.append(ShuffleUnit(self, num_output_channels, groups, stride, stride == 2, i == 0 and j == 0))
        mods.append(self.adapt_avgpool2d(output_size=1))
        return nn.Sequential(*mods)
