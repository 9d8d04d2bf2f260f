"""ResNet 18/34/26/50/101/152 on the MI355X layer factory -- mirror of src/resnet.py:17-196.

Config table (:29-36), v1.5 stride placement (the stride goes to the 3x3, :170-172), projection
shortcut = 1x1 conv + BN without ReLU (:90-91), GAP -> Dropout -> Linear head (:100-108).
Module names match the reference so state_dict keys interchange (features.2.dim_synch.0.weight,
features.2.res_bottleneck.1.1.running_var, classifier.1.weight, ...).  The block tail
`output += identity; output = F.relu(output)` (:191-194) is fused into the last BN's apply pass.
"""
import torch
import torch.nn as nn

from . import layers
from .basemodel import MultiClassBaseModel


def _input_link(block, x):
    """The block input x feeds the main branch and the shortcut; in training their two gradients
    are summed inside the last-running conv dgrad (ops.GradLink) instead of by autograd."""
    from . import ops
    first = block.res_block[0] if hasattr(block, 'res_block') else block.res_bottleneck[0]
    fused = block.training and torch.is_grad_enabled() and x.requires_grad and len(first) > 1 \
        and isinstance(first[1], nn.BatchNorm2d)
    return ops.GradLink(2) if fused else None


class ResBlock(nn.Module):                                   # src/resnet.py:119-156
    def __init__(self, network, num_filters, expansion, dim_synch=None, **kwargs):
        super().__init__()
        self.dim_synch = dim_synch
        self.res_block = nn.Sequential(
            network.conv2d_block(num_filters=num_filters, kernel_size=3, padding=1, **kwargs),
            network.conv2d_block(num_filters=num_filters * expansion, activation=False,
                                 kernel_size=3, padding=1))

    def forward(self, x):
        link = _input_link(self, x)
        if self.dim_synch is None:
            output = self.res_block[0](x, in_link=link)
            return self.res_block[1](output, residual=x, relu_after_add=True, res_link=link)
        # (its BN is applied inside the last block's BN pass -- when that block runs one)
        identity = self.dim_synch(x, in_link=link, defer_apply=self.res_block[1].applies_deferred_residual())
        output = self.res_block[0](x, in_link=link)
        return self.res_block[1](output, residual=identity, relu_after_add=True)


class ResBottleneck(nn.Module):                              # src/resnet.py:158-196
    def __init__(self, network, num_filters, expansion, dim_synch=None, **kwargs):
        super().__init__()
        self.dim_synch = dim_synch
        self.res_bottleneck = nn.Sequential(
            network.conv2d_block(num_filters=num_filters, kernel_size=1),
            network.conv2d_block(num_filters=num_filters, kernel_size=3, padding=1, **kwargs),
            network.conv2d_block(num_filters=num_filters * expansion, activation=False,
                                 kernel_size=1))

    def forward(self, x):
        link = _input_link(self, x)
        if self.dim_synch is None:
            output = self.res_bottleneck[0](x, in_link=link)
            output = self.res_bottleneck[1](output)
            return self.res_bottleneck[2](output, residual=x, relu_after_add=True, res_link=link)
        # (its BN is applied inside the last block's BN pass -- when that block runs one)
        identity = self.dim_synch(x, in_link=link, defer_apply=self.res_bottleneck[2].applies_deferred_residual())
        output = self.res_bottleneck[0](x, in_link=link)
        output = self.res_bottleneck[1](output)
        return self.res_bottleneck[2](output, residual=identity, relu_after_add=True)


class ResNet(MultiClassBaseModel):
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
        self.features = self.make_feature_layers()
        self.classifier = self.make_classifier_layers()
        if self.setting.init_params:
            self.init_params()

    def make_feature_layers(self):                           # src/resnet.py:53-83
        block_type_str, config = ResNet.config[str(self.setting.kind)]
        block_type = ResBottleneck if block_type_str == 'bottleneck' else ResBlock
        expansion = 4 if block_type_str == 'bottleneck' else 1
        layers = [self.conv2d_block(num_filters=64, kernel_size=7, stride=2, padding=3),
                  self.maxpool2d(kernel_size=3, stride=2, padding=1)]
        for num_filters, num_repeat, stride in config:
            layers += [self.residual_block(block_type, num_filters, expansion, stride)]
            for _ in range(1, num_repeat):
                layers += [self.residual_block(block_type, num_filters, expansion)]
        layers += [self.adapt_avgpool2d(output_size=1)]
        return nn.Sequential(*layers)

    def residual_block(self, block_type, num_filters, expansion, stride=1):   # src/resnet.py:85-98
        if stride != 1 or self.in_channels != num_filters * expansion:
            dim_synch = self.conv2d_block(self.in_channels, num_filters * expansion, set_output=False,
                                          activation=False, kernel_size=1, stride=stride)
        else:
            dim_synch = None
        return block_type(self, num_filters=num_filters, expansion=expansion, dim_synch=dim_synch,
                          stride=stride)

    def make_classifier_layers(self):                        # src/resnet.py:100-108
        return nn.Sequential(layers.Dropout(p=self.setting.dropout_rate),
                             layers.Linear(self.num_flat_features(), self.setting.num_classes))
