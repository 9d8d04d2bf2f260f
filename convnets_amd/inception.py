"""InceptionNet-v1 (GoogLeNet with BN, no auxiliary heads) as the reference builds it
(src/inceptionnet_v1.py:16-143): same layer list and module names (features.5.branch2.1.0.weight ...),
hence the same state_dict keys.  Every branch is conv2d_block / maxpool2d from the factory, the four
branch outputs are concatenated along channels (torch.cat on NHWC tensors -- plumbing, :142).
SURVEY.md 8f n4.  Most branch widths (16, 24, 48, 96, 112, 144, 160, 208 ...) are off the 64-channel
grid of the implicit-GEMM kernels, so those convs run on the small-channel / direct kernels."""
import torch
import torch.nn as nn

from . import layers
from .basemodel import MultiClassBaseModel


class InceptionBlock(nn.Module):                             # src/inceptionnet_v1.py:99-143
    def __init__(self, network, ch_conv1x1, ch_conv3x3_red, ch_conv3x3, ch_conv5x5_red, ch_conv5x5, pool_proj):
        super().__init__()
        self.branch1 = network.conv2d_block(network.in_channels, ch_conv1x1, set_output=False, kernel_size=1)
        self.branch2 = nn.Sequential(
            network.conv2d_block(network.in_channels, ch_conv3x3_red, set_output=False, kernel_size=1),
            network.conv2d_block(ch_conv3x3_red, ch_conv3x3, set_output=False, kernel_size=3, padding=1))
        self.branch3 = nn.Sequential(
            network.conv2d_block(network.in_channels, ch_conv5x5_red, set_output=False, kernel_size=1),
            network.conv2d_block(ch_conv5x5_red, ch_conv5x5, set_output=False, kernel_size=5, padding=2))
        self.branch4 = nn.Sequential(
            network.maxpool2d(set_output=True, kernel_size=3, stride=1, padding=1),
            network.conv2d_block(num_filters=pool_proj, set_output=True, kernel_size=1))
        network.in_channels = ch_conv1x1 + ch_conv3x3 + ch_conv5x5 + pool_proj

    def forward(self, x):
        # x feeds three 1x1 convs (and the pooling branch): the convs' data gradients are summed inside the last
        # one's dgrad epilogue (ops.GradLink); autograd only adds the pooling branch's gradient to that total
        from . import ops
        link = ops.GradLink(3) if (self.training and torch.is_grad_enabled() and x.requires_grad) else None
        b2 = self.branch2[1](self.branch2[0](x, in_link=link))
        b3 = self.branch3[1](self.branch3[0](x, in_link=link))
        return torch.cat([self.branch1(x, in_link=link), b2, b3, self.branch4(x)], dim=1)


class InceptionNetV1(MultiClassBaseModel):
    def __init__(self, setting):
        super().__init__(setting)
        self.features = self.make_feature_layers()
        self.classifier = self.make_classifier_layers()
        if self.setting.init_params:
            self.init_params()

    def make_feature_layers(self):                           # src/inceptionnet_v1.py:43-77
        B = InceptionBlock
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
        return nn.Sequential(*mods)

    def make_classifier_layers(self):                        # src/inceptionnet_v1.py:79-89
        return nn.Sequential(layers.Dropout(p=self.setting.dropout_rate),
                             layers.Linear(self.num_flat_features(), self.setting.num_classes))
