"""SqueezeNet 1.0 / 1.1 as the reference builds it (src/squeezenet.py:16-152): conv stem, Fire modules
(1x1 squeeze -> parallel 1x1 / 3x3 expand -> channel concat), max-pools, and a convolutional classifier
(Dropout -> 1x1 conv block to num_classes -> global average pool).  Same module names => same
state_dict keys.  SURVEY.md 8f n4.  Fire widths (16/32/48/64 squeeze) are off the 64-channel grid and
run on the EDGE instantiation of the implicit-GEMM kernels."""
import torch
import torch.nn as nn

from . import layers, ops
from .basemodel import MultiClassBaseModel


class Fire(nn.Module):                                       # src/squeezenet.py:122-152
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


class SqueezeNet(MultiClassBaseModel):
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
        self.features = self.make_feature_layers()
        self.classifier = self.make_classifier_layers()
        if self.setting.init_params:
            self.init_params()

    def make_feature_layers(self):                           # src/squeezenet.py:73-100
        mods = []
        for cfg in SqueezeNet.config[str(self.setting.kind)]:
            if cfg[0] == 'fire':
                mods.append(Fire(self, *cfg[1:]))
            elif cfg[0] == 'maxpool':
                mods.append(self.maxpool2d(kernel_size=cfg[1], stride=cfg[2]))
            else:
                mods.append(self.conv2d_block(num_filters=cfg[1], kernel_size=cfg[2], stride=cfg[3]))
        return nn.Sequential(*mods)

    def make_classifier_layers(self):                        # src/squeezenet.py:102-111
        return nn.Sequential(layers.Dropout(p=self.setting.dropout_rate),
                             self.conv2d_block(num_filters=self.setting.num_classes, kernel_size=1),
                             self.adapt_avgpool2d(output_size=1))

    def forward(self, x):                                    # src/squeezenet.py:113-120
        if x.dim() != 4 or tuple(x.shape[1:]) != tuple(self.setting.input_size):
            raise ValueError('expected input [N,%d,%d,%d], got %s' % (*self.setting.input_size, tuple(x.shape)))
        x = ops.nchw_to_nhwc(x, self.setting.compute_dtype)
        x = self.classifier(self.features(x))
        return torch.flatten(x, start_dim=1).float()
