"""ResNeXt as the reference builds it (src/resnext.py:16-118): the ResNet skeleton with double-width
stages, a 32-group 3x3 conv in every bottleneck (cardinality 32, :82), expansion 2, and -- unlike the
reference's ResNet -- NO global average pool: the classifier's Linear sees C*H*W features (:98-107).
Same module names as the reference => same state_dict keys.  SURVEY.md 8f n4.

The grouped 3x3 conv runs on the implicit-GEMM kernels as a block-diagonal conv per 64-channel chunk
(include/cnk.h, cnk_conv_desc.groups).  Only the bottleneck kinds (26/50/101/152) are built here: the
reference's basic-block variant hands `stride` to BOTH 3x3 convs of a block (src/resnext.py:126-129), so its
stride-2 stages cannot add the shortcut -- that configuration does not run in the reference either."""
import torch.nn as nn

from . import layers
from .basemodel import MultiClassBaseModel
from .resnet import ResBottleneck


class ResNeXt(MultiClassBaseModel):
    config = {
        '26': ('bottleneck', [(128, 2, 1), (256, 2, 2), (512, 2, 2), (1024, 2, 2)]),
        '50': ('bottleneck', [(128, 3, 1), (256, 4, 2), (512, 6, 2), (1024, 3, 2)]),
        '101': ('bottleneck', [(128, 3, 1), (256, 4, 2), (512, 23, 2), (1024, 3, 2)]),
        '152': ('bottleneck', [(128, 3, 1), (256, 8, 2), (512, 36, 2), (1024, 3, 2)]),
    }

    def __init__(self, setting):
        super().__init__(setting)
        if str(setting.kind) not in ResNeXt.config:
            raise NotImplementedError('convnets_amd.ResNeXt: bottleneck kinds %s only (see the module docstring)'
                                      % sorted(ResNeXt.config))
        self.features = self.make_feature_layers()
        self.classifier = self.make_classifier_layers()
        if self.setting.init_params:
            self.init_params()

    def make_feature_layers(self):                           # src/resnext.py:51-79
        _, config = ResNeXt.config[str(self.setting.kind)]
        mods = [self.conv2d_block(num_filters=64, kernel_size=7, stride=2, padding=3),
                self.maxpool2d(kernel_size=3, stride=2, padding=1)]
        for num_filters, num_repeat, stride in config:
            mods += [self.residual_block(num_filters, 2, stride)]
            for _ in range(1, num_repeat):
                mods += [self.residual_block(num_filters, 2)]
        return nn.Sequential(*mods)

    def residual_block(self, num_filters, expansion, stride=1, cardinality=32):   # src/resnext.py:81-96
        if stride != 1 or self.in_channels != num_filters * expansion:
            dim_synch = self.conv2d_block(self.in_channels, num_filters * expansion, set_output=False,
                                          activation=False, kernel_size=1, stride=stride)
        else:
            dim_synch = None
        return ResBottleneck(self, num_filters=num_filters, expansion=expansion, dim_synch=dim_synch,
                             stride=stride, groups=cardinality)

    def make_classifier_layers(self):                        # src/resnext.py:98-107
        return nn.Sequential(layers.Dropout(p=self.setting.dropout_rate),
                             layers.Linear(self.num_flat_features(), self.setting.num_classes))
