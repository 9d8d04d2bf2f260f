"""MobileNet-v1 as the reference builds it (src/mobilenet_v1.py:16-90): 3x3/2 conv block, thirteen
depthwise-separable blocks, GAP -> Dropout -> Linear head; same module names, hence the same
state_dict keys (features.1.conv_dw.weight, features.1.bn_pw.running_mean, classifier.1.weight ...).
SURVEY.md 8f n4: the model family the depthwise kernels unlock."""
import torch.nn as nn

from . import layers
from .basemodel import MultiClassBaseModel


class MobileNetV1(MultiClassBaseModel):
    def __init__(self, setting):
        super().__init__(setting)
        self.features = self.make_feature_layers()
        self.classifier = self.make_classifier_layers()
        if self.setting.init_params:
            self.init_params()

    def make_feature_layers(self):                           # src/mobilenet_v1.py:42-71
        B, DW = layers.Conv2dBlock, layers.Conv2dBlockDW
        cfg = [(64, 1), (128, 2), (128, 1), (256, 2), (256, 1), (512, 2), (512, 1), (512, 1), (512, 1),
               (512, 1), (512, 1), (1024, 2), (1024, 1)]
        mods = [B(self, num_filters=32, kernel_size=3, stride=2, padding=1)]
        for filters, stride in cfg:
            kw = dict(kernel_size=3, padding=1)
            if stride != 1:
                kw['stride'] = stride
            mods.append(DW(self, num_filters=filters, **kw))
        mods.append(self.adapt_avgpool2d(output_size=1))
        return nn.Sequential(*mods)

    def make_classifier_layers(self):                        # src/mobilenet_v1.py:73-82
        return nn.Sequential(layers.Dropout(p=self.setting.dropout_rate),
                             layers.Linear(self.num_flat_features(), self.setting.num_classes))
