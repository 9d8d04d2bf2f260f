"""VGG 11/13/16/19 in the reference's variant -- mirror of src/vggnet.py:17-103: BN after every
3x3 conv, 'M' = 2x2/2 max-pool, global average pool before a 512-4096-4096-classes head with three
Dropouts (VGG-16 @224/1000 classes = 37,698,472 parameters)."""
import torch.nn as nn

from . import layers
from .basemodel import MultiClassBaseModel


class VGGNet(MultiClassBaseModel):
    config = {
        '11': [64, 'M', 128, 'M', 256, 256, 'M', 512, 512, 'M', 512, 512, 'M'],
        '13': [64, 64, 'M', 128, 128, 'M', 256, 256, 'M', 512, 512, 'M', 512, 512, 'M'],
        '16': [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M'],
        '19': [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M',
               512, 512, 512, 512, 'M'],
    }

    def __init__(self, setting):
        super().__init__(setting)
        self.features = self.make_feature_layers()
        self.classifier = self.make_classifier_layers()
        if self.setting.init_params:
            self.init_params()

    def make_feature_layers(self):                           # src/vggnet.py:55-74
        layers = []
        for element in VGGNet.config[str(self.setting.kind)]:
            if element == 'M':
                layers += [self.maxpool2d(kernel_size=2, stride=2)]
            else:
                layers += [self.conv2d_block(num_filters=element, kernel_size=3, padding=1)]
        layers += [self.adapt_avgpool2d(output_size=1)]
        return nn.Sequential(*layers)

    def make_classifier_layers(self):                        # src/vggnet.py:76-94
        p = self.setting.dropout_rate
        return nn.Sequential(
            layers.Dropout(p=p), layers.Linear(self.num_flat_features(), 4096), layers.ReLU(inplace=True),
            layers.Dropout(p=p), layers.Linear(4096, 4096), layers.ReLU(inplace=True),
            layers.Dropout(p=p), layers.Linear(4096, self.setting.num_classes))
