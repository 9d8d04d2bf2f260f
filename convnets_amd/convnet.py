"""The small classic nets of BASELINE.json's configs 1-2 on the MI355X layer factory.

ConvNet mirrors the reference's demo model (src/basemodel.py:1070-1108): conv3x3(32) - pool -
conv5x5/2 p1 (64) - pool - 2048 - classes.  LeNet-5 and AlexNet do NOT exist in the reference
(SURVEY F5); they are written in its template_net.py style (src/template_net.py:17-66) on the same
factory, with the canonical shapes BASELINE.json names (LeNet 5x5; AlexNet 11x11/4, 5x5 p2, three
3x3 p1).  Their oracle is oracle/torch_ref.py's plain torch.nn definition of the same nets.
"""
import torch.nn as nn

from . import layers
from .basemodel import MultiClassBaseModel


class ConvNet(MultiClassBaseModel):
    def __init__(self, setting):
        super().__init__(setting)
        self.features = nn.Sequential(
            self.conv2d_block(num_filters=32, kernel_size=3),
            self.maxpool2d(kernel_size=2, stride=2),
            self.conv2d_block(num_filters=64, kernel_size=5, stride=2, padding=1),
            self.maxpool2d(kernel_size=2, stride=2))
        self.classifier = nn.Sequential(
            layers.Linear(self.num_flat_features(), 2048), layers.ReLU(inplace=True),
            layers.Dropout(p=self.setting.dropout_rate), layers.Linear(2048, self.setting.num_classes))
        if self.setting.init_params:
            self.init_params()


class LeNet5(MultiClassBaseModel):
    def __init__(self, setting):
        super().__init__(setting)
        self.features = nn.Sequential(
            self.conv2d_block(num_filters=6, kernel_size=5, padding=2),
            self.maxpool2d(kernel_size=2, stride=2),
            self.conv2d_block(num_filters=16, kernel_size=5),
            self.maxpool2d(kernel_size=2, stride=2))
        self.classifier = nn.Sequential(
            layers.Linear(self.num_flat_features(), 120), layers.ReLU(inplace=True),
            layers.Linear(120, 84), layers.ReLU(inplace=True),
            layers.Dropout(p=self.setting.dropout_rate), layers.Linear(84, self.setting.num_classes))
        if self.setting.init_params:
            self.init_params()


class AlexNet(MultiClassBaseModel):
    def __init__(self, setting):
        super().__init__(setting)
        self.features = nn.Sequential(
            self.conv2d_block(num_filters=64, kernel_size=11, stride=4, padding=2),
            self.maxpool2d(kernel_size=3, stride=2),
            self.conv2d_block(num_filters=192, kernel_size=5, padding=2),
            self.maxpool2d(kernel_size=3, stride=2),
            self.conv2d_block(num_filters=384, kernel_size=3, padding=1),
            self.conv2d_block(num_filters=256, kernel_size=3, padding=1),
            self.conv2d_block(num_filters=256, kernel_size=3, padding=1),
            self.maxpool2d(kernel_size=3, stride=2))
        p = self.setting.dropout_rate
        self.classifier = nn.Sequential(
            layers.Dropout(p=p), layers.Linear(self.num_flat_features(), 4096), layers.ReLU(inplace=True),
            layers.Dropout(p=p), layers.Linear(4096, 4096), layers.ReLU(inplace=True),
            layers.Linear(4096, self.setting.num_classes))
        if self.setting.init_params:
            self.init_params()
