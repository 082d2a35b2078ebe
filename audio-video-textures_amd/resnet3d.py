"""3D-ResNet encoder plugins for the non-SlowFast path: forward([B,3,T,H,W]) -> [B,C',t,h,w].

Same layer layout and state-dict keys as the reference's Kensho-Hara style network
(contrastive_video_textures/models/video_models/resnet3d.py:119-191, factories :223-318), including
its final fixed-size AvgPool3d (the operator pools again with AdaptiveAvgPool3d(1), models.py:253-260)
and [quirk] that "resnet50" is built from BasicBlocks (resnet3d.py:265-304).  ResNeXt/DenseNet
factories of the reference cannot be reached (kwarg mismatch) and are not built.
"""
import math

import torch.nn as nn


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv3d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm3d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv3d(planes, planes, 3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm3d(planes)
        self.downsample = downsample

    def forward(self, x):
        r = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + r)


class ResNet3d(nn.Module):
    def __init__(self, layers, sample_size, sample_duration):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv3d(3, 64, 7, stride=(1, 2, 2), padding=(3, 3, 3), bias=False)
        self.bn1 = nn.BatchNorm3d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool3d(3, stride=2, padding=1)
        self.layer1 = self._make(64, layers[0], 1)
        self.layer2 = self._make(128, layers[1], 2)
        self.layer3 = self._make(256, layers[2], 2)
        self.layer4 = self._make(512, layers[3], 2)
        self.avgpool = nn.AvgPool3d((int(math.ceil(sample_duration / 16)), int(math.ceil(sample_size / 32)),
                                     int(math.ceil(sample_size / 32))), stride=1)
        self.fc_dim = 512
        self.fc = nn.Linear(512, 1039)  # present in the reference's checkpoints, never applied
        for m in self.modules():
            if isinstance(m, nn.Conv3d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out")

    def _make(self, planes, blocks, stride):
        down = None
        if stride != 1 or self.inplanes != planes:
            down = nn.Sequential(nn.Conv3d(self.inplanes, planes, 1, stride=stride, bias=False), nn.BatchNorm3d(planes))
        layers = [BasicBlock(self.inplanes, planes, stride, down)]
        self.inplanes = planes
        layers += [BasicBlock(planes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.avgpool(x)


LAYERS = {"resnet10": [1, 1, 1, 1], "resnet18": [2, 2, 2, 2], "resnet34": [3, 4, 6, 3], "resnet50": [3, 4, 6, 3]}


def build(arch, sample_size, sample_duration):
    return ResNet3d(LAYERS[arch], sample_size, sample_duration)
