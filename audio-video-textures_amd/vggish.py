"""VGGish audio encoder plugin: forward([n,1,100,64]) -> [n,12288].

Mirrors the reference's module (contrastive_video_textures/models/audio_models/vggish.py:13-46):
six conv3x3+ReLU with four 2x2 max-pools, output permuted to NHWC and flattened; the three `fc`
layers exist (so `pytorch_vggish.pth` loads, main.py:337-338) but are never applied (vggish.py:45).
"""
import torch.nn as nn

from .train_ops import conv2d


class VGGish(nn.Module):
    def __init__(self):
        super().__init__()
        cfg = [(1, 64, True), (64, 128, True), (128, 256, False), (256, 256, True), (256, 512, False),
               (512, 512, True)]
        layers = []
        for cin, cout, pool in cfg:
            layers += [nn.Conv2d(cin, cout, 3, stride=1, padding=1), nn.ReLU(inplace=True)]
            if pool:
                layers.append(nn.MaxPool2d(2, stride=2))
        self.features = nn.Sequential(*layers)
        self.fc = nn.Sequential(nn.Linear(512 * 24, 4096), nn.ReLU(inplace=True), nn.Linear(4096, 4096),
                                nn.ReLU(inplace=True), nn.Linear(4096, 128), nn.ReLU(inplace=True))

    def forward(self, x):
        if self.training and x.is_cuda:
            # the m = 2 training branch (models.py:343-345, 405-407): the six convolutions forward and backward on the hand-written
            # split-plane kernels (train_ops.conv2d), ReLU / max-pool by torch — no MIOpen convolution
            for layer in self.features:
                x = conv2d(x, layer) if isinstance(layer, nn.Conv2d) else layer(x)
        else:
            x = self.features(x)
        x = x.permute(0, 2, 3, 1).contiguous()
        return x.view(x.size(0), -1)
