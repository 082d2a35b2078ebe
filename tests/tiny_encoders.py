"""Tiny plugin encoders shared by tools/gen_golden.py (which runs them inside the
imported reference) and the parity tests (which run them inside this build).

They follow the reference's encoder plugin contract
(contrastive_video_textures/models/models.py:253-263, 392-399):
  - non-SlowFast: forward([B,C,T,H,W]) -> [B,C',t,h,w]  (wrapped with AdaptiveAvgPool3d(1))
  - SlowFast:     forward([slow [B,3,8,H,W], fast [B,3,32,H,W]]) -> [B,D]
Weights come from torch.manual_seed(seed) + default init, so fixtures only carry
the seed and a checksum.
"""
import torch
import torch.nn as nn


class TinySlowFast(nn.Module):
    """Two-pathway toy: pooled pathways -> concat(slow, fast) -> linear, D = dim."""

    def __init__(self, dim=48):
        super().__init__()
        self.pool_s = nn.AdaptiveAvgPool3d((2, 4, 4))
        self.pool_f = nn.AdaptiveAvgPool3d((4, 4, 4))
        self.fc = nn.Linear(3 * 2 * 16 + 3 * 4 * 16, dim)

    def forward(self, x):
        slow, fast = x
        b = slow.shape[0]
        z = torch.cat((self.pool_s(slow).reshape(b, -1), self.pool_f(fast).reshape(b, -1)), dim=1)
        return self.fc(torch.tanh(z))


class TinyR3D(nn.Module):
    """Non-SlowFast toy: one strided conv3d -> [B,C',t,h,w]."""

    def __init__(self, dim=32):
        super().__init__()
        self.conv = nn.Conv3d(3, dim, kernel_size=(3, 3, 3), stride=(2, 2, 2), padding=1)

    def forward(self, x):
        return torch.tanh(self.conv(x))


def seeded(cls, seed, **kw):
    torch.manual_seed(seed)
    m = cls(**kw)
    return m.eval()


def checksum(module):
    return float(sum(p.double().abs().sum() for p in module.state_dict().values()))
