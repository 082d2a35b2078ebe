"""Deterministic weights for the SuperSloMo UNets (no checkpoint travels with the repo): every tensor of the reference's
state-dict layout (models/slowmo.py:165-177) drawn from its own seeded generator, He-uniform so activations keep their
scale through the LeakyReLU stack and the flows come out at pixel scale (the back-warps then actually move pixels)."""
import zlib

import torch

_LAYERS = [("conv1", None, 32, 7), ("conv2", 32, 32, 7)]
_w = (32, 64, 128, 256, 512, 512)
for _i, _k in enumerate((5, 3, 3, 3, 3)):
    _LAYERS += [("down%d.conv1" % (_i + 1), _w[_i], _w[_i + 1], _k), ("down%d.conv2" % (_i + 1), _w[_i + 1], _w[_i + 1], _k)]
for _i, (_a, _b) in enumerate(((512, 512), (512, 256), (256, 128), (128, 64), (64, 32))):
    _LAYERS += [("up%d.conv1" % (_i + 1), _a, _b, 3), ("up%d.conv2" % (_i + 1), 2 * _b, _b, 3)]
_LAYERS += [("conv3", 32, None, 3)]


def unet_state(cin, cout, seed, head_gain=1.0):
    sd = {}
    for name, a, b, k in _LAYERS:
        a = cin if a is None else a
        b = cout if b is None else b
        g = torch.Generator().manual_seed(seed * 1000003 + zlib.crc32(name.encode()))
        bound = (6.0 / (a * k * k) / (1 + 0.01)) ** 0.5
        gain = head_gain if name == "conv3" else 1.0
        sd[name + ".weight"] = (torch.rand((b, a, k, k), generator=g) * 2 - 1) * bound * gain
        sd[name + ".bias"] = (torch.rand((b,), generator=g) * 2 - 1) * 0.05 * gain
    return sd


def frame_pair(seed, h, w):
    """Two smooth uint8 frames, the second a shifted / dimmed copy of the first plus a moving blob."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing="ij")
    ph = torch.rand(6, generator=g) * 6.28

    def img(dx, dy, gain):
        r = 0.5 + 0.35 * torch.sin((xx + dx) / 9.0 + ph[0]) * torch.cos((yy + dy) / 13.0 + ph[1])
        gch = 0.5 + 0.35 * torch.sin((xx + dx) / 17.0 + ph[2]) * torch.sin((yy + dy) / 7.0 + ph[3])
        b = 0.5 + 0.35 * torch.cos((xx + dx + yy + dy) / 11.0 + ph[4])
        blob = torch.exp(-(((xx - w / 2 - 2 * dx) ** 2 + (yy - h / 2 - 2 * dy) ** 2) / (2 * (h / 8) ** 2)))
        f = torch.stack([r, gch, b], dim=2) * gain + 0.3 * blob.unsqueeze(2)
        return (f.clamp(0, 1) * 255).round().to(torch.uint8)

    return img(0.0, 0.0, 1.0), img(3.0, -2.0, 0.95)
