"""Synthetic inputs for benchmarks and precision tests (no datasets or checkpoints travel to the GPU box).

`structured_video` is a sequence of visually distinct "scenes" (colour, contrast and spatial frequency drawn per key
frame) cross-faded over time, so that neighbouring clip windows are similar and distant ones are not: transition rows
have a non-trivial survivor set, unlike iid noise, whose windows all embed to the same direction.
`randomise_bn` gives a random-init SlowFast non-degenerate residual branches (PySlowFast zero-initialises the last BN
of every block) and sparse, input-dependent final features, so cosine similarities spread over a wide range.
"""
import math

import torch
import torch.nn as nn


def structured_video(seed, n_frames, h, w, scene_len=24, device="cpu", chunk=2048):
    """uint8 [n_frames, h, w, 3] RGB (on `device`; generated in chunks so a 16k-frame video never exists in fp32)."""
    g = torch.Generator().manual_seed(seed)
    n_key = n_frames // scene_len + 2
    yy, xx = torch.meshgrid(torch.linspace(0, 1, h), torch.linspace(0, 1, w), indexing="ij")
    keys = []
    for _ in range(n_key):
        fx, fy = (torch.rand(2, generator=g) * 6 + 0.5).tolist()
        ph = (torch.rand(3, generator=g) * 2 * math.pi).tolist()
        colour = torch.rand(3, generator=g)
        contrast = 0.15 + 0.35 * torch.rand(1, generator=g).item()
        chans = [colour[c] + contrast * torch.sin(2 * math.pi * (fx * xx + fy * yy) + ph[c]) *
                 torch.cos(2 * math.pi * (fy * xx - fx * yy) * 0.5 + ph[(c + 1) % 3]) for c in range(3)]
        keys.append(torch.stack(chans, -1))
    keys = torch.stack(keys).to(device)  # [n_key, h, w, 3]
    out = torch.empty((n_frames, h, w, 3), dtype=torch.uint8, device=device)
    for f0 in range(0, n_frames, chunk):
        f1 = min(f0 + chunk, n_frames)
        t = torch.arange(f0, f1, dtype=torch.float32, device=device) / scene_len
        i0 = t.floor().long()
        fr = (t - i0.float()).view(-1, 1, 1, 1)
        fr = fr * fr * (3 - 2 * fr)  # smooth cross-fade
        vid = (1 - fr) * keys[i0] + fr * keys[i0 + 1]
        noise = torch.randn(vid.shape, generator=g) if str(device) == "cpu" else \
            torch.randn(vid.shape, device=device, generator=torch.Generator(device=device).manual_seed(seed * 1000003 + f0))
        out[f0:f1] = ((vid + 0.03 * noise).clamp(0, 1) * 255).round().to(torch.uint8)
    return out


def randomise_bn(model, seed, sparsity=1.0):
    """In place: BN scales in [0.5, 1.0] (incl. the zero-initialised last BN of each block), small random running
    statistics, and biases shifted down by `sparsity` standard deviations of the scale range so that post-ReLU features
    are sparse and depend on the input."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, (nn.BatchNorm3d, nn.BatchNorm2d)):
                n = m.weight.numel()
                m.weight.copy_(0.5 + 0.5 * torch.rand(n, generator=g))
                m.bias.copy_(0.2 * torch.randn(n, generator=g) - 0.25 * sparsity)
                m.running_mean.copy_(0.1 * torch.randn(n, generator=g))
                m.running_var.copy_(0.8 + 0.4 * torch.rand(n, generator=g))
    return model


def calibrate_bn(model, slow, fast):
    """One train-mode forward on the given clips with momentum 1: every BatchNorm's running statistics become the
    statistics of its actual input, as in a trained network.  Without this a random-init SlowFast in eval mode maps every
    clip to the same direction (cosines equal to three decimals) and every transition row is degenerate."""
    moms = {}
    for m in model.modules():
        if isinstance(m, nn.BatchNorm3d):
            moms[m] = m.momentum
            m.momentum = 1.0
    was_training = model.training
    model.train()
    with torch.no_grad():
        model([slow, fast])
    model.train(was_training)
    for m, mom in moms.items():
        m.momentum = mom
    return model
