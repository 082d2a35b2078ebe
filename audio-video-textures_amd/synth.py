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


def structured_video(seed, n_frames, h, w, scene_len=24, device="cpu", chunk=2048, variety=0):
    """uint8 [n_frames, h, w, 3] RGB (on `device`; generated in chunks so a 16k-frame video never exists in fp32).
    variety = 1: every scene additionally gets its own coarse colour LAYOUT (a random 4 x 4 grid of colours, bilinearly
    upsampled) and a wider range of texture frequency / contrast, so that unrelated scenes differ in many global statistics at
    once (the bench's inputs: transition rows with a sharp survivor set at the reference's default threshold 0.3)."""
    g = torch.Generator().manual_seed(seed)
    n_key = n_frames // scene_len + 2
    yy, xx = torch.meshgrid(torch.linspace(0, 1, h), torch.linspace(0, 1, w), indexing="ij")
    keys = []
    for _ in range(n_key):
        fx, fy = (torch.rand(2, generator=g) * 6 + 0.5).tolist()
        ph = (torch.rand(3, generator=g) * 2 * math.pi).tolist()
        colour = torch.rand(3, generator=g)
        contrast = 0.15 + 0.35 * torch.rand(1, generator=g).item()
        if variety:
            grid = torch.rand(1, 3, 4, 4, generator=g)
            layout = torch.nn.functional.interpolate(grid, size=(h, w), mode="bilinear", align_corners=True)[0]
            scale = float(torch.rand(1, generator=g)) ** 2 * 4 + 0.25  # frequency multiplier 0.25 .. 4.25, skewed low
            fx, fy, contrast = fx * scale, fy * scale, 0.05 + 0.45 * float(torch.rand(1, generator=g))
            colour = [0.6 * colour[c] + 0.8 * (layout[c] - 0.5) + 0.2 for c in range(3)]  # scene colour cast + its own layout
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


def randomise_bn(model, seed, sparsity=1.0, branch_scale=1.0):
    """In place: BN scales in [0.5, 1.0] (incl. the zero-initialised last BN of each block), small random running
    statistics, and biases shifted down by `sparsity` standard deviations of the scale range so that post-ReLU features
    are sparse and depend on the input.  branch_scale < 1 shrinks the scale of every block's LAST BatchNorm (`c_bn`; PySlowFast
    zero-initialises it and trained networks keep it small): the residual branches perturb the identity path instead of
    replacing it, and the embedding keeps more of the input's variety through the depth."""
    g = torch.Generator().manual_seed(seed)
    last = {id(m.c_bn) for m in model.modules() if hasattr(m, "c_bn")}
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, (nn.BatchNorm3d, nn.BatchNorm2d)):
                n = m.weight.numel()
                m.weight.copy_((0.5 + 0.5 * torch.rand(n, generator=g)) * (branch_scale if id(m) in last else 1.0))
                m.bias.copy_(0.2 * torch.randn(n, generator=g) - 0.25 * sparsity)
                m.running_mean.copy_(0.1 * torch.randn(n, generator=g))
                m.running_var.copy_(0.8 + 0.4 * torch.rand(n, generator=g))
    return model


def perturbed_copy(model, seed, rel=0.05):
    """A deep copy of `model` whose convolution weights are moved by `rel` of their own RMS (seeded noise) and whose BatchNorm
    scales / biases by `rel` of their spread: two encoders that share an initialisation and have diverged a little — what the
    reference's q_encoder / t_encoder are (both start from the same Kinetics checkpoint and are trained apart, main.py:329-334),
    so that sim[i, j] = <q_enc(seg_i), t_enc(seg_j)> is high for the same / neighbouring segments and low for unrelated ones."""
    import copy

    g = torch.Generator().manual_seed(seed)
    out = copy.deepcopy(model)
    with torch.no_grad():
        for m in out.modules():
            if isinstance(m, (nn.Conv3d, nn.Conv2d)):
                w = m.weight
                m.weight.add_(torch.randn(w.shape, generator=g).to(w.device) * (rel * float(w.float().pow(2).mean().sqrt())))
            elif isinstance(m, (nn.BatchNorm3d, nn.BatchNorm2d)):
                n = m.weight.numel()
                m.weight.add_(torch.randn(n, generator=g).to(m.weight.device) * (rel * 0.15))
                m.bias.add_(torch.randn(n, generator=g).to(m.bias.device) * (rel * 0.2))
    return out


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
