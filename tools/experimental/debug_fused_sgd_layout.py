#!/usr/bin/env python3
"""torch.optim.SGD(fused=True) against foreach on parameters that are NOT contiguous in torch's default order (channels-last Conv3d
weights, as every training convolution of this build holds them): one step from identical states."""
import torch

dev = "cuda:0"
torch.manual_seed(0)
for name, mk in (("contiguous", lambda t: t.contiguous()), ("channels_last_3d", lambda t: t.contiguous(memory_format=torch.channels_last_3d))):
    for shape in ((16, 8, 1, 3, 3), (64, 32, 3, 1, 1), (32, 32, 1, 1, 1)):
        w0, g0 = mk(torch.randn(shape, device=dev)), mk(torch.randn(shape, device=dev))
        out = {}
        for kind, kw in (("single", dict(foreach=False)), ("foreach", dict(foreach=True)), ("fused", dict(fused=True))):
            p = torch.nn.Parameter(w0.clone(memory_format=torch.preserve_format))
            opt = torch.optim.SGD([p], lr=0.1, momentum=0.9, weight_decay=1e-2, **kw)
            for _ in range(2):
                p.grad = g0.clone(memory_format=torch.preserve_format)
                opt.step()
            out[kind] = p.detach().clone()
        print("%-18s %-18s strides %s: |foreach - single| %.2e, |fused - single| %.2e" % (
            name, shape, tuple(w0.stride()), float((out["foreach"] - out["single"]).abs().max()), float((out["fused"] - out["single"]).abs().max())))
