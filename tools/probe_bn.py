"""Achieved HBM rate of the fused train-mode BatchNorm passes (csrc/bn_train.hip) on SlowFast-sized activations:
forward (statistics 4 B/elt read; apply 4 [+4] read + 4 written) and backward (statistics 12 read; apply 12 read + 4 [+4] written)."""
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, ".")
from avtex import train_ops  # noqa: E402

dev = "cuda:0"
for shape, res in (((15, 256, 8, 56, 56), True), ((15, 64, 8, 56, 56), False), ((15, 32, 32, 56, 56), True), ((15, 1024, 8, 14, 14), True),
                   ((1, 256, 8, 56, 56), True)):
    x = torch.randn(shape, device=dev).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    r = torch.randn(shape, device=dev).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True) if res else None
    bn = nn.BatchNorm3d(shape[1]).to(dev).train()
    gy = torch.randn(shape, device=dev).contiguous(memory_format=torch.channels_last_3d)
    n = x.numel()

    def fwd():
        return train_ops.bn_act(x, bn, res=r, relu=True)

    def t(f, reps=10):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(reps):
            f()
        torch.cuda.synchronize()
        return (time.time() - t0) / reps * 1e3

    tf = t(fwd)
    y = fwd()

    def both():
        yy = fwd()
        yy.backward(gy)
        x.grad = None
        if r is not None:
            r.grad = None

    tb = t(both) - tf
    fb = n * 4 * (1 + 2 + (1 if res else 0))
    bb = n * 4 * (3 + 3 + 1 + (1 if res else 0))
    print("%s res=%s: fwd %.3f ms = %.2f TB/s, bwd %.3f ms = %.2f TB/s" % (shape, res, tf, fb / tf / 1e9, tb, bb / tb / 1e9))
