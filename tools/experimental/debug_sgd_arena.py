#!/usr/bin/env python3
"""Which of (fused SGD, single-pass gradient arena) moves the parameters after three steps?  Pairwise max |diff| of four runs + a
single-tensor SGD run (foreach=False) as the reference."""
import copy
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from avtex import train_ops  # noqa: E402

DEV = "cuda:0"


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.c1 = nn.Conv3d(8, 16, (1, 3, 3), padding=(0, 1, 1), bias=False)
        self.b1 = nn.BatchNorm3d(16)
        self.c2 = nn.Conv3d(16, 8, (3, 1, 1), padding=(1, 0, 0), bias=False)

    def forward(self, x):
        return train_ops.conv3d(train_ops.bn_act(train_ops.conv3d(x, self.c1), self.b1), self.c2)


torch.manual_seed(1)
net = Net().to(DEV).to(memory_format=torch.channels_last_3d).train()
x = torch.randn(2, 8, 4, 12, 12, device=DEV).contiguous(memory_format=torch.channels_last_3d)


def run(fused, arena, foreach=None, steps=3, invalidate=False):
    m = copy.deepcopy(net)
    kw = dict(fused=True) if fused else dict(foreach=foreach)
    opt = torch.optim.SGD(m.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-2, **kw)
    acc = train_ops.MicroBatchGradients(m.parameters(), single_pass_arena=arena)
    grads = []
    for step in range(steps):
        acc.begin(1)
        (m(x * (1.0 + step)).square().mean()).backward()
        acc.finish()
        grads.append([p.grad.detach().clone() for p in m.parameters()])
        v0 = m.c1.weight._version
        opt.step()
        if step == 0:
            print("   (fused %s: Conv3d weight _version %d -> %d over optimizer.step())" % (fused, v0, m.c1.weight._version))
        if invalidate:
            train_ops.invalidate_weight_cache()
    torch.cuda.synchronize()
    return [p.detach().clone() for p in m.parameters()], grads


runs = {"single-tensor": run(False, False, foreach=False), "foreach": run(False, False, foreach=True), "foreach+arena": run(False, True, foreach=True),
        "fused": run(True, False), "fused+arena": run(True, True), "fused + invalidate": run(True, False, invalidate=True),
        "fused+arena + invalidate": run(True, True, invalidate=True)}
ref = runs["single-tensor"]
for k, (ps, gs) in runs.items():
    dp = max(float((a - b).abs().max()) for a, b in zip(ps, ref[0]))
    dg = [max(float((a - b).abs().max()) for a, b in zip(g, rg)) for g, rg in zip(gs, ref[1])]
    print("%-15s max |dparam| vs single-tensor %.3e; max |dgrad| per step %s" % (k, dp, ["%.2e" % v for v in dg]))
print("param scale %.3f, grad scale %s" % (max(float(p.abs().max()) for p in ref[0]), ["%.2e" % max(float(g.abs().max()) for g in gs) for gs in ref[1]]))
