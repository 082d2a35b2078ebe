#!/usr/bin/env python3
"""The training kernels at config-5 shapes (15 target clips) for rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE in separate runs):
one forward + backward of four representative SlowFast layers through avtex.train_ops (conv_x3 IO32 forward / dgrad, wgrad_x3,
bn_train).  Algorithmic bytes per launch are printed so the counters can be read against them."""
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avtex import train_ops  # noqa: E402

dev = "cuda:0"
CL = torch.channels_last_3d
for cin, cout, k, p, shape in ((64, 256, (1, 1, 1), (0, 0, 0), (15, 64, 8, 56, 56)),       # slow res2 c
                               (256, 256, (1, 3, 3), (0, 1, 1), (15, 256, 8, 14, 14)),      # slow res4 b
                               (1024, 256, (3, 1, 1), (1, 0, 0), (15, 1024, 8, 14, 14)),    # slow res4 a
                               (32, 32, (1, 3, 3), (0, 1, 1), (15, 32, 32, 14, 14))):       # fast res4 b
    conv = nn.Conv3d(cin, cout, k, padding=p, bias=False).to(dev).to(memory_format=CL).train()
    bn = nn.BatchNorm3d(cout).to(dev).train()
    x = torch.randn(shape, device=dev).contiguous(memory_format=CL).requires_grad_(True)
    for _ in range(2):
        y = train_ops.bn_act(train_ops.conv3d(x, conv), bn, relu=True)
        y.backward(torch.ones_like(y))
        x.grad = None
        conv.zero_grad(set_to_none=True)
        bn.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    m, taps = y.numel() // cout, k[0] * k[1] * k[2]
    xb, yb, wb = x.numel() * 4, y.numel() * 4, conv.weight.numel() * 4
    print("cin%d cout%d k%s M=%d: algorithmic MB  conv fwd %.1f  dgrad %.1f  wgrad %.1f | bn fwd stats %.1f apply %.1f  bwd stats %.1f apply %.1f" % (
        cin, cout, k, m, (xb + yb + wb) / 1e6, (xb + yb + wb) / 1e6, (xb + yb + wb) / 1e6, yb / 1e6, 2 * yb / 1e6, 3 * yb / 1e6, 4 * yb / 1e6))
