#!/usr/bin/env python3
"""GPU probe (run under rocprofv3 --kernel-trace --stats): one bottleneck tail at the slow pathway's res2 shape — pointwise 64 -> 256,
BatchNorm + shortcut + ReLU, pointwise 256 -> 64 — forward + backward, so that every bn_* kernel in the trace has ONE size
(120 x 8 x 56 x 56 x 256 fp32 = 3.08 GB) and its duration reads as a rate.  python tools/experimental/probe_bn_bwd_rate.py [channels [height = width]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import avtex
from avtex import train_ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
c = int(sys.argv[1]) if len(sys.argv) > 1 else 256
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 56
b, t, h, w = 120, 8, hw, hw
conv_c = torch.nn.Conv3d(c // 4, c, 1, bias=False).to(dev).to(memory_format=torch.channels_last_3d).train()
bn = torch.nn.BatchNorm3d(c).to(dev).train()
conv_a = torch.nn.Conv3d(c, c // 4, 1, bias=False).to(dev).to(memory_format=torch.channels_last_3d).train()
x = torch.randn(b, c // 4, t, h, w, device=dev).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
res = torch.randn(b, c, t, h, w, device=dev).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
for it in range(4):
    with train_ops.bn_replicas(8):
        y = train_ops.bn_act(train_ops.conv3d(x, conv_c, stats=bn), bn, res=res)
        z = train_ops.conv3d(y, conv_a)
    z.backward(torch.ones_like(z))
    torch.cuda.synchronize()
print("calls", {k: v for k, v in train_ops.CALLS.items() if v})
print("tensor GB", b * t * h * w * c * 4 / 1e9)
