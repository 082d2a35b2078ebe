#!/usr/bin/env python3
"""One convolution shape, launched repeatedly (for rocprofv3 PMC passes / timing): conv_layer_bench.py cin cout kt kh kw B T H W [res]"""
import sys, time
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avtex
from avtex.fused_slowfast import Act, FusedConv
import torch.nn as nn
cin, cout, kt, kh, kw, B, T, H, W = [int(x) for x in sys.argv[1:10]]
with_res = "res" in sys.argv[10:]
X3 = os.environ.get("PRECISION")  # f16x3 | bf16x3: the contract-grade split-plane kernels
dev = torch.device("cuda:0")
torch.manual_seed(0)
conv = nn.Conv3d(cin, cout, (kt, kh, kw), padding=(kt // 2, kh // 2, kw // 2), bias=False)
from avtex.fused_slowfast import PRECISIONS, new_act, split_planes
pd = PRECISIONS[X3] if X3 else None
M = B * T * H * W
if os.environ.get("IO32"):  # the TRAINING form: fp32 rows in / out through train_ops.conv3d (forward) or its input gradient (IO32=dgrad)
    from avtex import train_ops
    conv = conv.to(dev).train().to(memory_format=torch.channels_last_3d)
    x = torch.randn(B, cin, T, H, W, device=dev).contiguous(memory_format=torch.channels_last_3d)
    bwd = os.environ["IO32"] == "dgrad"
    if bwd:
        x.requires_grad_(True)
        y = train_ops.conv3d(x, conv)
        gy = torch.randn_like(y)
        conv.weight.requires_grad_(False)
        run = lambda: torch.autograd.grad(y, x, gy, retain_graph=True)
    else:
        conv.weight.requires_grad_(False)
        run = lambda: train_ops.conv3d(x, conv)
    with torch.set_grad_enabled(bwd):
        for _ in range(3): run()
        torch.cuda.synchronize(); t0 = time.time()
        n = 20
        for _ in range(n): run()
        torch.cuda.synchronize(); t = (time.time() - t0) / n
    fl = 2.0 * M * cin * kt * kh * kw * cout
    print("IO32 %s cin%d cout%d k(%d,%d,%d) M=%d: %.1f us, %.1f TF/s, %.0f GB/s" % (os.environ["IO32"], cin, cout, kt, kh, kw, M, t * 1e6, fl / t / 1e12, 4.0 * M * (cin + cout) / t / 1e9))
if not os.environ.get("IO32"):
    fc = FusedConv(conv, nn.BatchNorm3d(cout).eval(), True, dev, x3=pd)
    def mk(c):
        v = torch.randn(M, c, device=dev)
        if pd is None:
            return Act(v.to(torch.bfloat16), (B, T, H, W))
        hi, lo = split_planes(v, pd)
        return Act(hi, (B, T, H, W), lo=lo)
    x = mk(cin)
    res = mk(cout) if with_res else None
    out = new_act(M, cout, (B, T, H, W), dev, pd is not None)
    for _ in range(3): fc(x, out=out, res=res)
    torch.cuda.synchronize(); t0 = time.time()
    n = 20
    for _ in range(n): fc(x, out=out, res=res)
    torch.cuda.synchronize(); t = (time.time() - t0) / n
    fl = 2.0 * M * cin * kt * kh * kw * cout
    by = 2.0 * (M * cin + M * cout * (2 if with_res else 1)) * (2 if pd is not None else 1)
    print("cin%d cout%d k(%d,%d,%d) M=%d res=%s: %.1f us, %.1f TF/s, %.0f GB/s" % (cin, cout, kt, kh, kw, M, with_res, t * 1e6, fl / t / 1e12, by / t / 1e9))
