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
fc = FusedConv(conv, nn.BatchNorm3d(cout).eval(), True, dev, x3=pd)
M = B * T * H * W
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
