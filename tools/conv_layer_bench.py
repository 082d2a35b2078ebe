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
with_res = len(sys.argv) > 10
dev = torch.device("cuda:0")
torch.manual_seed(0)
conv = nn.Conv3d(cin, cout, (kt, kh, kw), padding=(kt // 2, kh // 2, kw // 2), bias=False)
fc = FusedConv(conv, nn.BatchNorm3d(cout).eval(), True, dev)
M = B * T * H * W
x = Act(torch.randn(M, cin, device=dev).to(torch.bfloat16), (B, T, H, W))
res = Act(torch.randn(M, cout, device=dev).to(torch.bfloat16), (B, T, H, W)) if with_res else None
out = Act(torch.empty(M, cout, dtype=torch.bfloat16, device=dev), (B, T, H, W))
for _ in range(3): fc(x, out=out, res=res)
torch.cuda.synchronize(); t0 = time.time()
n = 20
for _ in range(n): fc(x, out=out, res=res)
torch.cuda.synchronize(); t = (time.time() - t0) / n
fl = 2.0 * M * cin * kt * kh * kw * cout
by = 2.0 * (M * cin + M * cout * (2 if with_res else 1))
print("cin%d cout%d k(%d,%d,%d) M=%d res=%s: %.1f us, %.1f TF/s, %.0f GB/s" % (cin, cout, kt, kh, kw, M, with_res, t * 1e6, fl / t / 1e12, by / t / 1e9))
