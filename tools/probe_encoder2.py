#!/usr/bin/env python3
"""GPU probe 2: does MIOpen find-mode (cudnn.benchmark) help SlowFast? plus per-layer-type timing."""
import sys, time, os
import torch
sys.path.insert(0, ".")
import avtex
from avtex.slowfast import SlowFast
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
for dt, cl, b in [(torch.bfloat16, False, 16), (torch.bfloat16, True, 16)]:
    m = SlowFast().to(dev, dt).eval()
    if cl: m = m.to(memory_format=torch.channels_last_3d)
    slow = torch.randn(b, 3, 8, 224, 224, device=dev, dtype=dt); fast = torch.randn(b, 3, 32, 224, 224, device=dev, dtype=dt)
    if cl:
        slow = slow.contiguous(memory_format=torch.channels_last_3d); fast = fast.contiguous(memory_format=torch.channels_last_3d)
    with torch.no_grad():
        t0 = time.time(); m([slow, fast]); torch.cuda.synchronize(); t1 = time.time()
        for _ in range(2): m([slow, fast])
        torch.cuda.synchronize(); t2 = time.time()
        for _ in range(3): m([slow, fast])
        torch.cuda.synchronize(); t3 = time.time()
    per = (t3 - t2) / 3
    print("BENCHMARK=True dtype=%s cl=%s batch=%d first=%.1fs steady=%.3fs -> %.1f clips/s %.1f TF/s" % (dt, cl, b, t1 - t0, per, b / per, b * 100.6e9 / per / 1e12), flush=True)
# plain GEMM reference point: what hipBLASLt does on a 1x1x1-conv-shaped bf16 GEMM
for (M, K, N) in [(16 * 25088, 256, 64), (16 * 25088, 64, 256), (16 * 6272, 512, 128), (16 * 1568, 1024, 256), (16 * 392, 2048, 512)]:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16); w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    for _ in range(3): y = torch.nn.functional.linear(a, w)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10): y = torch.nn.functional.linear(a, w)
    torch.cuda.synchronize(); t = (time.time() - t0) / 10
    print("linear M=%d K=%d N=%d: %.1f us, %.1f TF/s, %.0f GB/s" % (M, K, N, t * 1e6, 2 * M * K * N / t / 1e12, (M * K + M * N) * 2 / t / 1e9), flush=True)
