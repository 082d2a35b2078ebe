#!/usr/bin/env python3
"""CPU thread scaling of one fp32 SlowFast-8x8-R50 forward (bench.py cpu_baseline leg)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avtex
from avtex.slowfast import SlowFast
torch.manual_seed(0)
enc = SlowFast().eval()
slow, fast = torch.randn(1, 3, 8, 224, 224), torch.randn(1, 3, 32, 224, 224)
print("cpu_count", os.cpu_count(), flush=True)
for n in (32, 64, 128, 256):
    if n > (os.cpu_count() or 1):
        continue
    torch.set_num_threads(n)
    with torch.no_grad():
        t0 = time.perf_counter(); enc([slow, fast]); t1 = time.perf_counter(); enc([slow, fast]); t2 = time.perf_counter()
    print("threads %d: first %.1f s, second %.1f s" % (n, t1 - t0, t2 - t1), flush=True)
