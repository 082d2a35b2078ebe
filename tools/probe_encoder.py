#!/usr/bin/env python3
"""GPU probe: SlowFast-8x8-R50 forward throughput on MIOpen by dtype / memory format / batch."""
import sys
import time

import torch

sys.path.insert(0, ".")
import avtex  # noqa: E402
from avtex.slowfast import SlowFast  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
cfgs = [(torch.bfloat16, True, 16), (torch.bfloat16, False, 16), (torch.float16, True, 16), (torch.float32, False, 8),
        (torch.bfloat16, True, 32), (torch.bfloat16, True, 64)]
if len(sys.argv) > 1:
    cfgs = [cfgs[int(i)] for i in sys.argv[1:]]
for dt, cl, b in cfgs:
    try:
        m = SlowFast().to(dev, dt).eval()
        if cl:
            m = m.to(memory_format=torch.channels_last_3d)
        slow = torch.randn(b, 3, 8, 224, 224, device=dev, dtype=dt)
        fast = torch.randn(b, 3, 32, 224, 224, device=dev, dtype=dt)
        if cl:
            slow = slow.contiguous(memory_format=torch.channels_last_3d)
            fast = fast.contiguous(memory_format=torch.channels_last_3d)
        with torch.no_grad():
            t0 = time.time()
            y = m([slow, fast]); torch.cuda.synchronize()
            t1 = time.time()
            for _ in range(2):
                y = m([slow, fast])
            torch.cuda.synchronize()
            t2 = time.time()
            n = 3
            for _ in range(n):
                y = m([slow, fast])
            torch.cuda.synchronize()
            t3 = time.time()
        per = (t3 - t2) / n
        print("dtype=%s channels_last=%s batch=%d first=%.2fs steady=%.3fs/batch -> %.1f clips/s, %.1f TFLOP/s"
              % (dt, cl, b, t1 - t0, per, b / per, b * 100.6e9 / per / 1e12), flush=True)
        del m, slow, fast, y
        torch.cuda.empty_cache()
    except Exception as e:  # noqa
        print("dtype=%s channels_last=%s batch=%d FAILED: %r" % (dt, cl, b, e), flush=True)
