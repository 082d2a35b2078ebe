#!/usr/bin/env python3
"""One contract-grade forward on a frame table (for rocprofv3 --pmc passes over the encoder kernels only).
usage: rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d out -- python3 tools/pmc_x3_forward.py [batch=166]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avtex  # noqa: E402
from avtex import ops  # noqa: E402
from avtex.fused_slowfast import SlowFastMFMA  # noqa: E402
from avtex.slowfast import SlowFast  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 166
dev = torch.device("cuda:0")
torch.manual_seed(0)
enc = SlowFastMFMA(SlowFast(), dev, precision="f16x3")
vid = torch.randint(0, 256, (b * 4 + 20, 128, 128, 3), dtype=torch.uint8, device=dev)
slow, fast = ops.clip_pack_frames(vid, np.arange(b, dtype=np.int64) * 4, 20, out_hw=224, planes="f16x3")
for _ in range(2):
    enc.forward_ndhwc4(slow, fast)
torch.cuda.synchronize()
print("done")
