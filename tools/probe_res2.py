#!/usr/bin/env python3
"""GPU probe: the fused slow-res2 bottleneck (csrc/res2_x3.hip) alone at the production shape — HIP-event time per launch, algorithmic
TFLOP/s and GB/s (x in + out once).  usage: probe_res2.py [batch=249] [reps=5]   (AVT_HIP_LIB selects a diagnostic build)"""
import sys

import torch

sys.path.insert(0, ".")
import avtex  # noqa: E402,F401
import avtex.fused_slowfast as fsf  # noqa: E402
from avtex import ops  # noqa: E402
from avtex.slowfast import ResBlock  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 249
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
torch.manual_seed(0)
blk = fsf._BlockX3(ResBlock(256, 256, 64, 1, 1).eval(), dev, ops.X3_F16)
dims = (b, 8, 56, 56)
m = b * 8 * 56 * 56
x = fsf.new_act(m, 256, dims, dev, True)
x.buf.view(torch.float16).normal_()
x.lo.view(torch.float16).normal_(0, 1e-3)
y = fsf.new_act(m, 256, dims, dev, True)
for _ in range(2):
    blk(x, out=y)
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    blk(x, out=y)
e.record()
torch.cuda.synchronize()
ms = a.elapsed_time(e) / reps
fl = m * 2.0 * (256 * 64 + 576 * 64 + 64 * 256)
import ctypes, os
_h = ctypes.CDLL(os.environ["AVT_HIP_LIB"]) if os.environ.get("AVT_HIP_LIB") else None
if _h is not None and hasattr(_h, "avt_debug_stamps_res2"):  # a -DR2_STAMP build: where wave 0's cycles went (summed over workgroups)
    buf = (ctypes.c_ulonglong * 10)()
    _h.avt_debug_stamps_res2(buf, 1)
    blk(x, out=y)
    torch.cuda.synchronize()
    _h.avt_debug_stamps_res2(buf, 1)
    names = ["chunk-top waits (DMA landed, LDS drain, barrier)", "phase A chunks", "A epilogue", "phase B taps", "B epilogue", "phase C MFMAs",
             "phase C epilogues + stores + next tile's loads"]
    tot = sum(buf[i] for i in range(7))
    print("  stamps: %d workgroups, %.0f cycles each; shader clock %.0f MHz" % (buf[7], tot / max(buf[7], 1), 100.0 * buf[8] / max(buf[9], 1)))
    for i, nm in enumerate(names):
        print("    %-52s %5.1f %%  (%.0f cycles / workgroup)" % (nm, 100.0 * buf[i] / tot, buf[i] / max(buf[7], 1)))
print("res2_x3 batch %d: %.3f ms per launch, %.1f TFLOP/s algorithmic (%.3f of 833), %.0f GB/s (x in + out)" % (
    b, ms, fl / ms / 1e9, fl / ms / 1e9 / 833.3, m * 2048.0 / ms / 1e6))
