#!/usr/bin/env python3
"""GPU experiment: the q and the t encoder on two HIP streams with COMPLEMENTARY CU masks (hipExtStreamCreateWithCUMask), so that
one encoder's HBM-bound layers (pointwise / fused bottlenecks / stems: ~60 % of a forward) run beside the other's matrix-bound
ones (the long-K tile) on disjoint halves of the chip, instead of one kernel after the other on all of it.
usage: probe_cumask.py [batch=166] [reps=3]"""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, ".")
import avtex  # noqa: E402
import avtex.fused_slowfast as fsf  # noqa: E402
from avtex import ops  # noqa: E402
from avtex.slowfast import SlowFast  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 166
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
torch.manual_seed(0)
q = fsf.SlowFastMFMA(SlowFast(), dev, precision="f16x3")
torch.manual_seed(1)
t = fsf.SlowFastMFMA(SlowFast(), dev, precision="f16x3")
pd = fsf.PRECISIONS["f16x3"]
mk = lambda t_: ops.SplitClip(*fsf.split_planes(torch.randn(b, t_, 224, 224, 4, device=dev), pd), pd)
slow, fast = mk(8), mk(32)
hb = b // 2
mkh = lambda t_: ops.SplitClip(*fsf.split_planes(torch.randn(hb, t_, 224, 224, 4, device=dev), pd), pd)
slow_h, fast_h = mkh(8), mkh(32)
hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = C.c_int


def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[sum(1 << k for k in range(32) if (32 * w + k) in bits) for w in range(8)])
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, "hipExtStreamCreateWithCUMask rc %d" % rc
    return torch.cuda.ExternalStream(s.value, device=dev)


def run(streams, offset=False, n=reps):
    """n forwards of each encoder; two streams: q on streams[0], t on streams[1] (t starts half a forward late when offset)."""
    torch.cuda.synchronize()
    t0 = time.time()
    if streams is None:
        if offset:
            q.forward_ndhwc4(slow_h, fast_h)
        for _ in range(n):
            q.forward_ndhwc4(slow, fast)
            t.forward_ndhwc4(slow, fast)
    else:
        main = torch.cuda.current_stream()
        for st in streams:
            st.wait_stream(main)
        if offset:  # half a forward of lead for q: the two encoders then run out of phase (one in its matrix-bound stages while the other streams)
            with torch.cuda.stream(streams[0]):
                q.forward_ndhwc4(slow_h, fast_h)
        for i in range(n):
            with torch.cuda.stream(streams[0]):
                q.forward_ndhwc4(slow, fast)
            with torch.cuda.stream(streams[1]):
                t.forward_ndhwc4(slow, fast)
        for st in streams:
            main.wait_stream(st)
    torch.cuda.synchronize()
    return (time.time() - t0) / n


run(None, n=1)
base = run(None)
print("one stream, whole chip: %.2f ms per q+t forward of %d clips -> %.0f clip-windows/s" % (base * 1e3, b, b / base), flush=True)
cfgs = {
    "two plain streams": lambda: [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)],
    "masks: bits 0-127 | 128-255": lambda: [masked_stream(set(range(0, 128))), masked_stream(set(range(128, 256)))],
    "masks: even | odd bits": lambda: [masked_stream(set(range(0, 256, 2))), masked_stream(set(range(1, 256, 2)))],
    "masks: 16-bit blocks alternating": lambda: [masked_stream({i for i in range(256) if (i // 16) % 2 == 0}),
                                                   masked_stream({i for i in range(256) if (i // 16) % 2 == 1})],
}
base_off = run(None, offset=True)
print("one stream + the half-batch lead: %.2f ms per q+t forward (the lead's share included)" % (base_off * 1e3), flush=True)
for name, mkst in cfgs.items():
    try:
        sts = mkst()
        run(sts, n=1)
        dt = run(sts)
        dto = run(sts, offset=True)
        print("%-36s %.2f ms per q+t forward -> %.0f clip-windows/s (%.3f x); out of phase: %.2f ms (%.3f x of the one-stream run with the same lead)" % (
            name, dt * 1e3, b / dt, base / dt, dto * 1e3, base_off / dto), flush=True)
    except Exception as e:  # noqa: BLE001
        print("%-36s failed: %s" % (name, e), flush=True)
