#!/usr/bin/env python3
"""GPU probe: the row select on the weak-scaling shard shape of `bench.py --gpus 8` — every rank's similarity
block is 4096 x 32768 — against the widths the register-resident form covers (<= 16384 columns).  HIP-event times of row_transition
(th 0.3 / 0.0) and row_topk (k = 8), algorithmic GB/s (rows * cols * 4 B read).  usage: probe_select_wide.py [rows=4096]"""
import sys

import torch

sys.path.insert(0, ".")
import avtex  # noqa: E402,F401
from avtex import ops  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")


def timed(fn, reps=10):
    fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for cols in (4096, 8192, 16384, 20000, 32768):
    g = torch.Generator(device=dev).manual_seed(cols)
    sim = torch.randn((rows, cols), device=dev, generator=g) * 2.0 + 3.0
    q_ids = torch.arange(rows, device=dev, dtype=torch.int64)
    out, errs = {}, []
    for name, fn in (("th0.3", lambda: ops.row_transition(sim, q_ids=q_ids, threshold=0.3, cap=64)),
                     ("th0.0", lambda: ops.row_transition(sim, q_ids=q_ids, threshold=0.0, cap=64)),
                     ("topk8", lambda: ops.row_topk(sim, 8, self_col=q_ids))):
        try:
            out[name] = timed(fn)
        except Exception as e:  # (a width a kernel form does not cover: reported, not fatal)
            errs.append("%s: %s" % (name, str(e)[:120]))
    gb = rows * cols * 4.0 / 1e9
    print("%5d x %5d: " % (rows, cols) + "  ".join("%s %.3f ms (%.0f GB/s)" % (k, v, gb / (v * 1e-3)) for k, v in out.items()) +
          ("  | " + "; ".join(errs) if errs else ""), flush=True)
