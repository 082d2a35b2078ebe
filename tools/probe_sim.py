#!/usr/bin/env python3
"""GPU probe: sim_gemm_nt time per MFMA mode at N x N x D (default 4096 x 4096 x 2304)."""
import sys
import torch
sys.path.insert(0, ".")
import avtex
from avtex import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d = int(sys.argv[2]) if len(sys.argv) > 2 else 2304
dev = torch.device("cuda:0")
q = torch.randn((n, d), generator=torch.Generator().manual_seed(0)).to(dev)
t = torch.randn((n, d), generator=torch.Generator().manual_seed(1)).to(dev)
qn, qh, ql = ops.l2norm_rows(q, want_split=True)
tn, th, tl = ops.l2norm_rows(t, want_split=True)
for mode, peak in (("f32", 157.3), ("bf16x3", 2500.0 / 3), ("bf16", 2500.0)):
    fn = (lambda: ops.sim_gemm_nt(qn, tn, 0.1, "f32")) if mode == "f32" else (lambda: ops.sim_gemm_nt(qh, th, 0.1, mode, q_lo=ql, t_lo=tl))
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        fn()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    tf = 2.0 * n * n * d / (ms * 1e-3) / 1e12
    print("%-7s %.4f ms  %.1f TFLOP/s  %.3f of %.0f" % (mode, ms, tf, tf / peak, peak))
