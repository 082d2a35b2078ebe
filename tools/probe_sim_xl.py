#!/usr/bin/env python3
"""The bf16x3 similarity on its own tile against the encoder's 256 x 256 LDS-DMA tile (ops.SIM_XL), HIP-event time per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avtex
from avtex import ops

dev = "cuda:0"
for nq, nt, d in ((4096, 4096, 2304), (2048, 16384, 2304), (2048, 2048, 2304), (4096, 4096, 1024)):
    g = torch.Generator().manual_seed(0)
    q = torch.randn((nq, d), generator=g).to(dev)
    t = (q.roll(-1, 0)[: min(nq, nt)].repeat((nt + nq - 1) // nq, 1)[:nt] + 0.1 * torch.randn((nt, d), generator=g).to(dev)).contiguous()
    _, qh, ql = ops.l2norm_rows(q, want_split=True)
    _, th, tl = ops.l2norm_rows(t, want_split=True)
    out = torch.empty((nq, nt), device=dev)
    res = []
    for xl in (False, "always"):
        ops.SIM_XL = xl
        for _ in range(3):
            ops.sim_gemm_nt(qh, th, 0.1, "bf16x3", q_lo=ql, t_lo=tl, out=out)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            ops.sim_gemm_nt(qh, th, 0.1, "bf16x3", q_lo=ql, t_lo=tl, out=out)
        b.record()
        torch.cuda.synchronize()
        res.append((a.elapsed_time(b) / 20, out.clone()))
    ops.SIM_XL = True
    fl = 2.0 * nq * nt * d
    print("Nq %5d Nt %5d D %4d: own tile %.3f ms %.0f TF/s (%.2f of 833)   256x256 LDS-DMA tile %.3f ms %.0f TF/s (%.2f)   max |d| %.2e" % (
        nq, nt, d, res[0][0], fl / res[0][0] / 1e9, fl / res[0][0] / 1e9 / 833.3, res[1][0], fl / res[1][0] / 1e9, fl / res[1][0] / 1e9 / 833.3,
        float((res[0][1] - res[1][1]).abs().max())))
