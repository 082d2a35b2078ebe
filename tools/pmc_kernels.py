#!/usr/bin/env python3
"""Launches every hand-written kernel at the bench.py shapes (no encoder), for rocprofv3 --pmc passes:
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python3 tools/pmc_kernels.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out -- python3 tools/pmc_kernels.py
(counters in their own passes, as MI355X_MICROARCH.md prescribes; FETCH_SIZE under-reports wide reads 2x on gfx950)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avtex
from avtex import ops
dev = torch.device("cuda:0")
N, D, W, S, B = 4096, 2304, 20, 4, 249  # B = bench.py's --enc-batch default
g = torch.Generator().manual_seed(123)
video = torch.randint(0, 256, (B * S + W, 128, 128, 3), generator=g, dtype=torch.uint8).to(dev)
starts = np.arange(B, dtype=np.int64) * S
q = torch.randn((N, D), generator=torch.Generator().manual_seed(0)).to(dev)
t = torch.randn((N, D), generator=torch.Generator().manual_seed(1)).to(dev)
q_ids = torch.arange(N, device=dev, dtype=torch.int64)
BD = 166  # dense packed clips (the bf16 path, the module path): what their kernels' 32-bit offsets take
for _ in range(3):
    ops.clip_pack(video, starts[:BD], W, out_hw=224, dtype=torch.bfloat16, layout="ndhwc4")
    ops.clip_pack(video, starts[:BD], W, out_hw=224, dtype=torch.bfloat16, layout="ncthw")
    qn, _, _ = ops.l2norm_rows(q)  # (as bench.py's default f32 similarity mode calls it: fp32 rows only)
    tn, _, _ = ops.l2norm_rows(t)
    sim = ops.sim_gemm_nt(qn, tn, 0.1, "f32")
    _, qh, ql = ops.l2norm_rows(q, want_f32=False, want_split=True)
    _, th, tl = ops.l2norm_rows(t, want_f32=False, want_split=True)
    ops.sim_gemm_nt(qh, th, 0.1, "bf16x3", q_lo=ql, t_lo=tl)
    ops.sim_gemm_nt(qh, th, 0.1, "bf16")
    ops.row_transition(sim, q_ids=q_ids, threshold=0.3, cap=64)
# one fused-encoder forward at the bench batch (128 clips): the conv3d_igemm / maxpool launches of a forward
from avtex.slowfast import SlowFast
from avtex.fused_slowfast import SlowFastMFMA
torch.manual_seed(0)
enc = SlowFastMFMA(SlowFast(), dev)
slow, fast = ops.clip_pack(video, starts[:BD], W, out_hw=224, dtype=torch.bfloat16, layout="ndhwc4")
for _ in range(2):
    enc.forward_ndhwc4(slow, fast)
del slow, fast
# ... and one forward of each contract-grade mode (split-plane kernels: conv_x3_kernel, the plane-pair pool / mean / pack)
for mode in ("f16x3", "bf16x3"):
    encx = SlowFastMFMA(SlowFast(), dev, precision=mode)
    sx, fx = ops.clip_pack_frames(video, starts, W, out_hw=224, planes=mode)  # the product's form: a frame table + the windows' index
    for _ in range(2):
        encx.forward_ndhwc4(sx, fx)
torch.cuda.synchronize()
print("done")
