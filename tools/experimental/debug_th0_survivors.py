#!/usr/bin/env python3
"""GPU debug: what survives threshold 0.0 (validate.py:553: output < max - 0.0 * max -> 0, i.e. exact ties with the row maximum) on the
first 128 windows of the bench video, for the fp32 nn.Module encoders calibrated as bench.py does at N windows.
python tools/experimental/debug_th0_survivors.py [N]"""
import os, sys
from types import SimpleNamespace
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import avtex
from avtex import agreement, ops
from avtex.texture import TextureEngine

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
args = bench.build_parser().parse_args(["--windows", str(N)])
video, q_mod, t_mod = bench.build_inputs(args, 0, dev)
n, W, S = 128, 20, 4
eng = TextureEngine(q_mod.float(), t_mod.float(), None, window=W, stride=S, temp=0.1, img_size=224, model_type=1, device=dev, enc_batch=16)
eng.set_video(video[: n * S + W])
qv, tv = eng.build_tables()
sim = agreement._build(qv, tv, 0.1)
q_ids = torch.arange(n, device=dev, dtype=torch.int64)
r = ops.row_transition(sim, q_ids=q_ids, threshold=0.0, cap=n)
cnt = r["cnt"].cpu().numpy()
print("N", N, "survivors at th 0.0: mean %.2f, rows with more than one: %d of %d" % (cnt.mean(), int((cnt > 1).sum()), n))
row = int(np.argmax(cnt))
s = sim[row].double().cpu().numpy()
order = np.argsort(-s)[:16]
print("row", row, "survivors", cnt[row], "top scores:", ["%.7f" % s[j] for j in order])
e = np.exp(s - s.max())
print("exp(s - max) of the top:", ["%.3e" % e[j] for j in order])
print("embedding norms (fp32 q):", qv.norm(dim=1)[:6].tolist(), " nonzero dims per row:", (qv != 0).sum(1)[:6].tolist())
print("score matrix: min %.4f max %.4f; fraction of rows whose maximum is attained more than once: %.3f" % (
    float(sim.min()), float(sim.max()), float(((sim == sim.max(1, keepdim=True).values).sum(1) > 1).float().mean())))
