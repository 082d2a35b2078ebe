#!/usr/bin/env python3
"""GPU probe: how sharp are the transition rows of the synthetic bench inputs?  For each BatchNorm sparsity setting of
synth.randomise_bn: encode n windows with both encoders (f16x3), build the matrix, report the survivors per row at th 0.3 / 0.0
and the cosine percentiles.  usage: probe_survivors.py [n=512] [sparsity ...]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import avtex  # noqa: E402
from avtex import ops, synth  # noqa: E402
from avtex.fused_slowfast import SlowFastMFMA  # noqa: E402
from avtex.slowfast import SlowFast  # noqa: E402
from avtex.texture import TextureEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sps = [float(a) for a in sys.argv[2:]] or [0.5, 1.0, 1.5, 2.0, 3.0]
import os
REL = [float(r) for r in os.environ.get("REL", "").split(",") if r]  # t encoder = perturbed copy of the q encoder
dev = torch.device("cuda:0")
W, S = 20, 4
VAR = int(os.environ.get("VARIETY", "0"))
NCAL = int(os.environ.get("NCAL", "8"))
BR = float(os.environ.get("BRANCH", "1.0"))
video = synth.structured_video(123, n * S + W, 128, 128, device=dev, variety=VAR)
print("variety %d, %d calibration clips, branch scale %.2f" % (VAR, NCAL, BR))
for sp, rel in [(sp, rel) for sp in sps for rel in (REL or [None])]:
    torch.manual_seed(0)
    q_mod = synth.randomise_bn(SlowFast().eval(), 10, sp, BR).to(dev)
    torch.manual_seed(1)
    t_mod = synth.randomise_bn(SlowFast().eval(), 11, sp, BR).to(dev) if rel is None else synth.perturbed_copy(q_mod, 11, rel)
    cal = np.linspace(0, n - 1, NCAL).astype(np.int64) * S
    slow, fast = ops.clip_pack(video, cal, W, out_hw=224, dtype=torch.float32)
    synth.calibrate_bn(q_mod, slow, fast)
    synth.calibrate_bn(t_mod, slow, fast)
    del slow, fast
    eng = TextureEngine(SlowFastMFMA(q_mod.eval(), dev, precision="f16x3"), SlowFastMFMA(t_mod.eval(), dev, precision="f16x3"), None,
                        window=W, stride=S, temp=0.1, img_size=224, model_type=1, device=dev, enc_batch=64)
    eng.set_video(video)
    qv, tv = eng.build_tables()
    qn, _, _ = ops.l2norm_rows(qv)
    tn, _, _ = ops.l2norm_rows(tv)
    sim = ops.sim_gemm_nt(qn, tn, 0.1, "f32")
    q_ids = torch.arange(n, device=dev, dtype=torch.int64)
    cos = (sim * 0.1).flatten()
    pct = torch.quantile(cos[:: max(1, cos.numel() // 100000)].float(), torch.tensor([0.01, 0.1, 0.5, 0.9, 0.99], device=dev)).tolist()
    line = "sparsity %.2f rel %s: cos pct(1,10,50,90,99) %s  max %.3f nan %d |" % (sp, rel, ["%.3f" % p for p in pct], float(cos.max()), int(torch.isnan(cos).sum()))
    for th in (0.3, 0.0):
        sel = ops.row_transition(sim, q_ids=q_ids, threshold=th, cap=n)
        line += " th %.1f: %.1f survivors/row (%.1f %%)" % (th, float(sel["cnt"].float().mean()), 100 * float(sel["cnt"].float().mean()) / (n - 1))
    print(line, flush=True)
