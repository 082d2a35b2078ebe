#!/usr/bin/env python3
"""Encoder precision vs the contract (north_star: similarity within 1e-3, stitch indices bit-exact, ON THE SAME FRAMES).
Real SlowFast-8x8-R50 x2 (random init, BN randomised + calibrated), structured video, N windows at 224^2:
tables from (a) the bf16 MFMA runner, (b) the fp32 nn.Module on MIOpen, [(c) the split-bf16 (x3) MFMA runner], plus
the fp32 noise floor (the same fp32 module on CPU for a few windows).  Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avtex  # noqa: E402
from avtex import agreement, ops, synth  # noqa: E402
from avtex.fused_slowfast import SlowFastMFMA  # noqa: E402
from avtex.slowfast import SlowFast  # noqa: E402
from avtex.texture import TextureEngine  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["bf16"]
    dev = torch.device("cuda:0")
    W, S = 20, 4
    video = synth.structured_video(5, n * S + W, 128, 128)
    torch.manual_seed(0)
    q_mod = synth.randomise_bn(SlowFast().eval(), 10, 0.5)
    torch.manual_seed(1)
    t_mod = synth.randomise_bn(SlowFast().eval(), 11, 0.5)
    q_mod, t_mod = q_mod.to(dev), t_mod.to(dev)
    frames = video.to(dev)
    cal = np.linspace(0, n - 1, 8).astype(np.int64) * S
    slow, fast = ops.clip_pack(frames, cal, W, out_hw=224, dtype=torch.float32)
    synth.calibrate_bn(q_mod, slow, fast)
    synth.calibrate_bn(t_mod, slow, fast)

    def tables(qe, te, batch):
        eng = TextureEngine(qe, te, None, window=W, stride=S, temp=0.1, img_size=224, model_type=1, device=dev,
                            enc_batch=batch)
        assert eng.set_video(video) == n
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        qv, tv = eng.build_tables()
        torch.cuda.synchronize()
        return qv.clone(), tv.clone(), time.perf_counter() - t0

    res = {}
    q32, t32, s32 = tables(q_mod.float().eval(), t_mod.float().eval(), 16)
    res["fp32_module_s"] = s32
    for mode in modes:
        if mode == "bf16":
            qe, te = SlowFastMFMA(q_mod, dev), SlowFastMFMA(t_mod, dev)
        else:
            qe, te = SlowFastMFMA(q_mod, dev, precision=mode), SlowFastMFMA(t_mod, dev, precision=mode)
        qv, tv, s = tables(qe, te, 32)
        r = agreement.compare_tables(qv, tv, q32, t32, 0.1, W, S)
        r["seconds"] = s
        res[mode + "_vs_fp32_module"] = r
    # noise floor between two fp32 implementations: the module on CPU (oneDNN) for a few windows
    k = min(6, n)
    slow, fast = ops.clip_pack(frames, np.arange(k) * S, W, out_hw=224, dtype=torch.float32)
    qc = q_mod.cpu().float()
    with torch.no_grad():
        e_cpu = qc([slow.cpu(), fast.cpu()])
    q_mod.to(dev)
    rel = ((e_cpu - q32[:k].cpu()).norm(dim=1) / e_cpu.norm(dim=1)).max().item()
    res["fp32_cpu_vs_fp32_gpu_rel_embedding_err"] = rel
    print(json.dumps(res))


if __name__ == "__main__":
    main()
