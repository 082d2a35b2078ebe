#!/usr/bin/env python3
"""Config 5 at ONE item per step (1 query clip + 15 target clips): which chain is the replayed step's critical path?

Three HIP graphs of the same model, each replayed alone:
  full   — the whole forward + backward (query encoder on its side stream next to the target encoder, as the product runs it)
  query  — the query encoder's forward + backward alone (ONE clip: every launch is a handful of workgroups)
  target — the target encoder's forward + backward alone (15 clips)
If query ~ full, the step waits for a chain of latency-bound launches and the wide tiles' throughput is beside the point.
usage: probe_one_item_chains.py [reps=20]"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import avtex as avt  # noqa: E402
from avtex import synth, train_ops  # noqa: E402
from avtex.dataset import DeviceSegmentBatcher  # noqa: E402
from avtex.slowfast import SlowFast  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda:0")
    args = SimpleNamespace(vdata="/tmp", adata=None, n_negs=14, img_size=224, enc_arch="slowfast", window=0, stride=0)
    torch.manual_seed(5)
    ds = avt.AudioVideoSegments(args, "x", split="train", video=(synth.structured_video(3, 600, 64, 64), 30.0))
    torch.manual_seed(0)
    m = avt.ContrastivePredictionTemporal(SlowFast(), SlowFast(), None, 1, 128, temp=0.1, window=ds.window, stride=ds.stride,
                                          enc_arch="slowfast", img_size=224)
    synth.randomise_bn(m, 4, 0.0)
    m = m.to(dev).train().to(memory_format=torch.channels_last_3d)
    np.random.seed(3)
    bat = DeviceSegmentBatcher(ds, dev).seed_from_numpy()
    q, t, _, _ = bat.batch(torch.tensor([20]))
    label = torch.zeros(1, dtype=torch.long, device=dev)
    t_flat = [t[0].reshape(-1, t[0].shape[2], t[0].shape[3], t[0].shape[-2], t[0].shape[-1]),  # models.forward's own reshape of the targets
              t[1].reshape(-1, t[1].shape[2], t[1].shape[3], t[1].shape[-2], t[1].shape[-1])]
    print("query", [tuple(v.shape) for v in q], "targets", [tuple(v.shape) for v in t_flat], flush=True)
    crit = avt.InfoNCECriterion()

    def full():
        m.zero_grad(set_to_none=True)
        with train_ops.bn_replicas(1):
            loss = crit(m(q, t), label)
        loss.backward()
        return loss.detach()

    def chain(enc, clips):
        def run():
            m.zero_grad(set_to_none=True)
            with train_ops.bn_replicas(1):
                y = m._run_enc(enc, clips)
            s = y.float().square().mean()
            s.backward()
            return s.detach()
        return run

    for name, fn in (("full", full), ("query (1 clip)", chain(m.q_encoder, q)), ("target (15 clips)", chain(m.t_encoder, t_flat)), ("full again", full)):
        g = train_ops.GraphedStep(fn, dev, warmup=3)
        for _ in range(3):
            g()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g()
        e1.record()
        torch.cuda.synchronize()
        print("%-18s %.2f ms per replay (forward + backward, no optimizer)" % (name, e0.elapsed_time(e1) / reps), flush=True)
        del g


if __name__ == "__main__":
    main()
