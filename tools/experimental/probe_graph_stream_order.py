#!/usr/bin/env python3
"""GPU probe (round 6): why does the graph-replayed one-item config-5 step run at 52.5 ms as the last leg of the default bench run and at
65 ms in a process of its own?  Variants, each in a process of its own (argv[1]):
  alone        bench.train_bench(--train-items 1 --train-graph 1)
  after_eager  the eager one-item leg first, then the graphed one (what the default bench run does)
  pre_streams  the package's side streams created (and used once) before anything else, then the graphed leg
  after_8      the 8-item eager leg first, then the graphed one-item leg"""
import gc
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "alone"
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
torch.backends.cudnn.benchmark = True
base = ["--mode", "train", "--steps", "12", "--warmup", "4"]


def leg(extra):
    args = bench.build_parser().parse_args(base + extra)
    r = bench.train_bench(args, 0, 1, dev)
    gc.collect()
    torch.cuda.empty_cache()
    return r


if variant == "after_eager":
    e = leg(["--train-items", "1"])
    print("  eager one item first: %.1f clips/s" % e["value"])
elif variant == "after_8":
    e = leg(["--steps", "3", "--warmup", "2"])
    print("  eager eight items first: %.1f clips/s" % e["value"])
elif variant == "pre_streams":
    from avtex import ops

    for s in ops.side_streams(dev, 3):
        with torch.cuda.stream(s):
            torch.zeros(1024, device=dev).add_(1.0)
    torch.cuda.synchronize()
elif variant.startswith("dummy"):  # dummyK: K extra streams (used once) before the graphed leg
    keep = [torch.cuda.Stream(device=dev) for _ in range(int(variant[5:]))]
    for s in keep:
        with torch.cuda.stream(s):
            torch.zeros(1024, device=dev).add_(1.0)
    torch.cuda.synchronize()
g = leg(["--train-items", "1", "--train-graph", "1"])
print("%s: graphed one-item step %.1f clips/s, %.2f ms per step (%s)" % (variant, g["value"], g["ms_per_step"], g["config"]["hip_graph"]))
