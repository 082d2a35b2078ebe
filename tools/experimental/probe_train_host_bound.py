"""Is config 5's training step (bench.py --mode train) bound by the host's launch rate or by the device?  torch.cuda.synchronize /
Event.synchronize are wrapped to add up the time the host spends WAITING for the device inside the timed steps: a device-bound step
shows the host waiting for most of it, a host-bound one hardly at all.   python tools/experimental/probe_train_host_bound.py [bench.py arguments, e.g. --train-items 1 --steps 12 --warmup 4]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402

WAIT = [0.0, 0]


def wrap(obj, name):
    orig = getattr(obj, name)

    def f(*a, **k):
        t0 = time.perf_counter()
        out = orig(*a, **k)
        WAIT[0] += time.perf_counter() - t0
        WAIT[1] += 1
        return out

    setattr(obj, name, f)


def main():
    # extra arguments go to bench.py's parser: `--train-items 1 --steps 12 --warmup 4` = config 5's per-rank shape on 8 GPUs
    args = bench.build_parser().parse_args(["--mode", "train", "--steps", "3", "--warmup", "2"] + sys.argv[1:])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.backends.cudnn.benchmark = True
    wrap(torch.cuda, "synchronize")
    wrap(torch.cuda.Event, "synchronize")
    wrap(torch.Tensor, "item")
    wrap(torch.Tensor, "cpu")
    wrap(torch.Tensor, "__float__")  # (the step's one host read: float(step_loss))
    t0 = time.perf_counter()
    line = bench.train_bench(args, 0, 1, dev)
    wall = time.perf_counter() - t0
    print("train_bench: %.1f clips/s, %.1f ms per step; whole call %.2f s, of which the host waited on the device %.2f s in %d calls"
          % (line["value"], line["ms_per_step"], wall, WAIT[0], WAIT[1]))


if __name__ == "__main__":
    main()
