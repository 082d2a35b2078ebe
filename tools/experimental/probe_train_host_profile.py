"""cProfile of config 5's training steps (bench.py --mode train): the step is host-bound (tools/experimental/probe_train_host_bound.py: the
host never waits for the device), so this is where its time goes.   python tools/experimental/probe_train_host_profile.py"""
import cProfile
import io
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402


def main():
    # (extra command-line words are handed to bench.py's parser: `--train-items 1` profiles the one-item step)
    args = bench.build_parser().parse_args(["--mode", "train", "--steps", "6", "--warmup", "3"] + sys.argv[1:])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.backends.cudnn.benchmark = True
    pr = cProfile.Profile()
    pr.enable()
    line = bench.train_bench(args, 0, 1, dev)
    pr.disable()
    print(line["value"], line["ms_per_step"])
    for key in ("tottime", "cumulative"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(32)
        print("\n".join(l[:170] for l in s.getvalue().splitlines()[:48]))


if __name__ == "__main__":
    main()
