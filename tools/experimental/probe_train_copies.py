"""Which Python lines make torch copy tensors inside one config-5 training step (bench.py --mode train)?  Tensor.contiguous /
clone / copy_ / to are wrapped to record the calling line and the bytes whenever a copy really happens; the C++-side copies
(autograd's own clones) are what is left of torch.profiler's aten::copy_ total, printed beside.
    python tools/experimental/probe_train_copies.py"""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402

SEEN = collections.defaultdict(lambda: [0, 0])


def _site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "dist-packages" not in fr.filename and "/lib/python" not in fr.filename and "probe_train_copies" not in fr.filename:
            return "%s:%d" % (os.path.basename(fr.filename), fr.lineno)
    return "?"


def wrap(name):
    orig = getattr(torch.Tensor, name)

    def f(self, *a, **k):
        out = orig(self, *a, **k)
        if self.is_cuda or (isinstance(out, torch.Tensor) and out.is_cuda):
            copied = name in ("clone", "copy_") or (isinstance(out, torch.Tensor) and out.data_ptr() != self.data_ptr())
            if copied:
                s = SEEN[(name, _site())]
                s[0] += 1
                s[1] += self.numel() * self.element_size()
        return out

    setattr(torch.Tensor, name, f)


def main():
    args = bench.build_parser().parse_args(["--mode", "train", "--steps", "1", "--warmup", "1"])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.backends.cudnn.benchmark = True
    for n in ("contiguous", "clone", "copy_", "to", "float"):
        wrap(n)
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        line = bench.train_bench(args, 0, 1, dev)
    print(line.get("value"), line.get("ms_per_step"), file=sys.stderr)
    tot = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if ev.name in ("aten::copy_", "aten::fill_", "aten::add_", "aten::index") and ev.device_time_total > 0:
            tot[ev.name][0] += 1
            tot[ev.name][1] += ev.device_time_total
    for k, (n, us) in tot.items():
        print("profiler: %-14s x%-5d %.2f ms (2 steps + setup)" % (k, n, us / 1e3))
    for (name, where), (n, b) in sorted(SEEN.items(), key=lambda kv: -kv[1][1])[:40]:
        print("%-11s x%-5d %9.1f MB  %s" % (name, n, b / 1e6, where))


if __name__ == "__main__":
    main()
