"""Does it matter WHICH of torch's pool streams the training step's query encoder runs on?  (The k-th stream a process creates lands on
hardware queue k mod GPU_MAX_HW_QUEUES; sharing a queue with the step's own stream serialises the two encoders.)
    python tools/experimental/probe_side_stream_index.py <k>   -> config-5 clips/s with the first k side streams skipped"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
import avtex.models as M

k = int(sys.argv[1]) if len(sys.argv) > 1 else 0
M._SIDE_SKIP = k
args = bench.build_parser().parse_args(["--mode", "train", "--steps", "6", "--warmup", "2"])
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
torch.backends.cudnn.benchmark = True
line = bench.train_bench(args, 0, 1, dev)
print("side stream index %d: %.1f clips/s, %.1f ms per step" % (k, line["value"], line["ms_per_step"]))
