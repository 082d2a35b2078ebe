"""Which torch ops of one batched training step launch the small device copies / fills: torch.profiler CPU-side op counts."""
import os, sys, collections
from types import SimpleNamespace
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avtex as avt
from avtex import synth, train_ops
from avtex.dataset import DeviceSegmentBatcher
from avtex.slowfast import SlowFast
from torch.profiler import ProfilerActivity, profile

dev = torch.device("cuda:0")
args = SimpleNamespace(vdata="/tmp", adata=None, n_negs=14, img_size=224, enc_arch="slowfast", window=0, stride=0)
torch.manual_seed(5)
ds = avt.AudioVideoSegments(args, "x", split="train", video=(synth.structured_video(3, 600, 64, 64), 30.0))
m = avt.ContrastivePredictionTemporal(SlowFast(), SlowFast(), None, 1, 128, temp=0.1, window=ds.window, stride=ds.stride,
                                      enc_arch="slowfast", img_size=224).to(dev).train().to(memory_format=torch.channels_last_3d)
opt = torch.optim.SGD(m.parameters(), lr=1e-4, momentum=0.9, weight_decay=1e-4)
np.random.seed(3)
bat = DeviceSegmentBatcher(ds, dev).seed_from_numpy()
crit = avt.InfoNCECriterion()
items = 2
def step():
    opt.zero_grad(set_to_none=True)
    q, t, _, _ = bat.batch(torch.tensor([20, 31]))
    with train_ops.bn_replicas(items):
        loss = crit(m(q, t), torch.zeros(items, dtype=torch.long, device=dev))
    loss.backward()
    opt.step()
for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.count)
for e in rows[:40]:
    print("%-60s count %5d  cpu %8.1f us  cuda %8.1f us" % (e.key[:60], e.count, e.self_cpu_time_total, e.self_device_time_total))
# who launches the small device copies / fills: the innermost torch op around each hipMemcpyAsync / hipMemsetAsync
import collections
par = collections.Counter()
for ev in prof.events():
    if ev.name in ("hipMemcpyAsync", "hipMemcpyWithStream", "hipMemsetAsync") or "Memcpy" in ev.name:
        p_ = ev.cpu_parent
        chain = []
        while p_ is not None and len(chain) < 3:
            chain.append(p_.name[:40])
            p_ = p_.cpu_parent
        par[(ev.name[:24], " < ".join(chain))] += 1
for k, v in par.most_common(15):
    print(v, k)
