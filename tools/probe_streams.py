#!/usr/bin/env python3
"""GPU probe: q- and t-encoder forwards on one stream vs two concurrent streams (64 clips)."""
import sys, time
import torch
sys.path.insert(0, ".")
import avtex
from avtex.slowfast import SlowFast
from avtex.fused_slowfast import SlowFastMFMA
dev = torch.device("cuda:0")
torch.manual_seed(0); q = SlowFastMFMA(SlowFast(), dev)
torch.manual_seed(1); t = SlowFastMFMA(SlowFast(), dev)
b = 64
slow = torch.randn(b, 8, 224, 224, 4, device=dev, dtype=torch.bfloat16); fast = torch.randn(b, 32, 224, 224, 4, device=dev, dtype=torch.bfloat16)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def seq():
    return q.forward_ndhwc4(slow, fast), t.forward_ndhwc4(slow, fast)
def par():
    main = torch.cuda.current_stream()
    s1.wait_stream(main); s2.wait_stream(main)
    with torch.cuda.stream(s1): a = q.forward_ndhwc4(slow, fast)
    with torch.cuda.stream(s2): c = t.forward_ndhwc4(slow, fast)
    main.wait_stream(s1); main.wait_stream(s2)
    return a, c
for name, fn in (("one stream", seq), ("two streams", par), ("one stream", seq), ("two streams", par)):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(5): out = fn()
    torch.cuda.synchronize(); per = (time.time() - t0) / 5
    print("%s: %.2f ms per pair of forwards -> %.1f windows/s" % (name, per * 1e3, b / per), flush=True)
a, c = seq(); a2, c2 = par(); torch.cuda.synchronize()
print("same results:", torch.equal(a, a2), torch.equal(c, c2))
