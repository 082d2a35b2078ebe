"""Per-layer time of the weight gradient of every distinct convolution of SlowFast-8x8-R50 at the training shapes
(15 target clips at 224^2): csrc/wgrad_x3.hip against MIOpen (aten.convolution_backward, wgrad only), channels-last fp32.
    python tools/probe_wgrad.py [batch=15]"""
import collections
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, ".")
from avtex import ops, train_ops  # noqa: E402
from avtex.slowfast import SlowFast  # noqa: E402

dev = "cuda:0"
b = int(sys.argv[1]) if len(sys.argv) > 1 else 15
net = SlowFast().to(dev).to(memory_format=torch.channels_last_3d).train()
shapes = collections.OrderedDict()


def hook(m, inp, out):
    key = (m.in_channels, m.out_channels, tuple(m.kernel_size), tuple(m.stride), tuple(m.padding), tuple(inp[0].shape))
    shapes[key] = shapes.get(key, 0) + 1


hs = [m.register_forward_hook(hook) for m in net.modules() if isinstance(m, nn.Conv3d)]
train_ops._CONV_X3 = 0
with torch.no_grad():
    net([torch.randn(b, 3, 8, 224, 224, device=dev), torch.randn(b, 3, 32, 224, 224, device=dev)])
for h in hs:
    h.remove()
rows = []
for (cin, cout, k, s, p, xs), n in shapes.items():
    if cin % 8:
        continue
    x = torch.randn(xs, device=dev).contiguous(memory_format=torch.channels_last_3d)
    w = torch.randn(cout, cin, *k, device=dev).contiguous(memory_format=torch.channels_last_3d)
    y = torch.nn.functional.conv3d(x, w, stride=s, padding=p)
    dy = torch.randn_like(y).contiguous(memory_format=torch.channels_last_3d)
    dw = torch.empty_like(w)

    def mine():
        ops.conv3d_wgrad_x3_f32(dy.permute(0, 2, 3, 4, 1), x.permute(0, 2, 3, 4, 1), dw.permute(0, 2, 3, 4, 1), (xs[0], xs[2], xs[3], xs[4]),
                                cin, cout, k, s, p, cin, cout)

    def lib():
        return torch.ops.aten.convolution_backward(dy, x, w, None, s, p, (1, 1, 1), False, (0, 0, 0), 1, [False, True, False])[1]

    def t(f):
        for _ in range(2):
            f()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        return (time.time() - t0) / 3 * 1e3

    tm, tl = t(mine), t(lib)
    fl = 2.0 * y.numel() * cin * k[0] * k[1] * k[2]
    rows.append((tm * n, tl * n, n, cin, cout, k, s, xs, fl / tm / 1e9, fl / tl / 1e9))
rows.sort(reverse=True)
print("total per %d-clip pass: wgrad_x3 %.1f ms, MIOpen %.1f ms" % (b, sum(r[0] for r in rows), sum(r[1] for r in rows)))
for r in rows:
    print("  x3 %7.3f ms  miopen %7.3f ms  x%d  cin%d cout%d k%s s%s in%s   %.0f | %.0f TF/s" % r)
