#!/usr/bin/env python3
"""Per-layer A/B of the weight gradient's tiles: the 128-wide tile against the 256 x 128 pipelined tile (avt_wgrad_x3_set_xl) on the
SlowFast layers of one config-5 rank (120 clips), device time by HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avtex
from avtex import ops, _lib

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 120
layers = [  # cin, cout, kernel, stride, pad, (t, h, w)
    (1024, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), (8, 14, 14)), (256, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1), (8, 14, 14)),
    (256, 1024, (1, 1, 1), (1, 1, 1), (0, 0, 0), (8, 14, 14)), (640, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), (8, 28, 28)),
    (128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1), (8, 28, 28)), (128, 512, (1, 1, 1), (1, 1, 1), (0, 0, 0), (8, 28, 28)),
    (512, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0), (8, 28, 28)), (2048, 512, (3, 1, 1), (1, 1, 1), (1, 0, 0), (8, 7, 7)),
    (512, 512, (1, 3, 3), (1, 1, 1), (0, 1, 1), (8, 7, 7)), (512, 2048, (1, 1, 1), (1, 1, 1), (0, 0, 0), (8, 7, 7)),
    (64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (8, 56, 56)), (64, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), (8, 56, 56)),
    (256, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), (8, 56, 56)), (128, 128, (3, 1, 1), (1, 1, 1), (1, 0, 0), (32, 14, 14)),
    # the fast pathway's few-channel layers and the lateral connections (the 64- / 32-wide forms of the pipelined tile)
    (8, 8, (1, 3, 3), (1, 1, 1), (0, 1, 1), (32, 56, 56)), (8, 32, (1, 1, 1), (1, 1, 1), (0, 0, 0), (32, 56, 56)),
    (32, 8, (3, 1, 1), (1, 1, 1), (1, 0, 0), (32, 56, 56)), (16, 16, (1, 3, 3), (1, 1, 1), (0, 1, 1), (32, 28, 28)),
    (64, 16, (3, 1, 1), (1, 1, 1), (1, 0, 0), (32, 28, 28)), (16, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), (32, 28, 28)),
    (32, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1), (32, 14, 14)), (128, 32, (3, 1, 1), (1, 1, 1), (1, 0, 0), (32, 14, 14)),
    (32, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0), (32, 14, 14)), (64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (32, 7, 7)),
    (256, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), (32, 7, 7)), (64, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), (32, 7, 7)),
    (8, 16, (7, 1, 1), (4, 1, 1), (3, 0, 0), (32, 56, 56)), (32, 64, (7, 1, 1), (4, 1, 1), (3, 0, 0), (32, 56, 56)),
]
tot = [0.0, 0.0]
for cin, cout, k, st, pd, (t, h, w) in layers:
    x = torch.randn(B, t, h, w, cin, device=dev)
    to, ho, wo = [(n + 2 * p - kk) // s_ + 1 for n, p, kk, s_ in zip((t, h, w), pd, k, st)]
    gy = torch.randn(B, to, ho, wo, cout, device=dev)
    dw = torch.empty((cout,) + k + (cin,), device=dev)
    ms = []
    for xl in (0, 2):
        _lib.lib().avt_wgrad_x3_set_xl(xl)
        for _ in range(2):
            ops.conv3d_wgrad_x3_f32(gy, x, dw, (B, t, h, w), cin, cout, k, st, pd, cin, cout)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            ops.conv3d_wgrad_x3_f32(gy, x, dw, (B, t, h, w), cin, cout, k, st, pd, cin, cout)
        b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b) / 5)
    _lib.lib().avt_wgrad_x3_set_xl(1)
    fl = 2.0 * B * to * ho * wo * cout * cin * k[0] * k[1] * k[2]
    tot[0] += ms[0]; tot[1] += ms[1]
    print("cin%-5d cout%-5d k%s in(%d,%d,%d,%d)  128-wide %.3f ms %.0f TF/s   256x128 %.3f ms %.0f TF/s   %+.0f %%" % (
        cin, cout, k, B, t, h, w, ms[0], fl / ms[0] / 1e9, ms[1], fl / ms[1] / 1e9, (ms[0] / ms[1] - 1) * 100))
print("sum %.2f ms -> %.2f ms" % tuple(tot))
