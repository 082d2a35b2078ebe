#!/usr/bin/env python3
"""GPU probe: time of the fused BatchNorm (+ shortcut + ReLU) forward after a pointwise convolution with statistics on its epilogue,
shape by shape (the per-layer table of tools/probe_train_layers.py showed `bn_fwd c256 rows 188160 +res` at 2 ms per call).
python tools/experimental/probe_bn_shape.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import avtex
from avtex import train_ops

dev = torch.device("cuda:0")
torch.manual_seed(0)


def t_ms(fn, n=5):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for (cin, cout, shape) in [(64, 256, (120, 32, 7, 7)), (256, 1024, (120, 8, 14, 14)), (64, 256, (120, 8, 56, 56)), (32, 128, (120, 32, 14, 14)),
                           (128, 512, (120, 8, 28, 28))]:
    b, t, h, w = shape
    conv = torch.nn.Conv3d(cin, cout, 1, bias=False).to(dev).to(memory_format=torch.channels_last_3d).train()
    bn = torch.nn.BatchNorm3d(cout).to(dev).train()
    x = torch.randn(b, cin, t, h, w, device=dev).contiguous(memory_format=torch.channels_last_3d)
    res = torch.randn(b, cout, t, h, w, device=dev).contiguous(memory_format=torch.channels_last_3d)
    with torch.no_grad(), train_ops.bn_replicas(8):
        y = train_ops.conv3d(x, conv, stats=bn)
        pre = getattr(y, "_avt_stats", None)
        t_conv = t_ms(lambda: train_ops.conv3d(x, conv, stats=bn))
        t_plain_conv = t_ms(lambda: train_ops.conv3d(x, conv))
        t_bn_pre = t_ms(lambda: train_ops.bn_act(y, bn, res=res))
        y2 = train_ops.conv3d(x, conv)
        t_bn = t_ms(lambda: train_ops.bn_act(y2, bn, res=res))
        t_bn_nores = t_ms(lambda: train_ops.bn_act(y, bn))
    print("cin%-4d cout%-4d %-18s conv+stats %.3f ms (plain %.3f)  bn+res from partials %.3f ms (rows of partials per group: %s)  "
          "bn+res own statistics %.3f ms  bn (no res) from partials %.3f ms" % (
              cin, cout, shape, t_conv, t_plain_conv, t_bn_pre, None if pre is None else pre[1], t_bn, t_bn_nores), flush=True)
