"""Debug: the input-gradient launch with BatchNorm backward statistics against torch, piece by piece."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn as nn
import avtex
from avtex import train_ops, ops

dev = "cuda:0"
torch.manual_seed(0)
c, cout, kernel, pad, dims, groups = 64, 64, (1, 3, 3), (0, 1, 1), (4, 2, 14, 14), 2
b, t, h, w = dims
x = (torch.randn(b, c, t, h, w, device=dev) * 1.5 + 0.4).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
conv = nn.Conv3d(c, cout, kernel, padding=pad, bias=False).to(dev).to(memory_format=torch.channels_last_3d).train()
bn = nn.BatchNorm3d(c).to(dev).train()
gy = torch.randn(b, cout, t, h, w, device=dev).contiguous(memory_format=torch.channels_last_3d)
seen = {}
orig = train_ops._BNAct.backward
def spy(ctx, dy):
    seen["dy"] = dy.detach().clone()
    seen["pre"] = dy.data_ptr() in train_ops._BWD_STATS
    out = orig(ctx, dy)
    seen["dx"] = out[0].detach().clone()
    return out
train_ops._BNAct.backward = staticmethod(spy)
res = {}
for epi in (1, 0):
    train_ops._EPI_BWD = epi
    x.grad = None
    with train_ops.bn_replicas(groups):
        y = train_ops.bn_act(x, bn, relu=True)
        z = train_ops.conv3d(y, conv)
    if epi:
        handle = y._avt_bn
    (z * gy).sum().backward()
    torch.cuda.synchronize()
    res[epi] = dict(seen, xg=x.grad.clone(), y=y.detach().clone())
    print("epi", epi, "pre", seen["pre"], "dy norm", float(seen["dy"].norm()), "dx norm", float(seen["dx"].norm()), "nan", bool(torch.isnan(seen["dx"]).any()))
mask = (res[0]["y"] > 0).float()
print("masked ref dy norm", float((res[0]["dy"] * mask).norm()), " fused dy vs masked ref", float((res[1]["dy"] - res[0]["dy"] * mask).norm()))
print("dx diff", float((res[1]["dx"] - res[0]["dx"]).norm()), "ref", float(res[0]["dx"].norm()))

# direct kernel calls: the convolution of gy with the transposed filter, relu off / on
planes = train_ops._weight_planes(conv.weight, True)
tab = train_ops._ktab(cout, kernel, h, w, cout, dev)
save = None
h_ = handle
for relu in (0, 1):
    out = torch.empty_like(x)
    bnargs = (h_.x, h_.mean, h_.invstd, h_.weight, h_.bias, None, relu)
    st = ops.conv3d_igemm_x3_f32_bwdstats(gy.permute(0, 2, 3, 4, 1), planes[0], planes[1], out.permute(0, 2, 3, 4, 1), tab, (b, t, h, w), cout, c,
                                          kernel, pad, ops.X3_BF16, bnargs, groups, c)
    torch.cuda.synchronize()
    print("relu", relu, "st", None if st is None else st[1], "out norm", float(out.norm()), "vs unmasked", float((out - res[0]["dy"]).norm()),
          "vs masked", float((out - res[0]["dy"] * mask).norm()))
print("mean", h_.mean[:4].tolist(), "invstd", h_.invstd[:4].tolist(), "x mean ch0", float(h_.x[:, 0].mean()))

import numpy as np
# (a) mask bits all ones -> nothing masked; (b) statistics of the relu-off run against torch
ones = torch.full((x.numel() // 4,), 0x0F, dtype=torch.uint8, device=dev)
out = torch.empty_like(x)
st = ops.conv3d_igemm_x3_f32_bwdstats(gy.permute(0, 2, 3, 4, 1), planes[0], planes[1], out.permute(0, 2, 3, 4, 1), tab, (b, t, h, w), cout, c,
                                      kernel, pad, ops.X3_BF16, (h_.x, h_.mean, h_.invstd, h_.weight, None, ones, 1), groups, c)
torch.cuda.synchronize()
print("bits all ones: out norm", float(out.norm()), "vs unmasked", float((out - res[0]["dy"]).norm()))
out = torch.empty_like(x)
st = ops.conv3d_igemm_x3_f32_bwdstats(gy.permute(0, 2, 3, 4, 1), planes[0], planes[1], out.permute(0, 2, 3, 4, 1), tab, (b, t, h, w), cout, c,
                                      kernel, pad, ops.X3_BF16, (h_.x, h_.mean, h_.invstd, h_.weight, h_.bias, None, 0), groups, c)
torch.cuda.synchronize()
part = st[0][: groups * st[1] * (c // 4) * 8 * 8].view(torch.float64).view(groups, st[1], c // 4, 2, 4).sum(1)   # [g][quad][stat][e]
s0 = part[:, :, 0, :].reshape(groups, c); s1 = part[:, :, 1, :].reshape(groups, c)
dyr = res[0]["dy"].permute(0, 2, 3, 4, 1).reshape(groups, -1, c).double()
xr = x.detach().permute(0, 2, 3, 4, 1).reshape(groups, -1, c).double()
xhat = (xr - h_.mean.view(groups, 1, c).double()) * h_.invstd.view(groups, 1, c).double()
print("sum g   kernel", s0[0, :3].tolist(), "torch", dyr.sum(1)[0, :3].tolist())
print("sum gxh kernel", s1[0, :3].tolist(), "torch", (dyr * xhat).sum(1)[0, :3].tolist())
print("if x were 0:", (dyr * ((0 - h_.mean.view(groups, 1, c).double()) * h_.invstd.view(groups, 1, c).double())).sum(1)[0, :3].tolist())

big = torch.full_like(h_.bias, 100.0)
for relu, beta, mk, name in ((1, big, None, "relu1 beta+100"), (2, big, None, "relu2 beta+100"), (1, None, ones, "relu1 ones"), (7, None, ones, "relu7 ones")):
    out = torch.full_like(x, 5.0)
    st = ops.conv3d_igemm_x3_f32_bwdstats(gy.permute(0, 2, 3, 4, 1), planes[0], planes[1], out.permute(0, 2, 3, 4, 1), tab, (b, t, h, w), cout, c,
                                          kernel, pad, ops.X3_BF16, (h_.x, h_.mean, h_.invstd, h_.weight, beta, mk, relu), groups, c)
    torch.cuda.synchronize()
    o = out.permute(0, 2, 3, 4, 1).reshape(-1, c)
    print(name, "out norm", float(out.norm()), "first", o[0, :4].tolist(), "ref", res[0]["dy"].permute(0, 2, 3, 4, 1).reshape(-1, c)[0, :4].tolist())
