"""Per-layer device time of ONE item of config 5's step (1 query + 15 targets at 224^2, both encoders, forward + backward):
every hand-written launch of train_ops is timed with events around the call (a synchronisation per call: the sum is the
device time of the launches, not the step's wall time).  python tools/probe_train_layers.py [items]"""
import collections
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avtex as avt  # noqa: E402
from avtex import _lib, ops, synth, train_ops  # noqa: E402
from avtex.dataset import DeviceSegmentBatcher  # noqa: E402
from avtex.slowfast import SlowFast  # noqa: E402

ROWS = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0, 0.0])  # key -> [calls, ms, flops, bytes, longest call ms]
ON = [False]


def timed(key, fn, flops, nbytes):
    if not ON[0]:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    e1.synchronize()
    row = ROWS[key]
    row[0] += 1
    ms = e0.elapsed_time(e1)
    row[1] += ms
    row[4] = max(row[4], ms)
    row[2] += flops
    row[3] += nbytes
    return r


_fwd, _wg, _wgs = ops.conv3d_igemm_x3_f32, ops.conv3d_wgrad_x3_f32, ops.conv3d_wgrad_x3_sub_f32


def fwd(x, wh, wl, ws, out, ktab, dims, cin, cout, kernel, stride, pad, ldi, ldo, plane_dtype, add=None):
    mo = out.numel() // cout
    taps = kernel[0] * kernel[1] * kernel[2]
    what = "fwd  " if plane_dtype == ops.X3_F16 else "dgrad"
    key = "%s cin%-4d cout%-4d k%s s%s in%s%s" % (what, cin, cout, tuple(kernel), tuple(stride), tuple(dims), " +add" if add is not None else "")
    return timed(key, lambda: _fwd(x, wh, wl, ws, out, ktab, dims, cin, cout, kernel, stride, pad, ldi, ldo, plane_dtype, add=add),
                 2.0 * mo * taps * cin * cout, 4.0 * (x.numel() + out.numel() * (2 if add is not None else 1)))


def wg(dy, x, dw, dims, cin, cout, kernel, stride, pad, ldx, ldy):
    mo = dy.numel() // cout
    taps = kernel[0] * kernel[1] * kernel[2]
    key = "wgrad cin%-4d cout%-4d k%s s%s in%s" % (cin, cout, tuple(kernel), tuple(stride), tuple(dims))
    return timed(key, lambda: _wg(dy, x, dw, dims, cin, cout, kernel, stride, pad, ldx, ldy), 2.0 * mo * taps * cin * cout,
                 4.0 * (x.numel() + dy.numel()))


def wgs(dy, x, dw, dims, cin, cout, kernel, stride, pad, out_dims, ldx, ldy, ldw, zero_dw):
    mo = dy.numel() // cout
    taps = kernel[0] * kernel[1] * kernel[2]
    key = "wgrad-slice cin%-4d cout%-4d k%s s%s in%s" % (cin, cout, tuple(kernel), tuple(stride), tuple(dims))
    return timed(key, lambda: _wgs(dy, x, dw, dims, cin, cout, kernel, stride, pad, out_dims, ldx, ldy, ldw, zero_dw),
                 2.0 * mo * taps * cin * cout, 4.0 * (x.numel() // 2 + dy.numel()))


ops.conv3d_igemm_x3_f32, ops.conv3d_wgrad_x3_f32, ops.conv3d_wgrad_x3_sub_f32 = fwd, wg, wgs

# round 5's forms: BatchNorm statistics on the epilogues (forward: _stats, input gradient: _bwdstats), the streaming pointwise
# kernel with fp32 I/O (pw), the strided input gradient's classes (_ex)
_fs, _fb, _pw, _pws, _pwb, _fex = (ops.conv3d_igemm_x3_f32_stats, ops.conv3d_igemm_x3_f32_bwdstats, ops.pw_x3_f32, ops.pw_x3_f32_stats,
                                   ops.pw_x3_f32_bwdstats, ops.conv3d_igemm_x3_f32_ex)


def _ckey(what, cin, cout, kernel, dims, tail):
    return "%s cin%-4d cout%-4d k%s in%s%s" % (what, cin, cout, tuple(kernel), tuple(dims), tail)


def fwd_stats(x, wh, wl, ws, out, ktab, dims, cin, cout, kernel, stride, pad, ldi, ldo, plane_dtype, groups, stat_c):
    mo, taps = out.numel() // cout, kernel[0] * kernel[1] * kernel[2]
    return timed(_ckey("fwd  ", cin, cout, kernel, dims, " +stats"),
                 lambda: _fs(x, wh, wl, ws, out, ktab, dims, cin, cout, kernel, stride, pad, ldi, ldo, plane_dtype, groups, stat_c),
                 2.0 * mo * taps * cin * cout, 4.0 * (x.numel() + out.numel()))


def dgrad_bst(dy, wh, wl, out, ktab, dims, cin, cout, kernel, pad, plane_dtype, bn, groups, stat_c, add=None):
    mo, taps = out.numel() // cout, kernel[0] * kernel[1] * kernel[2]
    return timed(_ckey("dgrad", cin, cout, kernel, dims, " +bst" + (" +add" if add is not None else "")),
                 lambda: _fb(dy, wh, wl, out, ktab, dims, cin, cout, kernel, pad, plane_dtype, bn, groups, stat_c, add=add),
                 2.0 * mo * taps * cin * cout, 4.0 * (dy.numel() + out.numel() * (3 if add is not None else 2)))


def pw(x, k, wh, wl, ws, out, n, plane_dtype, add=None):
    what = "fwd  " if plane_dtype == ops.X3_F16 else "dgrad"
    return timed("%s pw cin%-4d cout%-4d rows %d%s" % (what, k, n, x.numel() // x.shape[-1], " +add" if add is not None else ""),
                 lambda: _pw(x, k, wh, wl, ws, out, n, plane_dtype, add=add), 2.0 * (out.numel() // out.shape[-1]) * k * n,
                 4.0 * (x.numel() + out.numel() * (2 if add is not None else 1)))


def pw_stats(x, k, wh, wl, ws, out, n, plane_dtype, groups):
    r = [None]

    def run():
        r[0] = _pws(x, k, wh, wl, ws, out, n, plane_dtype, groups)
        return r[0]
    return timed("fwd   pw cin%-4d cout%-4d rows %d +stats" % (k, n, x.numel() // x.shape[-1]), run,
                 2.0 * (out.numel() // out.shape[-1]) * k * n, 4.0 * (x.numel() + out.numel()))


def pw_bst(dy, k, wh, wl, out, n, plane_dtype, bn, groups, add=None):
    return timed("dgrad pw cin%-4d cout%-4d rows %d +bst%s" % (k, n, dy.numel() // dy.shape[-1], " +add" if add is not None else ""),
                 lambda: _pwb(dy, k, wh, wl, out, n, plane_dtype, bn, groups, add=add), 2.0 * (out.numel() // out.shape[-1]) * k * n,
                 4.0 * (dy.numel() + out.numel() * (3 if add is not None else 2)))


def fwd_ex(x, wh, wl, ws, out, ktab, dims, cin, cout, kernel, pad, out_dims, ldi, ldo, plane_dtype, out_rows=(1, 0, 0)):
    mo, taps = out_dims[0] * out_dims[1] * out_dims[2] * dims[0], kernel[0] * kernel[1] * kernel[2]
    return timed(_ckey("dgrad-class", cin, cout, kernel, dims, ""),
                 lambda: _fex(x, wh, wl, ws, out, ktab, dims, cin, cout, kernel, pad, out_dims, ldi, ldo, plane_dtype, out_rows=out_rows),
                 2.0 * mo * taps * cin * cout, 4.0 * (x.numel() + mo * cout))


(ops.conv3d_igemm_x3_f32_stats, ops.conv3d_igemm_x3_f32_bwdstats, ops.pw_x3_f32, ops.pw_x3_f32_stats, ops.pw_x3_f32_bwdstats,
 ops.conv3d_igemm_x3_f32_ex) = fwd_stats, dgrad_bst, pw, pw_stats, pw_bst, fwd_ex
_sf, _sw, _cp = ops.stem_conv_x3_f32, ops.stem_wgrad_x3, ops.clip_planes_f32


def stem_fwd(xh, xl, wh, wl, ws, out, batch, t, h, pw, cout, kt, st, pt, tgroup, plane_dtype, frames_per_tile=0):
    c = cout // tgroup
    key = "fwd   stem (patch) cout%-3d kt%d in%s" % (c, kt - tgroup + 1, (batch, t, h, 2 * pw))
    return timed(key, lambda: _sf(xh, xl, wh, wl, ws, out, batch, t, h, pw, cout, kt, st, pt, tgroup, plane_dtype, frames_per_tile),
                 2.0 * out.numel() * (kt - tgroup + 1) * 49 * 3, 4.0 * out.numel() + 4.0 * xh.numel())


def stem_wg(xh, xl, dy, batch, t, h, pw, cout, kt, pt):
    key = "wgrad stem (patch) cout%-3d kt%d in%s" % (cout, kt, (batch, t, h, 2 * pw))
    return timed(key, lambda: _sw(xh, xl, dy, batch, t, h, pw, cout, kt, pt), 2.0 * dy.numel() * kt * 49 * 3,
                 4.0 * dy.numel() + 4.0 * xh.numel())


def clip_planes(x, plane_dtype):
    return timed("planes clip -> pixel-pair planes %s" % (tuple(x.shape),), lambda: _cp(x, plane_dtype), 0.0, 12.0 * x.numel() / 3 + 16.0 * x.numel() / 3)


ops.stem_conv_x3_f32, ops.stem_wgrad_x3, ops.clip_planes_f32 = stem_fwd, stem_wg, clip_planes

_bn_apply = train_ops._BNAct.apply
_bf, _bb = train_ops._BNAct.forward, train_ops._BNAct.backward


def bn_fwd(ctx, x, *a, **k):
    m, c = train_ops._rows(x)
    res = a[4]
    return timed("bn_fwd  c%-4d rows %d%s" % (c, m, " +res" if res is not None else ""), lambda: _bf(ctx, x, *a, **k), 0.0,
                 4.0 * x.numel() * (3 + (1 if res is not None else 0)))


def bn_bwd(ctx, dy):
    m, c = dy.numel() // dy.shape[1], dy.shape[1]
    return timed("bn_bwd  c%-4d rows %d%s" % (c, m, " +res" if ctx.has_res else ""), lambda: _bb(ctx, dy), 0.0,
                 4.0 * dy.numel() * (6 + (1 if ctx.has_res else 0)))


train_ops._BNAct.forward, train_ops._BNAct.backward = staticmethod(bn_fwd), staticmethod(bn_bwd)


def main():
    dev = torch.device("cuda:0")
    args = SimpleNamespace(vdata="/tmp", adata=None, n_negs=14, img_size=224, enc_arch="slowfast", window=0, stride=0)
    torch.manual_seed(5)
    ds = avt.AudioVideoSegments(args, "x", split="train", video=(synth.structured_video(3, 600, 64, 64), 30.0))
    torch.manual_seed(0)
    m = avt.ContrastivePredictionTemporal(SlowFast(), SlowFast(), None, 1, 128, temp=0.1, window=ds.window, stride=ds.stride,
                                          enc_arch="slowfast", img_size=224)
    synth.randomise_bn(m, 4, 0.0)
    m = m.to(dev).train().to(memory_format=torch.channels_last_3d)
    np.random.seed(3)
    bat = DeviceSegmentBatcher(ds, dev).seed_from_numpy()
    items = int(sys.argv[1]) if len(sys.argv) > 1 else 1  # items in the pass (each a BatchNorm group of its own)
    q, t, _, _ = bat.batch(torch.tensor([20 + 7 * i for i in range(items)]))
    label = torch.zeros(items, dtype=torch.long, device=dev)
    crit = avt.InfoNCECriterion()
    for it in range(3):
        ON[0] = it == 2
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        with train_ops.bn_replicas(items):
            loss = crit(m(q, t), label)
        loss.backward()
        e1.record()
        torch.cuda.synchronize()
        print("pass %d (%d items): %.1f ms%s" % (it, items, e0.elapsed_time(e1), " (timed per launch: serialised)" if ON[0] else ""))
        m.zero_grad(set_to_none=True)
    tot = sum(r[1] for r in ROWS.values())
    print("hand-written launches: %.1f ms in %d launches" % (tot, sum(r[0] for r in ROWS.values())))
    cat = collections.defaultdict(float)
    for k, r in ROWS.items():
        cat[k.split()[0]] += r[1]
    print("  by kind: " + ", ".join("%s %.1f ms" % kv for kv in sorted(cat.items(), key=lambda kv: -kv[1])))
    for k, r in sorted(ROWS.items(), key=lambda kv: -kv[1][1])[:110]:
        print("  %7.3f ms x%-2d %-84s %6.1f TF/s %6.0f GB/s  longest %.3f" % (r[1], r[0], k, r[2] / r[1] * 1e-9, r[3] / r[1] * 1e-6, r[4]))


if __name__ == "__main__":
    main()
