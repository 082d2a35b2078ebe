#!/usr/bin/env python3
"""GPU probe: contract-grade (split-plane) SlowFast throughput and per-layer time (HIP events around every launch).
usage: probe_x3.py [mode=f16x3] [batch=64]"""
import collections
import sys
import time

import torch

sys.path.insert(0, ".")
import avtex  # noqa: E402
import avtex.fused_slowfast as fsf  # noqa: E402
from avtex import ops  # noqa: E402
from avtex.slowfast import SlowFast  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
b = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
torch.manual_seed(0)
for arg in sys.argv[4:]:  # e.g. pwskip=256x1024: that pointwise layer leaves the streaming kernel for the general / 256 x 256 tile
    if arg.startswith("res2="):  # res2=0: the slow res2 identity blocks as three launches (round 5's path)
        fsf._RES2_X3 = int(arg[5:])
    if arg.startswith("pwskip="):
        fsf._PW_X3_SKIP = set(fsf._PW_X3_SKIP) | {tuple(int(v) for v in p.split("x")) for p in arg[7:].split(",")}
m = fsf.SlowFastMFMA(SlowFast(), dev, precision=mode)
pd = fsf.PRECISIONS[mode]
if len(sys.argv) > 3 and sys.argv[3] == "table":  # the product's input form: every distinct frame packed once + the windows' index
    vid = torch.randint(0, 256, (b * 4 + 20, 128, 128, 3), dtype=torch.uint8, device=dev)
    import numpy as np
    slow, fast = ops.clip_pack_frames(vid, np.arange(b, dtype=np.int64) * 4, 20, out_hw=224, planes=mode)
else:
    mk = lambda t_: ops.SplitClip(*fsf.split_planes(torch.randn(b, t_, 224, 224, 4, device=dev), pd), pd)
    slow, fast = mk(8), mk(32)
for _ in range(2):
    y = m.forward_ndhwc4(slow, fast)
torch.cuda.synchronize()
t0 = time.time()
n = 3
for _ in range(n):
    y = m.forward_ndhwc4(slow, fast)
torch.cuda.synchronize()
per = (time.time() - t0) / n
print("%s batch=%d: %.4fs/batch -> %.1f clips/s, %.1f TFLOP/s algorithmic (%.3f of %d)" % (
    mode, b, per, b / per, b * 100.6e9 / per / 1e12, b * 100.6e9 / per / 1e12 / 833.3, 833), flush=True)
recs, shapes = [], []
orig = ops.conv3d_igemm_x3


def spy(x_ptrs, wt_hi, wt_lo, bias, res_ptrs, out_ptrs, ktab, dims, cin, cout, kernel, stride, pad, ldi, ldo, ldr, relu,
        plane_dtype, wscale=None, out_dims=(0, 0, 0), out_rows=None, wblk=False):
    shapes.append("cin%d cout%d k%s s%s in%s%s%s" % (cin, cout, kernel, stride, tuple(dims), " +res" if res_ptrs else "", " wblk" if wblk else ""))
    return orig(x_ptrs, wt_hi, wt_lo, bias, res_ptrs, out_ptrs, ktab, dims, cin, cout, kernel, stride, pad, ldi, ldo, ldr,
                relu, plane_dtype, wscale, out_dims, out_rows, wblk)


def hook(name, launch, flops, nbytes):
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    launch()
    e.record()
    recs.append((a, e, flops, nbytes, name))


orig_pw = ops.pw_x3


def spy_pw(x_ptrs, ldx, k, w_hi, w_lo, bias, wscale, res_ptrs, ldr, y_ptrs, ldy, n, m_, relu, plane_dtype):
    shapes.append("pointwise cin%d cout%d rows %d%s" % (k, n, m_, " +res" if res_ptrs else ""))
    return orig_pw(x_ptrs, ldx, k, w_hi, w_lo, bias, wscale, res_ptrs, ldr, y_ptrs, ldy, n, m_, relu, plane_dtype)


orig_stem = ops.stem_conv_x3


def spy_stem(x_ptrs, wt_hi, wt_lo, bias, wscale, out_ptrs, batch, t, h, pw, cout, kt, st, pt, plane_dtype, relu=True, frames_per_tile=0,
             frame_idx=None, table_frames=0):
    shapes.append("stem (LDS patch) cout%d kt%d st%d in(%d, %d, %d, %d)%s%s" % (cout, kt, st, batch, t, h, pw, " frame-major" if frames_per_tile else "",
                                                                               " via a table of %d frames" % table_frames if frame_idx is not None else ""))
    return orig_stem(x_ptrs, wt_hi, wt_lo, bias, wscale, out_ptrs, batch, t, h, pw, cout, kt, st, pt, plane_dtype, relu, frames_per_tile,
                     frame_idx, table_frames)


orig_stem_m = ops.stem_conv_x3_merged


def spy_stem_m(x_ptrs, wt_hi, wt_lo, bias, wscale, out_ptrs, batch, t, h, pw, cout, kt, st, pt, plane_dtype, tap_frames, tap_tiles, ktm,
               table_frames, relu=True):
    shapes.append("stem (LDS patch) cout%d kt%d st%d in(%d, %d, %d, %d) frame-major via a table of %d frames, %d merged taps per group" % (
        cout, kt, st, batch, t, h, pw, table_frames, ktm))
    return orig_stem_m(x_ptrs, wt_hi, wt_lo, bias, wscale, out_ptrs, batch, t, h, pw, cout, kt, st, pt, plane_dtype, tap_frames, tap_tiles,
                       ktm, table_frames, relu)


orig_lat = ops.lateral_x3


def spy_lat(x_ptrs, ldx, cin, w_hi, w_lo, bias, wscale, y_ptrs, ldy, cout, batch, t, hw, kt, st, pt, relu, plane_dtype):
    shapes.append("lateral (streaming) cin%d cout%d k(%d, 1, 1) s(%d, 1, 1) in(%d, %d, hw %d)" % (cin, cout, kt, st, batch, t, hw))
    return orig_lat(x_ptrs, ldx, cin, w_hi, w_lo, bias, wscale, y_ptrs, ldy, cout, batch, t, hw, kt, st, pt, relu, plane_dtype)


ops.lateral_x3 = spy_lat
ops.stem_conv_x3_merged = spy_stem_m
pool_recs = []
orig_pool = ops.maxpool_hw3s2_x3


def spy_pool(x_ptrs, out_ptrs, bt, h, w, c, ldi, ldo, plane_dtype, tgroup=1, frame_idx=None):
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    r = orig_pool(x_ptrs, out_ptrs, bt, h, w, c, ldi, ldo, plane_dtype, tgroup, frame_idx)
    e.record()
    pool_recs.append((a, e, "maxpool 3x3/2 x3 in(%d, %d, %d, %d)" % (bt, h, w, c), 4.0 * bt * c * (h * w + (h // 2) * (w // 2))))
    return r


orig_bn = ops.bneck_x3


def spy_bn(x_ptrs, out_ptrs, packed, batch, t, h, w, cin, c, plane_dtype, tchunk=16):
    shapes.append("fused bottleneck cin%d c%d in(%d, %d, %d, %d) tchunk %d" % (cin, c, batch, t, h, w, tchunk))
    return orig_bn(x_ptrs, out_ptrs, packed, batch, t, h, w, cin, c, plane_dtype, tchunk)


orig_c33 = ops.conv33_x3


def spy_c33(x_ptrs, packed, out_ptrs, batch, t, h, w, ldi, ldo, plane_dtype, relu=True):
    shapes.append("conv33 direct cin64 cout64 in(%d, %d, %d, %d) ldi %d ldo %d" % (batch, t, h, w, ldi, ldo))
    return orig_c33(x_ptrs, packed, out_ptrs, batch, t, h, w, ldi, ldo, plane_dtype, relu)


orig_pc = ops.pw_chain_x3


def spy_pc(x_ptrs, ldx, k1, w1, bias1, wscale1, res_ptrs, ldr, y_ptrs, ldy, n1, relu1, w2, bias2, wscale2, z_ptrs, ldz, n2, m_, plane_dtype):
    shapes.append("pointwise chain cin%d -> %d (+res) -> %d rows %d" % (k1, n1, n2, m_))
    return orig_pc(x_ptrs, ldx, k1, w1, bias1, wscale1, res_ptrs, ldr, y_ptrs, ldy, n1, relu1, w2, bias2, wscale2, z_ptrs, ldz, n2, m_, plane_dtype)


orig_r2 = ops.res2_x3


def spy_r2(x_ptrs, ldi, out_ptrs, ldo, packed, batch, t, h, w, plane_dtype):
    shapes.append("fused slow res2 block 256 -> 64 -> 64 -> 256 (+x) in(%d, %d, %d, %d) ldi %d ldo %d" % (batch, t, h, w, ldi, ldo))
    return orig_r2(x_ptrs, ldi, out_ptrs, ldo, packed, batch, t, h, w, plane_dtype)


ops.res2_x3 = spy_r2
ops.pw_chain_x3 = spy_pc
ops.conv33_x3 = spy_c33
ops.bneck_x3 = spy_bn
ops.conv3d_igemm_x3 = spy
ops.pw_x3 = spy_pw
ops.stem_conv_x3 = spy_stem
ops.maxpool_hw3s2_x3 = spy_pool
fsf.PROFILER = hook
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
y = m.forward_ndhwc4(slow, fast)
e1.record()
torch.cuda.synchronize()
tot = e0.elapsed_time(e1)
agg = collections.OrderedDict()
for (a, e, fl, byt, sym), name in zip(recs, shapes):
    t = a.elapsed_time(e)
    d = agg.setdefault(name + "  " + sym.split(",f16")[0].split(",bf16")[0].replace("conv_x3_kernel", "tile").replace("<f16>", "").replace("<bf16>", "").replace("<x3>", ""), [0, 0.0, 0.0, 0.0])
    d[0] += 1
    d[1] += t
    d[2] += fl
    d[3] += byt
conv_ms = sum(v[1] for v in agg.values())
print("batch %d forward %.2f ms; conv launches %.2f ms (%d launches); pools+head+glue %.2f ms" % (b, tot, conv_ms, len(recs), tot - conv_ms))
for name, (n_, t, fl, byt) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%7.3f ms x%d  %-74s %7.1f TF/s %7.0f GB/s" % (t, n_, name, fl / t / 1e9, byt / t / 1e6))
for a, e, name, byt in pool_recs:
    t = a.elapsed_time(e)
    print("%7.3f ms     %-74s %7s      %7.0f GB/s" % (t, name, "", byt / t / 1e6))
