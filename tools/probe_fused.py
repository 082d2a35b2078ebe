#!/usr/bin/env python3
"""GPU probe: fused MFMA SlowFast throughput and per-layer time (HIP events around every conv launch)."""
import sys, time, collections
import torch
sys.path.insert(0, ".")
import avtex
from avtex import ops
from avtex.slowfast import SlowFast
from avtex.fused_slowfast import SlowFastMFMA, FusedConv
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
torch.manual_seed(0)
m = SlowFastMFMA(SlowFast(), dev)
import os
NDHWC = True
for b in [int(x) for x in (sys.argv[1:] or ["16", "32"])]:
    slow = torch.randn(b, 8, 224, 224, 4, device=dev, dtype=torch.bfloat16); fast = torch.randn(b, 32, 224, 224, 4, device=dev, dtype=torch.bfloat16)
    for _ in range(2): y = m.forward_ndhwc4(slow, fast)
    torch.cuda.synchronize(); t0 = time.time()
    n = 5
    for _ in range(n): y = m.forward_ndhwc4(slow, fast)
    torch.cuda.synchronize(); per = (time.time() - t0) / n
    print("fused batch=%d: %.4fs/batch -> %.1f clips/s, %.1f TFLOP/s" % (b, per, b / per, b * 100.6e9 / per / 1e12), flush=True)
# per-layer timing through the launch observer (leaf launches only)
import avtex.fused_slowfast as fsf
b = int(os.environ.get('PROBE_B', '16'))
slow = torch.randn(b, 8, 224, 224, 4, device=dev, dtype=torch.bfloat16); fast = torch.randn(b, 32, 224, 224, 4, device=dev, dtype=torch.bfloat16)
recs = []
def hook(name, launch, flops, nbytes):
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); launch(); e.record()
    recs.append((a, e, flops, nbytes))
shapes = []
orig = avtex.ops.conv3d_igemm
def spy(x_ptr, wt, bias, res_ptr, out_ptr, ktab, dims, cin, cout, kernel, stride, pad, ldi, ldo, ldr, relu, out_dims=(0, 0, 0), out_rows=None, wfrag=None):
    shapes.append("cin%d cout%d k%s s%s in%s%s%s%s" % (cin, cout, kernel, stride, tuple(dims), " +res" if res_ptr else "", " ->rows x2" if out_rows else "", " XB" if wfrag is not None else ""))
    return orig(x_ptr, wt, bias, res_ptr, out_ptr, ktab, dims, cin, cout, kernel, stride, pad, ldi, ldo, ldr, relu, out_dims, out_rows, wfrag)
orig_stem = avtex.ops.stem_conv
def spy_stem(x_ptr, wt, bias, out_ptr, batch, t, h, pw, cout, kt, st, pt, relu=True):
    shapes.append("stem(LDS patch) cout%d kt%d st%d in(%d, %d, %d, %d)" % (cout, kt, st, batch, t, h, pw))
    return orig_stem(x_ptr, wt, bias, out_ptr, batch, t, h, pw, cout, kt, st, pt, relu)
avtex.ops.stem_conv = spy_stem
orig_stemp = avtex.ops.stem_conv_pool
def spy_stemp(x_ptr, wt, bias, out_ptr, batch, t, h, pw, cout, kt, st, pt, tgroup, ldo):
    shapes.append("stem+pool (LDS patch) cout%d kt%d st%d in(%d, %d, %d, %d)" % (cout, kt, st, batch, t, h, pw))
    return orig_stemp(x_ptr, wt, bias, out_ptr, batch, t, h, pw, cout, kt, st, pt, tgroup, ldo)
avtex.ops.stem_conv_pool = spy_stemp
pool_recs = []
orig_pool = avtex.ops.maxpool_hw3s2
def spy_pool(x_ptr, out_ptr, bt, h, w, c, ldi, ldo, tgroup=1):
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); r = orig_pool(x_ptr, out_ptr, bt, h, w, c, ldi, ldo, tgroup); e.record()
    pool_recs.append((a, e, "maxpool 3x3/2 in(%d, %d, %d, %d)" % (bt, h, w, c), 2.0 * bt * c * (h * w + (h // 2) * (w // 2))))
    return r
avtex.ops.maxpool_hw3s2 = spy_pool
orig_bn = avtex.ops.bottleneck_fused
def spy_bn(x_ptr, out_ptr, packed, batch, t, h, w, c, tchunk=8):
    shapes.append("fused bottleneck C%d in(%d, %d, %d, %d) tchunk %d" % (c, batch, t, h, w, tchunk))
    return orig_bn(x_ptr, out_ptr, packed, batch, t, h, w, c, tchunk)
avtex.ops.bottleneck_fused = spy_bn
orig_bf = avtex.ops.bottleneck_first
def spy_bf(x_ptr, out_ptr, packed, batch, t, h, w, cin, c, tchunk=8):
    shapes.append("fused first block %d->%d in(%d, %d, %d, %d) tchunk %d" % (cin, c, batch, t, h, w, tchunk))
    return orig_bf(x_ptr, out_ptr, packed, batch, t, h, w, cin, c, tchunk)
avtex.ops.bottleneck_first = spy_bf
orig_c33 = avtex.ops.conv33_c64
def spy_c33(x_ptr, wb, bias, out_ptr, batch, t, h, w, ldo, relu=True):
    shapes.append("strip-resident 3x3 64->64 in(%d, %d, %d, %d)" % (batch, t, h, w))
    return orig_c33(x_ptr, wb, bias, out_ptr, batch, t, h, w, ldo, relu)
avtex.ops.conv33_c64 = spy_c33
orig_pw = avtex.ops.pw_chain
def spy_pw(x1_ptr, ldx, k1, w1, b1, res_ptr, ldr, y_ptr, ldy, n1, w2, b2, z_ptr, ldz, n2, m_, x2_ptr=0, ldx2=0, k2x=0):
    shapes.append("pointwise chain %d->%d%s%s ->%d rows %d" % (k1, n1, " +res" if res_ptr else "", " |%d" % k2x if k2x else "", n2, m_))
    return orig_pw(x1_ptr, ldx, k1, w1, b1, res_ptr, ldr, y_ptr, ldy, n1, w2, b2, z_ptr, ldz, n2, m_, x2_ptr, ldx2, k2x)
avtex.ops.pw_chain = spy_pw
avtex.ops.conv3d_igemm = spy
fsf.ops.conv3d_igemm = spy
fsf.PROFILER = hook
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); y = m.forward_ndhwc4(slow, fast); e1.record(); torch.cuda.synchronize()
tot = e0.elapsed_time(e1)
agg = collections.OrderedDict()
for (a, e, fl, byt), name in zip(recs, shapes):
    t = a.elapsed_time(e)
    d = agg.setdefault(name, [0, 0.0, 0.0, 0.0]); d[0] += 1; d[1] += t; d[2] += fl; d[3] += byt
conv_ms = sum(v[1] for v in agg.values())
print("batch %d forward %.2f ms; conv launches %.2f ms (%d launches); pools+head+glue %.2f ms" % (b, tot, conv_ms, len(recs), tot - conv_ms))
for name, (n, t, fl, byt) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%6.3f ms x%d  %-62s %7.1f TF/s %7.0f GB/s" % (t, n, name, fl / t / 1e9, byt / t / 1e6))
for a, e, name, byt in pool_recs:
    t = a.elapsed_time(e)
    print("%6.3f ms     %-62s %7s      %7.0f GB/s" % (t, name, "", byt / t / 1e6))
