"""SlowFast-8x8-R50 inference on the hand-written MFMA convolution (csrc/conv_igemm.hip).

Takes a `slowfast.SlowFast` module (the plugin the reference would get from ModelBuilder3D, models.py:565-580),
folds every BatchNorm into its convolution, repacks the weights [Cout, taps*Cin] in bf16, and runs the residual
stages and lateral fusions as implicit-GEMM launches on NDHWC (channels-last-3d) bf16 activations:
conv + BN + ReLU (+ residual add) is ONE kernel, and the fusion concat is a channel-slice write, so each
activation tensor crosses HBM once per consumer.  The two stem convolutions (Cin = 3, 49 / 245 taps) and their
max-pools still go through MIOpen (6.6 % of the FLOPs).  Same contract as the module it wraps:
    forward([slow [B,3,8,H,W], fast [B,3,32,H,W]]) -> [B, 2304] (fp32).
"""
import torch
import torch.nn as nn

from . import ops
from ._lib import AvtError


class Act:
    """A [M, C] channel slice of a row-major bf16 buffer [M, ld] holding NDHWC activations of extent dims."""

    __slots__ = ("buf", "dims", "c0", "C")

    def __init__(self, buf, dims, c0=0, C=None):
        self.buf, self.dims, self.c0 = buf, dims, c0
        self.C = buf.shape[1] - c0 if C is None else C

    @property
    def ptr(self):
        return self.buf.data_ptr() + 2 * self.c0

    @property
    def ld(self):
        return self.buf.shape[1]


class FusedConv:
    def __init__(self, conv, bn, relu, device):
        w = conv.weight.detach().float()
        cout = w.shape[0]
        if bn is not None:
            scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
            bias = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
            w = w * scale.view(-1, 1, 1, 1, 1)
        else:
            bias = conv.bias.detach().float() if conv.bias is not None else torch.zeros(cout)
        self.kernel, self.stride, self.pad = tuple(conv.kernel_size), tuple(conv.stride), tuple(conv.padding)
        self.cin, self.cout, self.relu = w.shape[1], cout, relu
        if self.cin % 8 or cout % 8:
            raise AvtError("FusedConv: channels must be multiples of 8 (got %d -> %d)" % (self.cin, cout))
        self.wt = w.permute(0, 2, 3, 4, 1).reshape(cout, -1).to(torch.bfloat16).contiguous().to(device)
        self.bias = bias.contiguous().to(device)
        self.dev = device
        self._tabs = {}

    def out_dims(self, dims):
        b, t, h, w = dims
        o = [(n + 2 * p - k) // s + 1 for n, p, k, s in zip((t, h, w), self.pad, self.kernel, self.stride)]
        return (b, o[0], o[1], o[2])

    def __call__(self, x, out=None, res=None, relu=None):
        if x.C != self.cin:
            raise AvtError("FusedConv: input has %d channels, conv expects %d" % (x.C, self.cin))
        key = (x.dims[2], x.dims[3], x.ld)
        tab = self._tabs.get(key)
        if tab is None:
            tab = torch.from_numpy(ops.conv3d_ktab(self.cin, self.kernel, x.dims[2], x.dims[3], x.ld)).to(self.dev)
            self._tabs[key] = tab
        od = self.out_dims(x.dims)
        if out is None:
            m = od[0] * od[1] * od[2] * od[3]
            out = Act(torch.empty((m, self.cout), dtype=torch.bfloat16, device=self.dev), od)
        ops.conv3d_igemm(x.ptr, self.wt, self.bias, res.ptr if res is not None else 0, out.ptr, tab, x.dims, self.cin,
                         self.cout, self.kernel, self.stride, self.pad, x.ld, out.ld, res.ld if res is not None else 0,
                         self.relu if relu is None else relu)
        return out


class _Block:
    def __init__(self, blk, device):
        self.b1 = FusedConv(blk.branch1, blk.branch1_bn, False, device) if hasattr(blk, "branch1") else None
        t = blk.branch2
        self.a = FusedConv(t.a, t.a_bn, True, device)
        self.b = FusedConv(t.b, t.b_bn, True, device)
        self.c = FusedConv(t.c, t.c_bn, True, device)  # ReLU applied after the residual add (fused)

    def __call__(self, x, out=None):
        sc = self.b1(x) if self.b1 is not None else x
        return self.c(self.b(self.a(x)), out=out, res=sc, relu=True)


class SlowFastMFMA(nn.Module):
    """Drop-in for a `SlowFast` module at inference time (eval-mode BatchNorm statistics)."""

    out_dim = 2304

    def __init__(self, model, device, stem_dtype=torch.bfloat16):
        super().__init__()
        self.dev = torch.device(device)
        model = model.eval()
        self.stem = model.s1.to(self.dev, stem_dtype)  # MIOpen (Cin = 3)
        self.stem_dtype = stem_dtype
        self.fuse = [FusedConv(f.conv_f2s, f.bn, True, self.dev) for f in (model.s1_fuse, model.s2_fuse, model.s3_fuse,
                                                                            model.s4_fuse)]
        self.stages = []
        for s in (model.s2, model.s3, model.s4, model.s5):
            self.stages.append([[_Block(getattr(s, "pathway%d_res%d" % (p, i)), self.dev) for i in range(s.depth)]
                                for p in range(2)])

    def parameters(self, recurse=True):  # so callers can read device / dtype like from any nn.Module
        return self.stem.parameters(recurse)

    @torch.no_grad()
    def forward(self, x):
        slow, fast = x
        b = slow.shape[0]
        ys, yf = self.stem([slow.to(self.dev, self.stem_dtype), fast.to(self.dev, self.stem_dtype)])
        cf = yf.shape[1]
        ds, df = (b,) + tuple(ys.shape[2:]), (b,) + tuple(yf.shape[2:])
        ms, mf = ds[0] * ds[1] * ds[2] * ds[3], df[0] * df[1] * df[2] * df[3]
        # NCDHW stem outputs -> NDHWC rows; the slow rows live in the concat buffer of the first lateral fusion
        f_act = Act(yf.permute(0, 2, 3, 4, 1).contiguous().view(mf, cf), df)
        cs = ys.shape[1]
        sbuf = torch.empty((ms, cs + 2 * cf), dtype=torch.bfloat16, device=self.dev)
        sbuf.view(*ds, cs + 2 * cf)[..., :cs].copy_(ys.permute(0, 2, 3, 4, 1))
        self.fuse[0](f_act, out=Act(sbuf, ds, cs, 2 * cf))
        s_act = Act(sbuf, ds)
        for k, (slow_blocks, fast_blocks) in enumerate(self.stages):
            for blk in fast_blocks:
                f_act = blk(f_act)
            last = k == len(self.stages) - 1
            for i, blk in enumerate(slow_blocks):
                if i == len(slow_blocks) - 1 and not last:
                    # last slow block of the stage writes straight into the next fusion's concat buffer
                    od = blk.b.out_dims(blk.a.out_dims(s_act.dims))
                    cs, cf = blk.c.cout, f_act.C
                    sbuf = torch.empty((od[0] * od[1] * od[2] * od[3], cs + 2 * cf), dtype=torch.bfloat16, device=self.dev)
                    blk(s_act, out=Act(sbuf, od, 0, cs))
                    self.fuse[k + 1](f_act, out=Act(sbuf, od, cs, 2 * cf))
                    s_act = Act(sbuf, od)
                else:
                    s_act = blk(s_act)
        # head (models.py:576-580 surgery): global average pool per pathway, concat slow | fast
        hs = s_act.buf.view(b, -1, s_act.buf.shape[1]).float().mean(1)
        hf = f_act.buf.view(b, -1, f_act.buf.shape[1]).float().mean(1)
        return torch.cat([hs, hf], 1)
