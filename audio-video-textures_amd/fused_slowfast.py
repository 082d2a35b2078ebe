"""SlowFast-8x8-R50 inference on the hand-written MFMA convolution (csrc/conv_igemm.hip).

Takes a `slowfast.SlowFast` module (the plugin the reference would get from ModelBuilder3D, models.py:565-580),
folds every BatchNorm into its convolution, repacks the weights [Cout, taps*Cin] in bf16, and runs the residual
stages and lateral fusions as implicit-GEMM launches on NDHWC (channels-last-3d) bf16 activations:
conv + BN + ReLU (+ residual add) is ONE kernel, and the fusion concat is a channel-slice write, so each
activation tensor crosses HBM once per consumer.

The stems (Cin = 3, kernel [kt,7,7], stride [1,2,2]) run on the same kernel through a PIXEL-PAIR view: the
channels-last clip [T,H,W,4] (3 channels + a zero, as ops.clip_pack(layout="ndhwc4") writes it) is read as
[T,H,W/2,8] — two pixels x four channels = one 16-byte chunk — which turns the stride-2, 7-wide kernel into a
stride-1, 4-pair-wide one (columns 2wo-4 .. 2wo+3, the first with zero weight).  The stem max-pool is a small
HBM-bound kernel writing straight into the first fusion's concat buffer.  Same contract as the module it wraps:
    forward([slow [B,3,8,H,W], fast [B,3,32,H,W]]) -> [B, 2304] (fp32);
`forward_ndhwc4` takes the channels-last clips directly (no layout copy).
"""
import torch
import torch.nn as nn

from . import ops
from ._lib import AvtError


# Which layers run on their fused / specialised kernels.  Plain module constants (no environment switches since round 4): every
# one of them names the shipped path; tests and probes set some of them to 0 to get the per-layer general-tile path, which is
# also the fallback for shapes a specialised kernel does not cover.
_FUSE_BLOCK = 1      # bf16: fast-pathway bottlenecks as one kernel (csrc/bottleneck_fused.hip)
_C33 = 1             # bf16: slow res2 b conv on the strip-resident kernel (csrc/conv33_c64.hip)
_CHAIN = 1           # bf16: slow res2 / res3: c (+ residual) of block i and a of block i + 1 in one pass (csrc/pw_chain.hip)
_CHAIN_MAXN = 512    # widest c to chain
_FUSE_KCAT = 1       # slow res2 first block: shortcut folded into c's GEMM (K = [x | b-output])
_XB = 1              # bf16: long-K layers with fragment-order weights that bypass the LDS (XB tile)
_FUSE_SCAT = 1       # bf16: slow res3-5 first blocks: strided shortcut folded into c's GEMM
_FUSE_TCHUNK = 0     # bf16 fused blocks: frames walked per workgroup (2 halo frames each); 0 = by width: 16 at 56 columns, else 32
_FUSE_BLOCK_X3 = 1   # contract grade: fast-pathway bottlenecks as one kernel (csrc/bneck_x3.hip)
_FUSE_TCHUNK_X3 = 0  # contract grade: frames walked per workgroup; 0 = by width (11 at 14 columns, else 8)
_CHAIN_X3 = 1        # contract grade: slow res2 c (+ residual) -> next a in one pass (pw_chain_x3_kernel)
_WBLK_X3 = 1         # contract grade: K-blocked weight planes for the 256 x 256 tile
_STEM_FM_X3 = 1      # contract grade: frame-major tiles in the time-grouped fast stem
_C33_X3 = 1          # contract grade: slow res2 b conv on the direct-operand kernel (csrc/conv33_x3.hip)
_RES2_X3 = 1         # contract grade (fp16 planes): slow res2 identity bottlenecks as ONE kernel, weights streamed (csrc/res2_x3.hip)
_PW_X3 = 1           # contract grade: pointwise layers on the streaming kernel (csrc/pw_x3.hip)
_STEM_LDS = 1        # both: the stems on the patch-resident kernel (csrc/stem_conv.hip)
_STEM_POOL = 1       # bf16: max-pool fused into the stem kernel: 1 = the slow stem (one frame tap), 2 = both (the MFMA-bound fast
#                      stem loses 4 % to the recomputed ninth row).  The split-plane pooled stem was built in round 3, measured
#                      slower than stem + pool (1.74 ms against 1.01 + 0.72 ms, profiles/r03/probe_stem_pool_x3.log) and removed
# (cin, cout) of pointwise layers that run FASTER on the general 128 x 128 tile than on pw_x3: wide-K, 128-output layers without
# a residual (slow res3's a convs: 2.02 -> 1.80 ms and 1.79 -> 1.69 ms per 166 clips; every other pointwise layer is slower there)
_PW_X3_SKIP = {(512, 128), (320, 128)}
# contract grade: (cin, cout) of the lateral [7,1,1] stride-4 connections that run on the streaming kernel's temporal-tap form
# (avt_lateral_x3).  Measured per 166 clips: 32 -> 64 0.91 -> 0.73 ms; 64 -> 128 (two channel chunks re-read the operand) 0.48 ->
# 0.56 and 8 -> 16 0.35 -> 0.37 are faster on the general tile; tests set _LATERAL_X3 = "all" to cover the three forms
_LATERAL_X3 = {(32, 64)}
_STEM_MERGE = 1      # contract grade, frame tables: the fast stem's frame taps that read one source frame are summed on the host
_KW1_CAP = 32        # pixel grouping of temporal-tap layers stops at this output width (profiles/r01/probe_layers.log)

# Optional launch observer for bench.py: PROFILER(name, launch_fn, flops, bytes) must call launch_fn().
PROFILER = None


class Act:
    """A [M, C] channel slice of a row-major bf16 buffer [M, ld] holding NDHWC activations of extent dims.
    Split-plane ("x3") activations carry a second buffer `lo` of identical geometry (value = buf + lo, include/avt.h)."""

    __slots__ = ("buf", "dims", "c0", "C", "lo")

    def __init__(self, buf, dims, c0=0, C=None, lo=None):
        self.buf, self.dims, self.c0, self.lo = buf, dims, c0, lo
        self.C = buf.shape[1] - c0 if C is None else C

    @property
    def ptr(self):
        return self.buf.data_ptr() + 2 * self.c0

    @property
    def ptrs(self):
        return (self.buf.data_ptr() + 2 * self.c0, self.lo.data_ptr() + 2 * self.c0)

    @property
    def ld(self):
        return self.buf.shape[1]

    def float(self, plane_dtype=None):
        """fp32 [M, C] copy of the slice (plane_dtype = ops.X3_* for split-plane activations)."""
        if self.lo is None:
            return self.buf[:, self.c0 : self.c0 + self.C].float()
        dt = torch.float16 if plane_dtype == ops.X3_F16 else torch.bfloat16
        sl = slice(self.c0, self.c0 + self.C)
        return self.buf.view(dt)[:, sl].float() + self.lo.view(dt)[:, sl].float()


def new_act(m, c, dims, device, x3=False):
    buf = torch.empty((m, c), dtype=torch.bfloat16, device=device)
    return Act(buf, dims, lo=torch.empty_like(buf) if x3 else None)


def split_planes(w, plane_dtype):
    """fp32 tensor -> (hi, lo) 16-bit planes typed bfloat16 (raw bits; fp16 planes are bit-cast), value = hi + lo."""
    w = w.detach().float()
    dt = torch.float16 if plane_dtype == ops.X3_F16 else torch.bfloat16
    hi = w.to(dt)
    lo = (w - hi.float()).to(dt)
    return hi.view(torch.bfloat16).contiguous(), lo.view(torch.bfloat16).contiguous()


def fold_bn(conv, bn):
    """-> (weight [Cout,Cin,kt,kh,kw] fp32 with the BN scale folded in, bias [Cout] fp32)."""
    w = conv.weight.detach().float()
    if bn is not None:
        scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
        bias = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
        w = w * scale.view(-1, 1, 1, 1, 1)
    else:
        bias = conv.bias.detach().float() if conv.bias is not None else torch.zeros(w.shape[0])
    return w, bias


def group_weights_w(w, g):
    """Few-channel layers: treat g adjacent pixels along W as g*C channels (same bytes in NDHWC), so a 16-byte chunk
    and an MFMA tile are full instead of mostly padding, and the GEMM has g times fewer rows to decode and gather.
    w [Cout, Cin, kt, kh, kw] (stride_w 1, pad_w kw//2) -> block-Toeplitz [g*Cout, g*Cin, kt, kh, kw'] over pixel
    GROUPS, kw' = 1 + 2*ceil((kw//2)/g); the MFMA work per pixel does not grow (structured zeros replace padding)."""
    cout, cin, kt, kh, kw = w.shape
    r = kw // 2
    rg = -(-r // g)  # groups of halo on each side
    kwg = 1 + 2 * rg
    wg = torch.zeros((g, cout, g, cin, kt, kh, kwg))  # [po, n, pi, c, dt, dh, dg]
    for po in range(g):
        for dg in range(kwg):
            for pi in range(g):
                dw = g * (dg - rg) + pi - po + r
                if 0 <= dw < kw:
                    wg[po, :, pi, :, :, :, dg] = w[:, :, :, :, dw]
    return wg.reshape(g * cout, g * cin, kt, kh, kwg), rg


def pack_wfrag(wt, cin, taps):
    """[Cout, taps*cin] (tap-major K, cin innermost) -> the XB tile's fragment order (include/avt.h): [tiles of 32 rows,
    nup units, 2 k-slices, 64 lanes, 8]; unit i = tap i % taps of channel chunk i // taps (single tap: chunk i)."""
    wt = wt.detach().float().cpu()
    cout, k = wt.shape
    nu = -(-k // 32)
    nup = -(-nu // 4) * 4 + 4
    tiles = -(-cout // 256) * 8
    wp = torch.zeros((tiles * 32, k + 1))  # column k = the zero that masked positions read
    wp[:cout, :k] = wt
    i = torch.arange(nup).view(-1, 1, 1, 1)
    cc, tap = (i // taps, i % taps) if taps > 1 else (i, torch.zeros_like(i))
    ks = torch.arange(2).view(1, -1, 1, 1)
    lane = torch.arange(64).view(1, 1, -1, 1)
    e = torch.arange(8).view(1, 1, 1, -1)
    cch = cc * 32 + ks * 16 + (lane >> 5) * 8 + e  # input channel
    kidx = torch.where((cch < cin) & (i < nu), tap * cin + cch, torch.full_like(cch, k)).expand(nup, 2, 64, 8)
    rows = (torch.arange(tiles).view(-1, 1, 1, 1, 1) * 32 + (lane & 31).view(1, 1, 1, 64, 1)).expand(tiles, nup, 2, 64, 8)
    return wp[rows, kidx.unsqueeze(0).expand(tiles, nup, 2, 64, 8)].to(torch.bfloat16).contiguous()


class FusedConv:
    def __init__(self, conv, bn, relu, device, packed=None, folded=None, x3=None):
        """packed = (wt [Cout, taps*Cin] fp32, bias, cin, kernel, stride, pad, crop) overrides the module;
        folded = (w [Cout,Cin,kt,kh,kw] fp32, bias, stride, pad) is a BN-folded weight in conv layout;
        x3 = None (bf16) | ops.X3_BF16 | ops.X3_F16: contract-grade split-plane arithmetic (csrc/conv_x3.hip)."""
        self.crop = (0, 0, 0)
        self.x3 = x3
        self._folded = None
        self._grouped = {}
        if packed is None:
            if folded is None:
                w, bias = fold_bn(conv, bn)
                self.stride, self.pad = tuple(conv.stride), tuple(conv.padding)
            else:
                w, bias, self.stride, self.pad = folded
            self.kernel = tuple(w.shape[2:])
            self.cin, cout = w.shape[1], w.shape[0]
            wt = w.permute(0, 2, 3, 4, 1).reshape(cout, -1)
            self._folded = (w, bias)
        else:
            wt, bias, self.cin, self.kernel, self.stride, self.pad, self.crop = packed
            cout = wt.shape[0]
        self.cout, self.relu = cout, relu
        if relu == 2 and x3 is None:
            raise AvtError("FusedConv: LeakyReLU (relu=2) exists in the split-plane kernel only")
        if self.cin % 8:
            raise AvtError("FusedConv: input channels must be a multiple of 8 (got %d)" % self.cin)
        if cout % 8:  # pad the output channels with zero filters (the caller's buffer must be that wide)
            raise AvtError("FusedConv: output channels must be a multiple of 8 (got %d)" % cout)
        self.wfrag = self.wt_lo = self.wscale = self.wblk = None
        taps = self.kernel[0] * self.kernel[1] * self.kernel[2]
        if x3 is not None:
            wf = wt.detach().float()
            if x3 == ops.X3_F16:
                # fp16 planes: each output channel's weights are scaled by a power of two into [2^9, 2^10) so that the low
                # plane of every weight that matters is a normal fp16; the kernel multiplies the accumulator back (exact)
                mx = wf.abs().amax(dim=1).clamp_min(1e-30)
                e = torch.floor(torch.log2(mx))
                sc = torch.pow(2.0, 9.0 - e)
                wf = wf * sc.view(-1, 1)
                self.wscale = (1.0 / sc).float().contiguous().to(device)
            hi, lo = split_planes(wf, x3)
            self.wt, self.wt_lo = hi.to(device), lo.to(device)
            # layers the 256 x 256 tile can run: the planes once more in K-blocked order [K / 32, Cout, 32] (a wave's 16 weight
            # rows per staging instruction are 8 whole cache lines instead of 16 half lines; avt_conv3d_igemm_x3_wblk)
            k_all = hi.shape[1]
            if _WBLK_X3 and cout % 256 == 0 and k_all % 32 == 0 and ops.conv3d_igemm_x3_xl_picked(cout, k_all, 1 << 20):
                blk = lambda p: p.view(cout, k_all // 32, 32).permute(1, 0, 2).contiguous().to(device)
                self.wblk = (blk(hi), blk(lo))
            # pointwise stride-1 layers: the streaming kernel (csrc/pw_x3.hip) with LDS-resident weight fragments
            self.pw = None
            if (_PW_X3 and relu != 2 and self.kernel == (1, 1, 1) and self.stride == (1, 1, 1) and self.pad == (0, 0, 0) and
                    not any(self.crop) and cout % 32 == 0 and ops.pw_x3_supported(self.cin, cout) and
                    (self.cin, cout) not in _PW_X3_SKIP):
                self.pw = (pack_pw_planes(hi).to(device), pack_pw_planes(lo).to(device))
            # Conv3d [kt,1,1] with a temporal stride (the lateral fast -> slow connections): the streaming kernel's temporal-tap
            # form — the operand gathered from the kt input frames of a position, no tap table (avt_lateral_x3, round 4)
            self.lat = None
            if ((_LATERAL_X3 == "all" or (self.cin, cout) in _LATERAL_X3) and relu != 2 and self.kernel[0] > 1 and self.kernel[1:] == (1, 1) and self.stride[1:] == (1, 1) and
                    self.pad[1:] == (0, 0) and not any(self.crop) and ops.lateral_x3_supported(self.cin, cout, self.kernel[0])):
                n_pad = -(-cout // 32) * 32
                padr = lambda p: torch.cat([p.cpu(), torch.zeros((n_pad - cout, p.shape[1]), dtype=p.dtype)], 0) if n_pad > cout else p
                padv = lambda v, fill: None if v is None else torch.cat([v.cpu().float(), torch.full((n_pad - cout,), fill)]).contiguous().to(device)
                self.lat = (pack_pw_planes(padr(hi)).to(device), pack_pw_planes(padr(lo)).to(device), padv(bias, 0.0), padv(self.wscale, 1.0))
        else:
            self.wt = wt.to(torch.bfloat16).contiguous().to(device)
            if _XB and ops.conv3d_wfrag_supported(self.cin, cout, self.kernel):
                self.wfrag = pack_wfrag(self.wt.float().cpu(), self.cin, taps).to(device)
        self.bias = bias.float().contiguous().to(device)
        self.dev = device
        self._tabs = {}
        # algorithmic flops per GEMM row (bench.py): the convolution's own 2*K*Cout; pixel-paired / grouped forms,
        # whose packed K carries structured zeros, overwrite it with the flops of the convolution they compute
        self.alg_flops_per_row = 2.0 * self.wt.shape[1] * cout

    def out_dims(self, dims):
        b, t, h, w = dims
        o = [(n + 2 * p - k) // s + 1 - c for n, p, k, s, c in zip((t, h, w), self.pad, self.kernel, self.stride, self.crop)]
        return (b, o[0], o[1], o[2])

    def kernel_symbol(self, m_out):
        """The device kernel the dispatcher of csrc/conv_igemm.hip / conv_x3.hip picks for this layer (bench.py names its
        roofline rows by it; mirrors avt_conv3d_igemm_wfrag_bf16 / avt_conv3d_igemm_x3)."""
        if self.x3 is not None:
            if getattr(self, "pw", None) is not None or getattr(self, "lat", None) is not None:
                return "pw_x3_kernel<%s>" % ("f16" if self.x3 == ops.X3_F16 else "bf16")
            if ops.conv3d_igemm_x3_xl_picked(self.cout, self.wt.shape[1], m_out):
                return "conv_x3_xl_kernel<%s>" % ("f16" if self.x3 == ops.X3_F16 else "bf16")
            tile = "128,32,32" if self.cout <= 32 else ("128,64,64" if self.cout <= 64 else "128,128,64")
            return "conv_x3_kernel<%s,%s>" % (tile, "f16" if self.x3 == ops.X3_F16 else "bf16")
        if self.wfrag is not None:
            return "conv_xb_kernel"
        taps = self.kernel[0] * self.kernel[1] * self.kernel[2]
        nk = -(-self.wt.shape[1] // 64)
        if self.cout >= 256 and nk >= 16 and (self.cin % 32 == 0 or taps == 1):
            return "conv_xl_kernel<256,256,2>"
        return "conv_igemm_kernel<%s>" % ("256,32,64" if self.cout <= 32 else ("256,64,64" if self.cout <= 64 else "128,128,64"))

    def group_factor(self, x, out, res):
        """Pixel-group factor along W for few-channel layers (see group_weights_w); 1 = plain."""
        if self._folded is None or self.stride[2] != 1 or self.pad[2] != self.kernel[2] // 2:
            return 1
        if getattr(self, "lat", None) is not None and res is None:
            return 1  # the temporal-tap streaming kernel takes the layer as it is
        if x.ld != x.C or x.c0 or (out is not None and (out.ld != out.C or out.c0)) or \
                (res is not None and (res.ld != res.C or res.c0)):
            return 1  # channel slices of wider rows cannot be re-viewed
        small = min(self.cin, self.cout)
        g = 4 if small <= 16 else (2 if small <= 32 else 1)
        if self.kernel[2] == 1 and self.kernel[0] > 1 and _KW1_CAP:
            # temporal taps only: grouping just fills the tile (block-diagonal weights), so stop at a 32-wide output
            # (pointwise layers stay grouped: they are HBM-bound and gain from 4x fewer, fuller rows)
            while g > 1 and g * self.cout > _KW1_CAP:
                g //= 2
        while g > 1 and x.dims[3] % g:
            g //= 2
        return g

    def __call__(self, x, out=None, res=None, relu=None, out_rows=None):
        """out_rows = (stride, H, W): write output position (f, ho, wo) to row (f*H + stride*ho)*W + stride*wo of `out`."""
        if x.C != self.cin:
            raise AvtError("FusedConv: input has %d channels, conv expects %d" % (x.C, self.cin))
        g = 1 if out_rows is not None else self.group_factor(x, out, res)
        if g > 1:
            sub = self._grouped.get(g)
            if sub is None:
                w, bias = self._folded
                wg, rg = group_weights_w(w, g)
                sub = FusedConv(None, None, self.relu, self.dev,
                                folded=(wg, bias.repeat(g), self.stride, (self.pad[0], self.pad[1], rg)), x3=self.x3)
                sub._folded = None  # never re-group
                sub.alg_flops_per_row = g * self.alg_flops_per_row
                self._grouped[g] = sub
            b, t, h, w_ = x.dims
            od = self.out_dims(x.dims)
            m_out = od[0] * od[1] * od[2] * od[3]
            if out is None:
                out = new_act(m_out, self.cout, od, self.dev, self.x3 is not None)
            view = lambda a, d, c: Act(a.buf.view(-1, g * c), (d[0], d[1], d[2], d[3] // g),
                                       lo=a.lo.view(-1, g * c) if a.lo is not None else None)
            sub(view(x, x.dims, self.cin), out=view(out, od, self.cout),
                res=view(res, od, self.cout) if res is not None else None, relu=relu)
            return out
        key = (x.dims[2], x.dims[3], x.ld)
        tab = self._tabs.get(key)
        if tab is None:
            tab = torch.from_numpy(ops.conv3d_ktab(self.cin, self.kernel, x.dims[2], x.dims[3], x.ld)).to(self.dev)
            self._tabs[key] = tab
        od = self.out_dims(x.dims)
        if out is None:
            m = od[0] * od[1] * od[2] * od[3]
            out = new_act(m, self.cout, od, self.dev, self.x3 is not None)

        def launch_x3():
            m_rows = od[0] * od[1] * od[2] * od[3]
            blocked = self.wblk is not None and ops.conv3d_igemm_x3_xl_picked(self.cout, self.wt.shape[1], m_rows)
            wh, wl = self.wblk if blocked else (self.wt, self.wt_lo)
            ops.conv3d_igemm_x3(x.ptrs, wh, wl, self.bias, res.ptrs if res is not None else None, out.ptrs, tab,
                                x.dims, self.cin, self.cout, self.kernel, self.stride, self.pad, x.ld, out.ld,
                                res.ld if res is not None else 0, self.relu if relu is None else relu, self.x3,
                                wscale=self.wscale, out_dims=od[1:] if any(self.crop) else (0, 0, 0), out_rows=out_rows,
                                wblk=blocked)

        def launch_pw():
            m_rows = x.dims[0] * x.dims[1] * x.dims[2] * x.dims[3]
            ops.pw_x3(x.ptrs, x.ld, self.cin, self.pw[0], self.pw[1], self.bias, self.wscale,
                      res.ptrs if res is not None else None, res.ld if res is not None else 0, out.ptrs, out.ld, self.cout,
                      m_rows, self.relu if relu is None else relu, self.x3)

        def launch_lat():
            b_, t_, h_, w_ = x.dims
            ops.lateral_x3(x.ptrs, x.ld, self.cin, self.lat[0], self.lat[1], self.lat[2], self.lat[3], out.ptrs, out.ld, self.cout, b_, t_,
                           h_ * w_, self.kernel[0], self.stride[0], self.pad[0], self.relu if relu is None else relu, self.x3)

        def launch():
            if self.x3 is not None:
                if getattr(self, "lat", None) is not None and out_rows is None and res is None:
                    return launch_lat()
                return launch_pw() if (self.pw is not None and out_rows is None) else launch_x3()
            ops.conv3d_igemm(x.ptr, self.wt, self.bias, res.ptr if res is not None else 0, out.ptr, tab, x.dims,
                             self.cin, self.cout, self.kernel, self.stride, self.pad, x.ld, out.ld,
                             res.ld if res is not None else 0, self.relu if relu is None else relu,
                             out_dims=od[1:] if any(self.crop) else (0, 0, 0), out_rows=out_rows, wfrag=self.wfrag)

        if PROFILER is None:
            launch()
        else:
            m_out = od[0] * od[1] * od[2] * od[3]
            m_in = x.dims[0] * x.dims[1] * x.dims[2] * x.dims[3]
            nb = 2.0 * (m_in * self.cin + m_out * self.cout * (2 if res is not None else 1)) + self.wt.numel() * 2
            PROFILER(self.kernel_symbol(m_out), launch, m_out * self.alg_flops_per_row,
                     nb * (2 if self.x3 is not None else 1))
        return out


def stem_lds_image(wt, kt, frame_major=False):
    """Packed stem weights [Cout, kt*7*4*8] -> the LDS image order of csrc/stem_conv.hip (include/avt.h):
    [Cout/32, kt, 7 dh, 2 tiles, 4 dp, 16 rows, 8], channel = 32*group + 8*(row//4) + 4*tile + row%4 — or, frame_major (the
    time-grouped fast stem in the split-plane kernels), channel = 32*group + 16*tile + row: a tile is two output FRAMES and
    the kernel skips the frame taps a tile never meets."""
    cout = wt.shape[0]
    if frame_major:
        w = wt.reshape(cout // 32, 2, 4, 4, kt, 7, 4, 8)  # [G, tile, q, i, dt, dh, dp, 8]
        return w.permute(0, 4, 5, 1, 6, 2, 3, 7).contiguous().reshape(cout // 32, -1)
    w = wt.reshape(cout // 32, 4, 2, 4, kt, 7, 4, 8)  # [G, q, tile, i, dt, dh, dp, 8]
    return w.permute(0, 4, 5, 2, 6, 1, 3, 7).contiguous().reshape(cout // 32, -1)


def merged_stem_taps(conv, win_len, device):
    """The time-grouped fast stem for clips whose 32 slots are sampled from a `win_len`-frame window (linspace(0, W-1, 32)
    .long(): W < 32 repeats frames).  Output-frame group `to` (frames 4 to .. 4 to + 3) meets slots 4 to - 2 .. 4 to + 5; the
    slots that read ONE source frame have their weight slabs summed (a convolution is linear in its weights), so the group
    walks its DISTINCT source frames only — 5 instead of 8 at W = 20: the MFMAs and the patch staging of the other taps are
    never issued.  -> dict(wt_hi, wt_lo: frame-major LDS images over To * ktm slabs; bias, wscale; src int32 [To, ktm] source-
    frame offset from the window start (-1 = no tap); tiles int32 [To * ktm] tile bit mask; ktm) or None without duplicates."""
    hit = conv._merged.get(win_len)
    if hit is not None or win_len in conv._merged:
        return hit
    wg, bias = conv.wg                      # [g, c, kt_g, 7, 4, 2, 4], [c]
    g, c, kt_g = wg.shape[0], wg.shape[1], wg.shape[2]
    fast_off, _ = ops.clip_sample_table(win_len)
    n_slots, pt = len(fast_off), conv.pad[0]
    n_grp = (n_slots + 2 * pt - kt_g) // g + 1
    groups = []
    for to in range(n_grp):
        by_src = {}
        for dt in range(kt_g):
            k = g * to - pt + dt
            if 0 <= k < n_slots:
                by_src.setdefault(int(fast_off[k]), []).append(dt)
        groups.append(sorted(by_src.items()))
    ktm = max(len(x) for x in groups)
    if ktm >= kt_g:  # every slot of some group is its own frame: nothing to merge
        conv._merged[win_len] = None
        return None
    kt0 = kt_g - g + 1                       # tile n (output frames 2n, 2n + 1) meets taps 2n .. 2n + kt0 of the unmerged image
    wflat = wg.reshape(g * c, kt_g, -1)
    wm = torch.zeros((g * c, n_grp * ktm, wflat.shape[2]))
    src = torch.full((n_grp, ktm), -1, dtype=torch.int32)
    tiles = torch.zeros((n_grp * ktm,), dtype=torch.int32)
    for to, items in enumerate(groups):
        for j, (off, dts) in enumerate(items):
            wm[:, to * ktm + j] = sum(wflat[:, dt] for dt in dts)
            src[to, j] = off
            tiles[to * ktm + j] = sum(1 << n for n in range(2) if any(2 * n <= dt <= 2 * n + kt0 for dt in dts))
    m = FusedConv(None, None, True, device,
                  packed=(wm.reshape(g * c, -1), bias.repeat(g), 8, (n_grp * ktm, conv.kernel[1], 4), conv.stride, conv.pad, (0, 0, 1)),
                  x3=conv.x3)
    out = {"wt_hi": stem_lds_image(m.wt, n_grp * ktm, True), "wt_lo": stem_lds_image(m.wt_lo, n_grp * ktm, True), "bias": m.bias,
           "wscale": m.wscale, "src": src.to(device), "tiles": tiles.to(device), "ktm": ktm, "groups": n_grp}
    conv._merged[win_len] = out
    return out


def stem_conv(stem, device, tgroup=1, x3=None):
    """Stem Conv3d(3, C, [kt,7,7], stride [1,2,2], pad [kt//2,3,3]) + BN + ReLU in pixel-pair form (see module doc).

    tgroup = g > 1 (fast stem, C = 8): g consecutive output frames are computed as g*C channels of ONE output row
    (block-Toeplitz weights over kt+g-1 input frames, temporal stride g), so the 32-wide MFMA tile is full instead
    of 3/4 zero padding: (kt+g-1)/(g*kt) of the MFMA work of the plain form (8/20 for kt=5, g=4)."""
    w, bias = fold_bn(stem.conv, stem.bn)  # [C, 3, kt, 7, 7]
    c, _, kt, kh, kw = w.shape
    if (kh, kw) != (7, 7) or tuple(stem.conv.stride) != (1, 2, 2) or tuple(stem.conv.padding) != (kt // 2, 3, 3):
        raise AvtError("stem_conv: expected a [kt,7,7] stride-[1,2,2] stem")
    wp = torch.zeros((c, kt, kh, 4, 2, 4))  # [n, dt, dh, pair, pixel-in-pair, channel(4)]
    for dwp in range(4):
        for p in range(2):
            k = 2 * dwp + p - 1  # original column tap: column = 2wo - 4 + 2*dwp + p = 2wo - 3 + k
            if 0 <= k < kw:
                wp[:, :, :, dwp, p, :3] = w[:, :, :, :, k].permute(0, 2, 3, 1)
    if tgroup > 1:
        g = tgroup
        wg = torch.zeros((g, c, kt + g - 1) + tuple(wp.shape[2:]))
        for j in range(g):
            wg[j, :, j : j + kt] = wp
        conv = FusedConv(None, None, True, device,
                         packed=(wg.reshape(g * c, -1), bias.repeat(g), 8, (kt + g - 1, kh, 4), (g, 2, 1),
                                 (kt // 2, 3, 2), (0, 0, 1)), x3=x3)
        conv.tgroup, conv.frame_channels = g, c
        conv.alg_flops_per_row = g * 2.0 * (kt * kh * kw * 3) * c
        # split-plane kernels: frame-major tiles (4 frames x 8 channels only), whose structurally-zero frame taps are skipped
        conv.frames_per_tile = 2 if (_STEM_FM_X3 and x3 is not None and g == 4 and c == 8) else 0
        fm = conv.frames_per_tile > 0
        conv.wt_lds = stem_lds_image(conv.wt, kt + g - 1, fm) if conv.cout % 32 == 0 else None
        conv.wt_lds_lo = stem_lds_image(conv.wt_lo, kt + g - 1, fm) if conv.cout % 32 == 0 and x3 is not None else None
        conv.wg, conv._merged = (wg, bias) if fm else None, {}  # (fp32 block-Toeplitz weights: merged_stem_taps sums their taps)
        return conv
    conv = FusedConv(None, None, True, device,
                     packed=(wp.reshape(c, -1), bias, 8, (kt, kh, 4), (1, 2, 1), (kt // 2, 3, 2), (0, 0, 1)), x3=x3)
    conv.tgroup, conv.frame_channels, conv.frames_per_tile = 1, c, 0
    conv.alg_flops_per_row = 2.0 * (kt * kh * kw * 3) * c
    conv.wt_lds = stem_lds_image(conv.wt, kt) if conv.cout % 32 == 0 else None
    conv.wt_lds_lo = stem_lds_image(conv.wt_lo, kt) if conv.cout % 32 == 0 and x3 is not None else None
    return conv


def pack_pw_planes(plane):
    """A split-plane weight plane [N, K] (16-bit raw, typed bfloat16) -> csrc/pw_x3.hip's fragments [N/16][ceil(K/32)][64][8]
    (pack_pw's order: output rows permuted so a lane ends with 8 consecutive channels; K zero-padded to whole k-steps)."""
    raw = plane.detach().cpu().view(torch.int16)
    n_out, k = raw.shape
    ks = -(-k // 32)
    wp = torch.zeros((n_out, ks * 32), dtype=torch.int16)
    wp[:, :k] = raw
    lane = torch.arange(64)
    n, q = lane & 15, lane >> 4
    e = torch.arange(8)
    shape = (n_out // 16, ks, 64, 8)
    nt = torch.arange(n_out // 16).view(-1, 1, 1, 1)
    row = (32 * (nt // 2) + 8 * (n >> 2).view(1, 1, -1, 1) + 4 * (nt % 2) + (n & 3).view(1, 1, -1, 1)).expand(shape)
    col = (torch.arange(ks).view(1, -1, 1, 1) * 32 + q.view(1, 1, -1, 1) * 8 + e.view(1, 1, 1, -1)).expand(shape)
    return wp[row, col].contiguous().view(torch.bfloat16)


def _bottleneck_fragments(wa, wb, wc, wsc=None, sc_kgroup=0):
    """fp32 weights of a [3,1,1] -> [1,3,3] -> [1,1,1] bottleneck (wa [Cm,Cin,3,1,1], wb [Cm,Cm,1,3,3], wc [C,Cm,1,1,1], optional
    shortcut wsc [C,Cin,1,1,1]) -> fp32 tensors in the MFMA-fragment order of csrc/bottleneck_fused.hip / bneck_x3.hip
    (include/avt.h): (wa_f, wb_f, wc_f, ws_f | None), the bottleneck width zero-padded to CMP = 16 (Cm <= 16; b's taps packed
    in pairs) or 32 (Cm = 32; one tap per MFMA k-step).  sc_kgroup: k-group of the 8-channel first block's shortcut weights
    (0: the operand is x(t) in every group; 1: the operand holds frames t-1, t, t+1 in groups 0, 1, 2 — bneck_x3)."""
    cm, c, cin = wa.shape[0], wc.shape[0], wa.shape[1]
    cmp_ = 16 if cm <= 16 else 32
    first = cin == 8  # res2's first block: 8 input channels, the three frame taps share one k-step (see include/avt.h)
    nta = cmp_ // 16
    lane = torch.arange(64)
    n, q = lane & 15, lane >> 4
    e = torch.arange(8)
    L, E = n.numel(), e.numel()
    if first:
        # one k-step holds the three frame taps: k-group q = dt (q = 3: zeros), 8 channels each
        wa_p = torch.zeros((cmp_, 8, 4))
        wa_p[:cm, :, :3] = wa[:, :, :, 0, 0]
        wa_f = wa_p[n.view(-1, 1).expand(L, E), e.view(1, -1).expand(L, E), q.view(-1, 1).expand(L, E)].view(1, 1, 1, L, E)
    else:
        wa_p = torch.zeros((cmp_, cin, 3))
        wa_p[:cm] = wa[:, :, :, 0, 0]
        ka = cin // 32
        # wa fragments [3 dt][ka][nta][64][8]: row nt*16 + n, k = 32*k + 8*q + e
        k_idx = (torch.arange(ka).view(-1, 1, 1, 1) * 32 + q.view(1, 1, -1, 1) * 8 + e.view(1, 1, 1, -1)).expand(ka, nta, L, E)
        r_idx = (torch.arange(nta).view(1, -1, 1, 1) * 16 + n.view(1, 1, -1, 1)).expand(ka, nta, L, E)
        wa_f = torch.stack([wa_p[r_idx, k_idx, dt] for dt in range(3)])
    # wb fragments [nb][nta][64][8]
    wb_p = torch.zeros((cmp_, cmp_, 10))  # [n, ch, tap]; tap 9 = zeros
    wb_p[:cm, :cm, :9] = wb[:, :, 0].reshape(cm, cm, 9)
    nb = 5 if cmp_ == 16 else 9
    j = torch.arange(nb).view(-1, 1, 1, 1)
    if cmp_ == 16:
        tap = (2 * j + (q >> 1).view(1, 1, -1, 1)).expand(nb, nta, L, E)
        ch = (8 * (q & 1).view(1, 1, -1, 1) + e.view(1, 1, 1, -1)).expand(nb, nta, L, E)
    else:
        tap = j.expand(nb, nta, L, E)
        ch = (8 * q.view(1, 1, -1, 1) + e.view(1, 1, 1, -1)).expand(nb, nta, L, E)
    rb = (torch.arange(nta).view(1, -1, 1, 1) * 16 + n.view(1, 1, -1, 1)).expand(nb, nta, L, E)
    wb_f = wb_p[rb, ch, tap]
    # wc fragments [c/16][64][8]: tile nt, row r -> channel 32*(nt//2) + 8*(r//4) + 4*(nt%2) + r%4, k = 8*q + e
    wc_p = torch.zeros((c, 32))
    wc_p[:, :cm] = wc[:, :, 0, 0, 0]
    nt = torch.arange(c // 16).view(-1, 1, 1)
    chan = (32 * (nt // 2) + 8 * (n >> 2).view(1, -1, 1) + 4 * (nt % 2) + (n & 3).view(1, -1, 1)).expand(c // 16, L, E)
    kk = (q.view(1, -1, 1) * 8 + e.view(1, 1, -1)).expand(c // 16, L, E)
    wc_f = wc_p[chan, kk]
    if wsc is None:
        return wa_f, wb_f, wc_f, None
    # shortcut conv fragments [c/16][ks][64][8] in c's row order, k = 32*ks + 8*q + e over the input channels
    ks = max(cin // 32, 1)
    ws_p = torch.zeros((c, ks * 32))
    ws_p[:, 8 * sc_kgroup : 8 * sc_kgroup + cin] = wsc[:, :, 0, 0, 0]
    chan4 = chan.view(c // 16, 1, L, E).expand(c // 16, ks, L, E)
    kk4 = (torch.arange(ks).view(1, -1, 1, 1) * 32 + kk.view(c // 16, 1, L, E)).expand(c // 16, ks, L, E)
    return wa_f, wb_f, wc_f, ws_p[chan4, kk4]


def pack_bottleneck(wa, ba, wb, bb, wc, bc, device, shortcut=None):
    """BN-folded weights of a [3,1,1] -> [1,3,3] -> [1,1,1] bottleneck (wa [Cm,C,3,1,1], wb [Cm,Cm,1,3,3], wc [C,Cm,1,1,1])
    -> the MFMA-fragment order of csrc/bottleneck_fused.hip (include/avt.h), bf16."""
    wa, ba, wb, bb, wc, bc = [v.detach().float().cpu() for v in (wa, ba, wb, bb, wc, bc)]  # packing is host work
    cmp_ = 16 if wa.shape[0] <= 16 else 32
    wsc = bsc = None
    if shortcut is not None:
        wsc, bsc = [v.detach().float().cpu() for v in shortcut]  # [C, Cin, 1, 1, 1], [C]
    wa_f, wb_f, wc_f, ws_f = _bottleneck_fragments(wa, wb, wc, wsc)
    padw = lambda v: torch.cat([v.float(), torch.zeros(cmp_ - v.numel())])
    dev = lambda v, dt: v.to(dt).contiguous().to(device)
    out = (dev(wa_f, torch.bfloat16), dev(padw(ba), torch.float32), dev(wb_f, torch.bfloat16), dev(padw(bb), torch.float32),
           dev(wc_f, torch.bfloat16))
    if shortcut is None:
        return out + (dev(bc.float(), torch.float32),)
    return out + (dev((bc + bsc).float(), torch.float32), dev(ws_f, torch.bfloat16))


def pack_bottleneck_x3(wa, ba, wb, bb, wc, bc, x3, device, shortcut=None):
    """The same bottleneck for csrc/bneck_x3.hip (contract-grade split-plane arithmetic): -> (wfrag, coef).
    wfrag [NF][2 planes][64][8] 16-bit (typed bfloat16): the fragments of a, b, c (and the 8-channel first block's shortcut
    conv, weights at k-group 1 = frame t of the operand), each as its hi and its lo plane; coef fp32 [sa | ba | sb | bb |
    sc | bc]: with fp16 planes every output channel's weights are scaled by a power of two into [2^9, 2^10) (low plane
    normal, FusedConv's rule) and s* multiplies the accumulator back — exactly; c and its shortcut share one scale per
    channel because they share the accumulator.  bf16 planes: scales 1."""
    wa, ba, wb, bb, wc, bc = [v.detach().float().cpu() for v in (wa, ba, wb, bb, wc, bc)]
    cm, c = wa.shape[0], wc.shape[0]
    cmp_ = 16 if cm <= 16 else 32
    wsc = bsc = None
    if shortcut is not None:
        wsc, bsc = [v.detach().float().cpu() for v in shortcut]

    def scale_of(*ws):  # per output channel, over every weight that accumulates into it
        if x3 != ops.X3_F16:
            return torch.ones(ws[0].shape[0])
        mx = torch.stack([w.reshape(w.shape[0], -1).abs().amax(dim=1) for w in ws]).amax(dim=0).clamp_min(1e-30)
        return torch.pow(2.0, 9.0 - torch.floor(torch.log2(mx)))

    sa, sb = scale_of(wa), scale_of(wb)
    sc = scale_of(wc, wsc) if wsc is not None else scale_of(wc)
    v5 = lambda s_: s_.view(-1, 1, 1, 1, 1)
    wa_f, wb_f, wc_f, ws_f = _bottleneck_fragments(wa * v5(sa), wb * v5(sb), wc * v5(sc),
                                                    None if wsc is None else wsc * v5(sc),
                                                    sc_kgroup=1 if wa.shape[1] == 8 else 0)
    frags = [wa_f.reshape(-1, 64, 8), wb_f.reshape(-1, 64, 8), wc_f.reshape(-1, 64, 8)]
    if ws_f is not None:
        frags.append(ws_f.reshape(-1, 64, 8))
    allf = torch.cat(frags, 0)
    hi, lo = split_planes(allf, x3)
    wfrag = torch.stack([hi, lo], 1).contiguous().to(device)  # [NF][2][64][8]
    padw = lambda v, fill: torch.cat([v.float(), torch.full((cmp_ - v.numel(),), fill)])
    coef = torch.cat([padw(1.0 / sa, 1.0), padw(ba, 0.0), padw(1.0 / sb, 1.0), padw(bb, 0.0), 1.0 / sc,
                      bc if bsc is None else bc + bsc]).float().contiguous().to(device)
    return wfrag, coef


def pack_c33(wb, device):
    """BN-folded [64,64,1,3,3] weights -> csrc/conv33_c64.hip's fragment order (output rows permuted so a lane ends
    with 8 consecutive channels, include/avt.h)."""
    wb = wb.detach().float().cpu()
    cm = wb.shape[0]
    lane = torch.arange(64)
    n, q = lane & 15, lane >> 4
    e = torch.arange(8)
    wb9 = wb[:, :, 0].reshape(cm, cm, 9)  # [out, in, tap]
    shape = (9, 2, cm // 16, 64, 8)
    tap = torch.arange(9).view(-1, 1, 1, 1, 1).expand(shape)
    kh = torch.arange(2).view(1, -1, 1, 1, 1)
    nt = torch.arange(cm // 16).view(1, 1, -1, 1, 1)
    row = (32 * (nt // 2) + 8 * (n >> 2).view(1, 1, 1, -1, 1) + 4 * (nt % 2) + (n & 3).view(1, 1, 1, -1, 1)).expand(shape)
    ch = (kh * 32 + q.view(1, 1, 1, -1, 1) * 8 + e.view(1, 1, 1, 1, -1)).expand(shape)
    return wb9[row, ch, tap].to(torch.bfloat16).contiguous().to(device)


def pack_c33_x3(wb, bias, x3, device):
    """BN-folded [64,64,1,3,3] weights + bias -> csrc/conv33_x3.hip's (wfrag, coef) (include/avt.h): 32 x 32 x 16 MFMA fragments
    of both planes, output rows permuted so that a lane of the accumulator ends with runs of 8 consecutive channels; fp16
    planes: every output channel scaled by a power of two into [2^9, 2^10), undone by coef's scale."""
    wb = wb.detach().float().cpu()
    co, ci = wb.shape[0], wb.shape[1]
    w9 = wb[:, :, 0].reshape(co, ci, 9)  # [out, in, tap]
    if x3 == ops.X3_F16:
        mx = w9.reshape(co, -1).abs().amax(dim=1).clamp_min(1e-30)
        sc = torch.pow(2.0, 9.0 - torch.floor(torch.log2(mx)))
    else:
        sc = torch.ones(co)
    w9 = w9 * sc.view(-1, 1, 1)
    lane = torch.arange(64)
    rho, kh = lane & 31, lane >> 5
    h = (rho >> 2) & 1
    r = (rho & 3) + 4 * (rho >> 3)
    shape = (9, 4, 2, 64, 8)  # [tap][k-slice][n-tile][lane][e]
    n = torch.arange(2).view(1, 1, -1, 1, 1)
    ch = (32 * n + ((2 * (r >> 3) + h) * 8 + (r & 7)).view(1, 1, 1, -1, 1)).expand(shape)
    kk = (torch.arange(4).view(1, -1, 1, 1, 1) * 16 + (kh * 8).view(1, 1, 1, -1, 1) + torch.arange(8).view(1, 1, 1, 1, -1)).expand(shape)
    tap = torch.arange(9).view(-1, 1, 1, 1, 1).expand(shape)
    frags = w9[ch, kk, tap]  # fp32 [9][4][2][64][8]
    hi, lo = split_planes(frags, x3)
    wfrag = torch.stack([hi, lo], 3).contiguous().to(device)  # [9][4][2][2 planes][64][8]
    coef = torch.cat([1.0 / sc, bias.detach().float().cpu()]).float().contiguous().to(device)
    return wfrag, coef


def pack_res2_x3(wa, ba, wb, bb, wc, bc, x3, device):
    """BN-folded weights of a slow-res2 identity bottleneck (wa [64,256,1,1,1], wb [64,64,1,3,3], wc [256,64,1,1,1]) -> csrc/res2_x3.hip's
    (wfrag, coef) (include/avt.h avt_res2_x3): the weight STREAM of 17 chunks x 8 pairs x 2 planes of 16 x 16 x 32 MFMA operands in the
    order the kernel consumes them, every operand in pack_pw_planes' form (output rows permuted so that two n-tiles give a lane 8
    consecutive channels); fp16 planes: every output channel scaled by a power of two into [2^9, 2^10), undone by coef's scales."""
    wa, ba, wb, bb, wc, bc = [v.detach().float().cpu() for v in (wa, ba, wb, bb, wc, bc)]
    cm, c = wa.shape[0], wc.shape[0]
    assert (cm, c) == (64, 256) and wa.shape[1] == c and tuple(wb.shape) == (cm, cm, 1, 3, 3) and wc.shape[1] == cm

    def scale_of(w):
        if x3 != ops.X3_F16:
            return torch.ones(w.shape[0])
        mx = w.reshape(w.shape[0], -1).abs().amax(dim=1).clamp_min(1e-30)
        return torch.pow(2.0, 9.0 - torch.floor(torch.log2(mx)))

    sa, sb, sc = scale_of(wa), scale_of(wb), scale_of(wc)
    a2 = wa.reshape(cm, c) * sa.view(-1, 1)                       # [64, 256]
    b9 = wb[:, :, 0].reshape(cm, cm, 9) * sb.view(-1, 1, 1)       # [out, in, tap]
    c2 = wc.reshape(c, cm) * sc.view(-1, 1)                       # [256, 64]

    def frags(mat):  # [N, K] -> (hi, lo) [N/16][K/32][64][8] (raw 16-bit, typed bfloat16)
        hi, lo = split_planes(mat, x3)
        return pack_pw_planes(hi), pack_pw_planes(lo)

    pairs = []  # (hi [64, 8], lo [64, 8]) in stream order
    ah, al = frags(a2)
    for jc in range(4):          # a: chunk jc = k-steps 2 jc, 2 jc + 1; pair (kk, nt) at 4 kk + nt
        for kk in range(2):
            for nt in range(4):
                pairs.append((ah[nt, 2 * jc + kk], al[nt, 2 * jc + kk]))
    for t in range(9):           # b: one tap per chunk, pair (kk, nt) at 4 kk + nt
        bh, bl = frags(b9[:, :, t].contiguous())
        for kk in range(2):
            for nt in range(4):
                pairs.append((bh[nt, kk], bl[nt, kk]))
    ch, cl = frags(c2)
    for jc in range(4):          # c: chunk jc = n-tiles 4 jc .. + 3, pair (nn, kk) at 2 nn + kk
        for nn in range(4):
            for kk in range(2):
                pairs.append((ch[4 * jc + nn, kk], cl[4 * jc + nn, kk]))
    wfrag = torch.stack([torch.stack([h, l]) for h, l in pairs]).contiguous().to(device)  # [136 pairs][2 planes][64][8]
    coef = torch.cat([1.0 / sa, ba, 1.0 / sb, bb, 1.0 / sc, bc]).float().contiguous().to(device)
    return wfrag, coef


def pack_pw(w, device):
    """BN-folded pointwise weights [N, K(,1,1,1)] -> csrc/pw_chain.hip's fragments [N/16][ceil(K/32)][64][8]: output
    rows permuted so a lane ends with 8 consecutive channels (include/avt.h), K zero-padded to whole 32-wide k-steps."""
    w = w.detach().float().cpu().reshape(w.shape[0], -1)
    n_out, k = w.shape
    ks = -(-k // 32)
    wp = torch.zeros((n_out, ks * 32))
    wp[:, :k] = w
    lane = torch.arange(64)
    n, q = lane & 15, lane >> 4
    e = torch.arange(8)
    shape = (n_out // 16, ks, 64, 8)
    nt = torch.arange(n_out // 16).view(-1, 1, 1, 1)
    row = (32 * (nt // 2) + 8 * (n >> 2).view(1, 1, -1, 1) + 4 * (nt % 2) + (n & 3).view(1, 1, -1, 1)).expand(shape)
    col = (torch.arange(ks).view(1, -1, 1, 1) * 32 + q.view(1, 1, -1, 1) * 8 + e.view(1, 1, 1, -1)).expand(shape)
    return wp[row, col].to(torch.bfloat16).contiguous().to(device)


class _Block:
    def __init__(self, blk, device):
        self.b1 = FusedConv(blk.branch1, blk.branch1_bn, False, device) if hasattr(blk, "branch1") else None
        t = blk.branch2
        self.a = FusedConv(t.a, t.a_bn, True, device)
        self.b = FusedConv(t.b, t.b_bn, True, device)
        self.c = FusedConv(t.c, t.c_bn, True, device)  # ReLU applied after the residual add (fused)
        self.dev = device
        # identity-shortcut fast-pathway blocks ([3,1,1] -> [1,3,3] -> [1,1,1], stride 1, width <= 32): one kernel
        self.fused = None
        if (_FUSE_BLOCK and self.b1 is None and self.a.kernel == (3, 1, 1) and self.b.kernel == (1, 3, 3) and
                self.c.kernel == (1, 1, 1) and self.a.stride == (1, 1, 1) and self.b.stride == (1, 1, 1) and
                self.a.cout <= 32 and self.c.cout == self.a.cin and self.c.cout in (32, 64, 128)):
            (wa, ba), (wb, bb), (wc, bc) = self.a._folded, self.b._folded, self.c._folded
            self.fused = pack_bottleneck(wa, ba, wb, bb, wc, bc, device)
        # ... and the first fast blocks with a 1x1x1 shortcut conv: res2 (8 -> 32, stride 1), res3 / res4 (32 -> 64,
        # 64 -> 128: b and the shortcut have spatial stride 2)
        self.fused_first = None
        first8 = self.a.cin == 8 and self.c.cout == 32 and self.b.stride == (1, 1, 1)
        strided = self.a.cin in (32, 64) and self.c.cout == 2 * self.a.cin and self.b.stride == (1, 2, 2)
        if (_FUSE_BLOCK and self.b1 is not None and self.a.kernel == (3, 1, 1) and self.b.kernel == (1, 3, 3) and
                self.c.kernel == (1, 1, 1) and self.a.stride == (1, 1, 1) and self.b1.stride == self.b.stride and
                (first8 or strided) and self.a.cout <= 32 and self.b1.kernel == (1, 1, 1)):
            (wa, ba), (wb, bb), (wc, bc) = self.a._folded, self.b._folded, self.c._folded
            self.fused_first = pack_bottleneck(wa, ba, wb, bb, wc, bc, device, shortcut=self.b1._folded)

        # slow res2: b ([1,3,3] 64 -> 64, stride 1) on the strip-resident kernel
        self.c33 = None
        if (_C33 and self.b.kernel == (1, 3, 3) and self.b.stride == (1, 1, 1) and self.b.cin == 64 and self.b.cout == 64 and
                self.b._folded is not None):
            self.c33 = (pack_c33(self.b._folded[0], device), self.b.bias)
        # first block of a stage with a stride-1 1x1x1 shortcut conv and pointwise a (slow res2): c and the shortcut are
        # ONE GEMM over K = [x | b-output] when b writes its output into spare columns of x's row buffer — no shortcut
        # launch, no residual read (self.extra = columns the caller must leave free behind x)
        self.ccat, self.extra, self._pw = None, 0, {}
        if (_FUSE_KCAT and self.b1 is not None and self.fused_first is None and self.a.kernel == (1, 1, 1) and
                self.c.kernel == (1, 1, 1) and self.b1.kernel == (1, 1, 1) and self.a.stride == (1, 1, 1) and
                self.b.stride == (1, 1, 1) and self.b1.stride == (1, 1, 1) and self.c.stride == (1, 1, 1)):
            (wsc, bsc), (wc, bc) = self.b1._folded, self.c._folded
            self.ccat = FusedConv(None, None, True, device,
                                  folded=(torch.cat([wsc, wc], 1), bc + bsc, (1, 1, 1), (0, 0, 0)))
            self.ccat.alg_flops_per_row = self.c.alg_flops_per_row + self.b1.alg_flops_per_row
            self.extra = self.c.cin
        # first block of a stage with a STRIDED 1x1x1 shortcut (slow res3 / res4 / res5): b ([1,3,3], stride 2) writes its
        # output behind x's channels in x's OWN rows (2 ho, 2 wo) (avt_conv3d_igemm_rows_bf16), so c and the shortcut are one
        # stride-2 pointwise GEMM over K = [x | b-output]: no shortcut launch, no shortcut tensor written and re-read
        self.scat = None
        if (_FUSE_SCAT and self.b1 is not None and self.fused_first is None and self.ccat is None and
                self.c.kernel == (1, 1, 1) and self.b1.kernel == (1, 1, 1) and self.b.kernel == (1, 3, 3) and
                self.b.stride == (1, 2, 2) and self.b1.stride == (1, 2, 2) and self.a.stride == (1, 1, 1) and
                self.c.stride == (1, 1, 1) and self.a.kernel[1:] == (1, 1) and self.b1._folded is not None):
            (wsc, bsc), (wc, bc) = self.b1._folded, self.c._folded
            self.scat = FusedConv(None, None, True, device,
                                  folded=(torch.cat([wsc, wc], 1), bc + bsc, (1, 2, 2), (0, 0, 0)))
            self.scat.alg_flops_per_row = self.c.alg_flops_per_row + self.b1.alg_flops_per_row
            self.extra = self.c.cin

    def _scat_ok(self, x):
        return (self.scat is not None and x.c0 == 0 and x.ld >= x.C + self.extra and x.C == self.a.cin and
                x.dims[2] % 2 == 0 and x.dims[3] % 2 == 0)

    def can_chain(self, nxt, x):
        """True when this block's c (+ residual) and the next block's a run as ONE pointwise pass (csrc/pw_chain.hip).
        nxt may be the first block of the next stage: its a then also reads the lateral features behind y (x2)."""
        if not _CHAIN or nxt is None or nxt.a.kernel != (1, 1, 1) or nxt.a.stride != (1, 1, 1):
            return False
        if self.c.kernel != (1, 1, 1) or self.c._folded is None or nxt.a._folded is None or nxt.a.cin < self.c.cout:
            return False
        if self.c.cout > _CHAIN_MAXN or self._scat_ok(x):
            return False
        if any(v is not None for v in (self.fused, self.fused_first, nxt.fused, nxt.fused_first, nxt.ccat)):
            return False
        kcat = self.ccat is not None and x.c0 == 0 and x.ld >= x.C + self.extra and x.C == self.a.cin
        k1 = x.C + self.extra if kcat else self.c.cin
        return ops.pw_chain_supported(k1, self.c.cout, nxt.a.cout, not kcat, nxt.a.cin - self.c.cout)

    def _chain(self, x1, k1, first, res, nxt, dims, y=None, x2=None):
        """-> (y, z): y = relu(first(x1) [+ res]) (this block's output), z = relu(nxt.a([y | x2])) (the next block's a)."""
        key = (id(first), id(nxt))
        if self._pw.get("key") != key:
            self._pw = {"key": key, "w1": pack_pw(first._folded[0], self.dev), "w2": pack_pw(nxt.a._folded[0], self.dev)}
        m = dims[0] * dims[1] * dims[2] * dims[3]
        n1, n2 = first.cout, nxt.a.cout
        k2x = nxt.a.cin - n1
        if (k2x > 0) != (x2 is not None) or (x2 is not None and x2.C != k2x):
            raise AvtError("pw_chain: the next a conv reads %d channels, got y (%d) + x2 (%s)" % (nxt.a.cin, n1, x2 and x2.C))
        if y is None:
            y = Act(torch.empty((m, n1), dtype=torch.bfloat16, device=self.dev), dims)
        z = Act(torch.empty((m, n2), dtype=torch.bfloat16, device=self.dev), dims)

        def launch():
            ops.pw_chain(x1.ptr, x1.ld, k1, self._pw["w1"], first.bias, res.ptr if res is not None else 0,
                         res.ld if res is not None else 0, y.ptr, y.ld, n1, self._pw["w2"], nxt.a.bias, z.ptr, z.ld, n2, m,
                         x2_ptr=x2.ptr if x2 is not None else 0, ldx2=x2.ld if x2 is not None else 0, k2x=k2x)

        if PROFILER is None:
            launch()
        else:
            PROFILER("pw_chain_kernel", launch, m * (first.alg_flops_per_row + nxt.a.alg_flops_per_row),
                     2.0 * m * (k1 + n1 * (2 if res is not None else 1) + k2x + n2))
        return y, z

    def _b(self, m, out=None):
        """The block's b conv: the strip-resident kernel for 64 -> 64 at the production width, else the implicit GEMM."""
        if (self.c33 is not None and m.c0 == 0 and m.ld == m.C and ops.conv33_c64_supported(m.C, self.b.cout, m.dims[3])):
            b, t, h, w = m.dims
            if out is None:
                out = Act(torch.empty((b * t * h * w, self.b.cout), dtype=torch.bfloat16, device=self.dev), m.dims)

            def launch():
                ops.conv33_c64(m.ptr, self.c33[0], self.c33[1], out.ptr, b, t, h, w, out.ld, relu=True)

            if PROFILER is None:
                launch()
            else:
                rows = b * t * h * w
                PROFILER("c33_kernel", launch, rows * self.b.alg_flops_per_row, 2.0 * rows * 2 * self.b.cout)
            return out
        return self.b(m, out=out)

    def __call__(self, x, out=None, chain=None, a_pre=None, x2=None):
        """chain = the next block (can_chain(...) holds): returns (y, a-output of the next block) — y written to `out`
        when given, x2 = the further inputs of that a conv (stage boundary); a_pre = this block's a-output when the
        previous block's chained pass has already produced it."""
        if self.ccat is not None and x.c0 == 0 and x.ld >= x.C + self.extra and x.C == self.a.cin:
            self._b(self.a(x), out=Act(x.buf, x.dims, x.C, self.extra))  # b's output lands behind x in the same rows
            if chain is not None:
                return self._chain(Act(x.buf, x.dims, 0, x.C + self.extra), x.C + self.extra, self.ccat, None, chain, x.dims)
            return self.ccat(Act(x.buf, x.dims, 0, x.C + self.extra), out=out)
        if self._scat_ok(x):
            m = a_pre if a_pre is not None else self.a(x)
            self.b(m, out=Act(x.buf, x.dims, x.C, self.extra), out_rows=(2, x.dims[2], x.dims[3]))
            return self.scat(Act(x.buf, x.dims, 0, x.C + self.extra), out=out)
        if chain is not None or a_pre is not None:
            sc = self.b1(x) if self.b1 is not None else x
            m = self._b(a_pre if a_pre is not None else self.a(x))
            if chain is not None:
                return self._chain(m, self.c.cin, self.c, sc, chain, m.dims, y=out, x2=x2)
            return self.c(m, out=out, res=sc, relu=True)
        if (self.fused is not None and out is None and x.c0 == 0 and x.ld == x.C and
                ops.bottleneck_fused_supported(x.C, x.dims[3])):
            b, t, h, w = x.dims
            y = Act(torch.empty((b * t * h * w, x.C), dtype=torch.bfloat16, device=self.dev), x.dims)

            def launch():
                ops.bottleneck_fused(x.ptr, y.ptr, self.fused, b, t, h, w, x.C, tchunk=_FUSE_TCHUNK or (16 if w >= 56 else 32))

            if PROFILER is None:
                launch()
            else:
                m = b * t * h * w
                fl = m * (self.a.alg_flops_per_row + self.b.alg_flops_per_row + self.c.alg_flops_per_row)
                PROFILER("bottleneck_kernel", launch, fl, 2.0 * (2 * m * x.C))
            return y
        if (self.fused_first is not None and out is None and x.c0 == 0 and x.ld == x.C and
                ops.bottleneck_first_supported(x.C, self.c.cout, x.dims[3]) and x.dims[2] % self.b.stride[1] == 0):
            b, t, h, w = x.dims
            st = self.b.stride[1]
            od = (b, t, h // st, w // st)
            mo = od[0] * od[1] * od[2] * od[3]
            y = Act(torch.empty((mo, self.c.cout), dtype=torch.bfloat16, device=self.dev), od)

            def launch():
                ops.bottleneck_first(x.ptr, y.ptr, self.fused_first, b, t, h, w, x.C, self.c.cout,
                                     tchunk=_FUSE_TCHUNK or (16 if w >= 56 else 32))

            if PROFILER is None:
                launch()
            else:
                m = b * t * h * w
                fl = (m * self.a.alg_flops_per_row +
                      mo * (self.b.alg_flops_per_row + self.c.alg_flops_per_row + self.b1.alg_flops_per_row))
                PROFILER("bottleneck_kernel", launch, fl, 2.0 * (m * x.C + mo * self.c.cout))
            return y
        sc = self.b1(x) if self.b1 is not None else x
        return self.c(self._b(self.a(x)), out=out, res=sc, relu=True)


class _BlockX3:
    """A residual block in the contract-grade mode.  Fast-pathway blocks with an identity shortcut (res2-4) and res2's
    8-channel first block run as ONE kernel (csrc/bneck_x3.hip: the x3 counterpart of the bf16 path's bottleneck_fused);
    every other block is four split-plane convolutions (shortcut, a, b, c + residual)."""

    def __init__(self, blk, device, x3):
        self.b1 = FusedConv(blk.branch1, blk.branch1_bn, False, device, x3=x3) if hasattr(blk, "branch1") else None
        t = blk.branch2
        self.a = FusedConv(t.a, t.a_bn, True, device, x3=x3)
        self.b = FusedConv(t.b, t.b_bn, True, device, x3=x3)
        self.c = FusedConv(t.c, t.c_bn, True, device, x3=x3)  # ReLU after the residual add (fused)
        self.x3, self.dev = x3, device
        self.fused = None
        shape_ok = (self.a.kernel == (3, 1, 1) and self.b.kernel == (1, 3, 3) and self.c.kernel == (1, 1, 1) and
                    self.a.stride == (1, 1, 1) and self.a.cout <= 32)
        s1 = self.b.stride == (1, 1, 1)
        ident = s1 and self.b1 is None and self.c.cout == self.a.cin and self.c.cout in (32, 64, 128)
        first8 = (s1 and self.b1 is not None and self.a.cin == 8 and self.c.cout == 32 and self.b1.kernel == (1, 1, 1) and
                  self.b1.stride == (1, 1, 1))
        strided = (self.b1 is not None and self.b.stride == (1, 2, 2) and self.b1.stride == (1, 2, 2) and
                   self.b1.kernel == (1, 1, 1) and self.a.cin in (32, 64) and self.c.cout == 2 * self.a.cin)
        self.st = 2 if strided else 1
        if _FUSE_BLOCK_X3 and shape_ok and (ident or first8 or strided):
            (wa, ba), (wb, bb), (wc, bc) = self.a._folded, self.b._folded, self.c._folded
            self.fused = pack_bottleneck_x3(wa, ba, wb, bb, wc, bc, x3, device,
                                            shortcut=self.b1._folded if (first8 or strided) else None)
        # first block of a stage with a stride-1 1x1x1 shortcut conv and a pointwise a (slow res2): c and the shortcut are ONE
        # pointwise GEMM over K = [x | b-output] when b writes its output into spare columns behind x's channels — no shortcut
        # launch, no shortcut tensor written and read back as the residual (self.extra = columns the caller leaves free)
        # slow res2: b ([1,3,3] 64 -> 64, stride 1) with the activations as direct MFMA operands (csrc/conv33_x3.hip)
        self.c33 = None
        if (_C33_X3 and self.fused is None and self.b.kernel == (1, 3, 3) and self.b.stride == (1, 1, 1) and
                self.b._folded is not None and ops.conv33_x3_supported(self.b.cin, self.b.cout)):
            self.c33 = pack_c33_x3(self.b._folded[0], self.b._folded[1], x3, device)
        # slow res2's identity blocks (256 -> 64 -> [1,3,3] 64 -> 256 at width 56): the whole bottleneck in ONE kernel, the weights
        # streamed from L2 (csrc/res2_x3.hip, round 6; fp16 planes)
        self.res2 = None
        if (_RES2_X3 and x3 == ops.X3_F16 and self.fused is None and self.b1 is None and self.a.kernel == (1, 1, 1) and
                self.a.stride == (1, 1, 1) and self.b.kernel == (1, 3, 3) and self.b.stride == (1, 1, 1) and self.c.kernel == (1, 1, 1) and
                self.c.stride == (1, 1, 1) and self.a.cin == self.c.cout and all(v._folded is not None for v in (self.a, self.b, self.c)) and
                ops.res2_x3_supported(self.c.cout, self.a.cout, 56)):
            (wa, ba), (wb, bb), (wc, bc) = self.a._folded, self.b._folded, self.c._folded
            self.res2 = pack_res2_x3(wa, ba, wb, bb, wc, bc, x3, device)
        self.ccat, self.extra = None, 0
        if (_FUSE_KCAT and self.b1 is not None and self.fused is None and self.a.kernel == (1, 1, 1) and
                self.c.kernel == (1, 1, 1) and self.b1.kernel == (1, 1, 1) and self.a.stride == (1, 1, 1) and
                self.b.stride == (1, 1, 1) and self.b1.stride == (1, 1, 1) and self.c.stride == (1, 1, 1)):
            (wsc, bsc), (wc, bc) = self.b1._folded, self.c._folded
            self.ccat = FusedConv(None, None, True, device, folded=(torch.cat([wsc, wc], 1), bc + bsc, (1, 1, 1), (0, 0, 0)), x3=x3)
            self.ccat.alg_flops_per_row = self.c.alg_flops_per_row + self.b1.alg_flops_per_row
            self.extra = self.c.cin

    def can_chain(self, nxt):
        """True when this block's c (+ residual + ReLU) and the next block's a (+ ReLU) run as ONE pointwise pass
        (csrc/pw_x3.hip, the chained form): both pointwise on the streaming kernel, identity shortcut here."""
        if self.res2 is not None or (nxt is not None and nxt.res2 is not None):
            return False  # (a block that runs as one kernel neither hands its c to a chain nor takes its a from one)
        return (_CHAIN_X3 and nxt is not None and self.b1 is None and self.fused is None and nxt.fused is None and
                getattr(self.c, "pw", None) is not None and getattr(nxt.a, "pw", None) is not None and nxt.ccat is None and
                nxt.a.cin == self.c.cout and ops.pw_chain_x3_supported(self.c.cin, self.c.cout, nxt.a.cout))

    def __call__(self, x, out=None, chain=None, a_pre=None):
        """chain = the next block (can_chain(...) holds): returns (y, a-output of the next block); a_pre = this block's a-output
        when the previous block's chained pass has already produced it."""
        if chain is not None or a_pre is not None:
            m = self._b(a_pre if a_pre is not None else self.a(x))
            sc = self.b1(x) if self.b1 is not None else x
            if chain is None:
                return self.c(m, out=out, res=sc, relu=True)
            rows = m.dims[0] * m.dims[1] * m.dims[2] * m.dims[3]
            y = out if out is not None else new_act(rows, self.c.cout, m.dims, self.dev, True)
            z = new_act(rows, chain.a.cout, m.dims, self.dev, True)

            def launch():
                ops.pw_chain_x3(m.ptrs, m.ld, self.c.cin, self.c.pw, self.c.bias, self.c.wscale, sc.ptrs, sc.ld, y.ptrs, y.ld,
                                self.c.cout, True, chain.a.pw, chain.a.bias, chain.a.wscale, z.ptrs, z.ld, chain.a.cout, rows,
                                self.x3)

            if PROFILER is None:
                launch()
            else:
                PROFILER("pw_chain_x3_kernel", launch, rows * (self.c.alg_flops_per_row + chain.a.alg_flops_per_row),
                         4.0 * rows * (self.c.cin + 2 * self.c.cout + chain.a.cout))
            return y, z
        if (self.res2 is not None and chain is None and a_pre is None and x.lo is not None and x.C == self.a.cin and
                ops.res2_x3_supported(self.c.cout, self.a.cout, x.dims[3])):
            b, t, h, w = x.dims
            rows = b * t * h * w
            y = out if out is not None else new_act(rows, self.c.cout, x.dims, self.dev, True)

            def launch():
                ops.res2_x3(x.ptrs, x.ld, y.ptrs, y.ld, self.res2, b, t, h, w, self.x3)

            if PROFILER is None:
                launch()
            else:
                fl = rows * (self.a.alg_flops_per_row + self.b.alg_flops_per_row + self.c.alg_flops_per_row)
                PROFILER("res2_x3_kernel", launch, fl, 4.0 * rows * 2 * self.c.cout)
            return y
        if (self.fused is not None and out is None and x.c0 == 0 and x.ld == x.C and
                ops.bneck_x3_supported(x.C, self.c.cout, x.dims[3]) and x.dims[2] % self.st == 0):
            b, t, h, w = x.dims
            od = (b, t, h // self.st, w // self.st)
            mo = od[0] * od[1] * od[2] * od[3]
            y = new_act(mo, self.c.cout, od, self.dev, True)

            def launch():
                ops.bneck_x3(x.ptrs, y.ptrs, self.fused, b, t, h, w, x.C, self.c.cout, self.x3,
                             tchunk=_FUSE_TCHUNK_X3 or (11 if w == 14 else 8))  # measured: profiles/r03/probe_bneck_x3_tchunk.log;
                # 14-wide stage (round 4): 11 frames per workgroup = 3 chunks per clip: 996 workgroups = 3.9 rounds of the 256 CUs at
                # 166 clips (4 chunks: 5.2 rounds) and 2 halo frames per 11 instead of per 8 (profiles/r04/probe_bneck_early_load_ab.log)

            if PROFILER is None:
                launch()
            else:
                m = b * t * h * w
                fl = m * self.a.alg_flops_per_row + mo * (self.b.alg_flops_per_row + self.c.alg_flops_per_row +
                                                          (self.b1.alg_flops_per_row if self.b1 is not None else 0.0))
                PROFILER("bneck_x3_kernel", launch, fl, 4.0 * (m * x.C + mo * self.c.cout))
            return y
        if self.ccat is not None and x.c0 == 0 and x.lo is not None and x.ld >= x.C + self.extra and x.C == self.a.cin:
            self._b(self.a(x), out=Act(x.buf, x.dims, x.C, self.extra, lo=x.lo))  # b's output lands behind x in the same rows
            return self.ccat(Act(x.buf, x.dims, 0, x.C + self.extra, lo=x.lo), out=out)
        sc = self.b1(x) if self.b1 is not None else x
        return self.c(self._b(self.a(x)), out=out, res=sc, relu=True)

    def _b(self, m, out=None):
        """The block's b conv: the direct-operand kernel for 64 -> 64 [1,3,3], else the implicit GEMM."""
        if self.c33 is None:
            return self.b(m, out=out)
        b, t, h, w = m.dims
        if out is None:
            out = new_act(b * t * h * w, self.b.cout, m.dims, self.dev, True)

        def launch():
            ops.conv33_x3(m.ptrs, self.c33, out.ptrs, b, t, h, w, m.ld, out.ld, self.x3, relu=True)

        if PROFILER is None:
            launch()
        else:
            rows = b * t * h * w
            PROFILER("conv33_x3_kernel", launch, rows * self.b.alg_flops_per_row, 4.0 * rows * 2 * self.b.cout)
        return out


PRECISIONS = {"bf16": None, "bf16x3": ops.X3_BF16, "f16x3": ops.X3_F16}


class SlowFastMFMA(nn.Module):
    """Drop-in for a `SlowFast` module at inference time (eval-mode BatchNorm statistics).

    precision = "bf16"  : the fast path — bf16 activations and weights, every fused kernel (2^-9 per element and layer:
                          embeddings ~5 % from the fp32 module's on BN-calibrated weights, scores off by up to 1e-1);
                "bf16x3": contract grade — split-bf16 planes, three MFMA passes per product (csrc/conv_x3.hip);
                "f16x3" : the same with fp16 planes (11 + 11 bits instead of 8 + 8; assumes |activation| < 65504)."""

    out_dim = 2304

    input_layout = "ndhwc4"  # what ops.clip_pack should emit for forward_ndhwc4

    def __init__(self, model, device, precision="bf16"):
        super().__init__()
        if precision not in PRECISIONS:
            raise AvtError("SlowFastMFMA: precision must be one of %s" % sorted(PRECISIONS))
        self.dev = torch.device(device)
        self.precision = precision
        self.x3 = x3 = PRECISIONS[precision]
        self.planes = precision if x3 is not None else None  # ops.clip_pack(planes=...) for forward_ndhwc4's input
        model = model.eval()
        self.stem_s = stem_conv(model.s1.pathway0_stem, self.dev, x3=x3)
        self.stem_f = stem_conv(model.s1.pathway1_stem, self.dev, tgroup=4, x3=x3)
        self._anchor = nn.Parameter(torch.zeros(1, dtype=torch.bfloat16, device=self.dev), requires_grad=False)
        self.fuse = [FusedConv(f.conv_f2s, f.bn, True, self.dev, x3=x3) for f in (model.s1_fuse, model.s2_fuse,
                                                                                   model.s3_fuse, model.s4_fuse)]
        self.stages = []
        for s in (model.s2, model.s3, model.s4, model.s5):
            mk = (lambda blk: _BlockX3(blk, self.dev, x3)) if x3 is not None else (lambda blk: _Block(blk, self.dev))
            self.stages.append([[mk(getattr(s, "pathway%d_res%d" % (p, i))) for i in range(s.depth)] for p in range(2)])

    @torch.no_grad()
    def forward(self, x):
        """[slow [B,3,8,H,W], fast [B,3,32,H,W]] (the plugin contract): one layout copy, then forward_ndhwc4."""
        def cl4(v):
            b, c, t, h, w = v.shape
            if self.x3 is not None:  # split the fp32 clip into its two planes
                f = torch.zeros((b, t, h, w, 4), dtype=torch.float32, device=self.dev)
                f[..., :3] = v.to(self.dev, torch.float32).permute(0, 2, 3, 4, 1)
                hi, lo = split_planes(f, self.x3)
                return ops.SplitClip(hi, lo, self.x3)
            o = torch.zeros((b, t, h, w, 4), dtype=torch.bfloat16, device=self.dev)
            o[..., :3] = v.to(self.dev, torch.bfloat16).permute(0, 2, 3, 4, 1)
            return o

        return self.forward_ndhwc4(cl4(x[0]), cl4(x[1]))

    def _stem(self, conv, clip, out=None):
        """clip [B,T,H,W,4] bf16 -> conv+BN+ReLU at (H/2, W/2) -> max-pool -> Act at (H/4, W/4)."""
        b, t, h, w, _ = clip.shape
        if t % conv.tgroup:
            raise AvtError("stem: %d frames do not split into groups of %d" % (t, conv.tgroup))
        x = Act(clip.view(b * t * h * (w // 2), 8), (b, t, h, w // 2))
        lds_path = _STEM_LDS and conv.wt_lds is not None and ops.stem_conv_supported(h, w // 2, conv.cout)
        kt, st, pt = conv.kernel[0], conv.stride[0], conv.pad[0]
        od = conv.out_dims(x.dims)
        m_out = od[0] * od[1] * od[2] * od[3]
        if lds_path and _STEM_POOL and (kt == 1 or _STEM_POOL > 1) and (h // 2) % 8 == 0:
            # production shape: patch-resident stem kernel with the max-pool fused (the conv output stays on chip)
            pd = (b, t, od[2] // 2, od[3] // 2)
            cf = conv.frame_channels
            if out is None:
                out = Act(torch.empty((pd[0] * pd[1] * pd[2] * pd[3], cf), dtype=torch.bfloat16, device=self.dev), pd)

            def launch():
                ops.stem_conv_pool(x.ptr, conv.wt_lds, conv.bias, out.ptr, b, t, h, w // 2, conv.cout, kt, st, pt,
                                   conv.tgroup, out.ld)

            if PROFILER is None:
                launch()
            else:
                PROFILER("stem_kernel", launch, m_out * conv.alg_flops_per_row,
                         2.0 * (x.buf.numel() + pd[0] * pd[1] * pd[2] * pd[3] * cf) + conv.wt.numel() * 2)
            return out, pd
        if lds_path:
            y = Act(torch.empty((m_out, conv.cout), dtype=torch.bfloat16, device=self.dev), od)

            def launch():
                ops.stem_conv(x.ptr, conv.wt_lds, conv.bias, y.ptr, b, t, h, w // 2, conv.cout, kt, st, pt, relu=True)

            if PROFILER is None:
                launch()
            else:
                PROFILER("stem_kernel", launch, m_out * conv.alg_flops_per_row,
                         2.0 * (x.buf.numel() + m_out * conv.cout) + conv.wt.numel() * 2)
        else:
            y = conv(x)
        _, tg, h2, w2 = y.dims
        pd = (b, t, (h2 - 1) // 2 + 1, (w2 - 1) // 2 + 1)
        cf = conv.frame_channels
        if out is None:
            out = Act(torch.empty((pd[0] * pd[1] * pd[2] * pd[3], cf), dtype=torch.bfloat16, device=self.dev), pd)
        ops.maxpool_hw3s2(y.ptr, out.ptr, b * tg, h2, w2, conv.cout, y.ld, out.ld, tgroup=conv.tgroup)
        return out, pd

    def _stem_x3(self, conv, clip, out=None):
        """Contract-grade stem: pixel-pair convolution on the plain split-plane kernel, then the plane-pair max-pool."""
        b, t, h, w, _ = clip.shape
        lds_path = _STEM_LDS and conv.wt_lds_lo is not None and ops.stem_conv_supported(h, w // 2, conv.cout)
        table = isinstance(clip, ops.FrameClip)
        if table and not lds_path:  # (shapes the patch-resident kernel does not cover: gather the clips the table stands for)
            clip, table = clip.dense(), False
        # a frame table [F, h, w, 4] read through clip.idx (round 4), or the dense clips [b, t, h, w, 4]
        nf = clip.table_frames if table else b * t
        kt = conv.kernel[0]
        # a stem without temporal taps (the slow pathway's [1,7,7]) is a per-frame function: on a frame table it runs ONCE per
        # distinct frame and the pool hands every (window, slot) its frame (overlapping windows share about half of them)
        per_frame = table and kt == 1 and conv.tgroup == 1 and conv.stride[0] == 1
        sb, st_ = (nf, 1) if per_frame else (b, t)
        x = Act(clip.hi.view(nf * h * (w // 2), 8), (sb, st_, h, w // 2), lo=clip.lo.view(nf * h * (w // 2), 8))
        pool_idx = clip.idx.reshape(-1) if per_frame else None
        merged = None
        if (table and lds_path and _STEM_MERGE and conv.frames_per_tile == 2 and getattr(conv, "wg", None) is not None and
                clip.start is not None and t == 32):
            merged = merged_stem_taps(conv, clip.win_len, self.dev)
        if merged is not None:
            # the fast stem on a frame table, the frame taps of one source frame summed into one (5 patches and weight slabs per
            # output-frame group instead of 8 at W = 20)
            od = conv.out_dims(x.dims)
            m_out = od[0] * od[1] * od[2] * od[3]
            y = new_act(m_out, conv.cout, od, self.dev, True)
            src = merged["src"]
            taps = torch.where(src.unsqueeze(0) >= 0, clip.start.view(-1, 1, 1) + src.unsqueeze(0), src.unsqueeze(0)).contiguous()

            def launch():
                ops.stem_conv_x3_merged(x.ptrs, merged["wt_hi"], merged["wt_lo"], merged["bias"], merged["wscale"], y.ptrs, b, t, h,
                                        w // 2, conv.cout, conv.kernel[0], conv.stride[0], conv.pad[0], self.x3, taps,
                                        merged["tiles"], merged["ktm"], nf, relu=True)

            if PROFILER is None:
                launch()
            else:
                PROFILER("stem_kernel<x3>", launch, m_out * conv.alg_flops_per_row,
                         4.0 * (x.buf.numel() + m_out * conv.cout) + conv.wt.numel() * 4)
        elif lds_path:
            # production shape: the patch-resident stem kernel in its plane-pair form (no im2col gather)
            od = conv.out_dims(x.dims)
            m_out = od[0] * od[1] * od[2] * od[3]
            y = new_act(m_out, conv.cout, od, self.dev, True)

            def launch():
                ops.stem_conv_x3(x.ptrs, conv.wt_lds, conv.wt_lds_lo, conv.bias, conv.wscale, y.ptrs, sb, st_, h, w // 2,
                                 conv.cout, conv.kernel[0], conv.stride[0], conv.pad[0], self.x3, relu=True,
                                 frames_per_tile=conv.frames_per_tile, frame_idx=clip.idx if (table and not per_frame) else None,
                                 table_frames=nf if (table and not per_frame) else 0)

            if PROFILER is None:
                launch()
            else:
                PROFILER("stem_kernel<x3>", launch, m_out * conv.alg_flops_per_row,
                         4.0 * (x.buf.numel() + m_out * conv.cout) + conv.wt.numel() * 4)
        else:
            y = conv(x)
        _, tg, h2, w2 = y.dims
        pd = (b, t, (h2 - 1) // 2 + 1, (w2 - 1) // 2 + 1)
        cf = conv.frame_channels
        if out is None:
            out = new_act(pd[0] * pd[1] * pd[2] * pd[3], cf, pd, self.dev, True)
        if per_frame:
            ops.maxpool_hw3s2_x3(y.ptrs, out.ptrs, b * t, h2, w2, conv.cout, y.ld, out.ld, self.x3, frame_idx=pool_idx)
        else:
            ops.maxpool_hw3s2_x3(y.ptrs, out.ptrs, b * tg, h2, w2, conv.cout, y.ld, out.ld, self.x3, tgroup=conv.tgroup)
        return out, pd

    @torch.no_grad()
    def _forward_x3(self, slow, fast):
        """The contract-grade forward: the module's layers one by one on split-plane activations (no fused forms)."""
        b = slow.shape[0]
        f_act, df = self._stem_x3(self.stem_f, fast)
        cs, cf = self.stem_s.frame_channels, self.stem_f.frame_channels
        hs, ws = (slow.shape[2] // 2 - 1) // 2 + 1, (slow.shape[3] // 2 - 1) // 2 + 1
        ds = (b, slow.shape[1], hs, ws)
        extra = getattr(self.stages[0][0][0], "extra", 0)  # spare columns for the first slow block's K-concatenated c
        cat = new_act(ds[0] * ds[1] * ds[2] * ds[3], cs + 2 * cf + extra, ds, self.dev, True)
        sl = lambda a, c0, c: Act(a.buf, a.dims, c0, c, lo=a.lo)
        self._stem_x3(self.stem_s, slow, out=sl(cat, 0, cs))
        self.fuse[0](f_act, out=sl(cat, cs, 2 * cf))
        s_act = sl(cat, 0, cs + 2 * cf)
        for k, (slow_blocks, fast_blocks) in enumerate(self.stages):
            for blk in fast_blocks:
                f_act = blk(f_act)
            last = k == len(self.stages) - 1
            pre = None  # the a-output of the coming slow block, when the previous block's chained pass produced it
            for i, blk in enumerate(slow_blocks):
                nxt = slow_blocks[i + 1] if i + 1 < len(slow_blocks) else None
                if blk.can_chain(nxt):
                    s_act, pre = blk(s_act, chain=nxt, a_pre=pre)
                    continue
                a_pre, pre = pre, None
                if i == len(slow_blocks) - 1 and not last:  # straight into the next fusion's concat buffer
                    od = blk.b.out_dims(blk.a.out_dims(s_act.dims))
                    cs, cf = blk.c.cout, f_act.C
                    cat = new_act(od[0] * od[1] * od[2] * od[3], cs + 2 * cf, od, self.dev, True)
                    blk(s_act, out=sl(cat, 0, cs), a_pre=a_pre)
                    self.fuse[k + 1](f_act, out=sl(cat, cs, 2 * cf))
                    s_act = cat
                else:
                    s_act = blk(s_act, a_pre=a_pre)
        emb = torch.empty((b, s_act.C + f_act.C), dtype=torch.float32, device=self.dev)
        ops.mean_positions_x3(s_act.ptrs, b, s_act.buf.shape[0] // b, s_act.C, s_act.ld, emb, 0, self.x3)
        ops.mean_positions_x3(f_act.ptrs, b, f_act.buf.shape[0] // b, f_act.C, f_act.ld, emb, s_act.C, self.x3)
        return emb

    @torch.no_grad()
    def forward_ndhwc4(self, slow, fast):
        """slow [B,8,H,W,4], fast [B,32,H,W,4] bf16 channels-last clips (ops.clip_pack layout "ndhwc4"); ops.SplitClip
        pairs (clip_pack(..., planes=self.planes)) in the contract-grade modes."""
        if self.x3 is not None:
            return self._forward_x3(slow, fast)
        b = slow.shape[0]
        f_act, df = self._stem(self.stem_f, fast)
        cs, cf = self.stem_s.frame_channels, self.stem_f.frame_channels
        hs, ws = (slow.shape[2] // 2 - 1) // 2 + 1, (slow.shape[3] // 2 - 1) // 2 + 1
        ds = (b, slow.shape[1], hs, ws)
        # the slow stem is pooled straight into the concat buffer of the first lateral fusion
        extra = self.stages[0][0][0].extra  # spare columns for the first slow block's K-concatenated c (see _Block)
        sbuf = torch.empty((ds[0] * ds[1] * ds[2] * ds[3], cs + 2 * cf + extra), dtype=torch.bfloat16, device=self.dev)
        self._stem(self.stem_s, slow, out=Act(sbuf, ds, 0, cs))
        self.fuse[0](f_act, out=Act(sbuf, ds, cs, 2 * cf))
        s_act = Act(sbuf, ds, 0, cs + 2 * cf)
        pre = None  # the a-output of the coming slow block, when the previous block's chained pass produced it
        for k, (slow_blocks, fast_blocks) in enumerate(self.stages):
            for blk in fast_blocks:
                f_act = blk(f_act)
            last = k == len(self.stages) - 1
            for i, blk in enumerate(slow_blocks):
                nxt = slow_blocks[i + 1] if i + 1 < len(slow_blocks) else None
                if blk.can_chain(nxt, s_act):
                    s_act, pre = blk(s_act, chain=nxt, a_pre=pre)
                    continue
                a_pre, pre = pre, None
                if i == len(slow_blocks) - 1 and not last:
                    # last slow block of the stage writes straight into the next fusion's concat buffer
                    od = blk.b.out_dims(blk.a.out_dims(s_act.dims))
                    cs, cf = blk.c.cout, f_act.C
                    nxt0 = self.stages[k + 1][0][0]
                    sbuf = torch.empty((od[0] * od[1] * od[2] * od[3], cs + 2 * cf + nxt0.extra), dtype=torch.bfloat16,
                                       device=self.dev)  # (+ spare columns for the next block's K-concatenated c)
                    if blk.can_chain(nxt0, s_act):
                        # ... and the next stage's first a conv reads [y | lateral]: lateral first, then one chained pass
                        self.fuse[k + 1](f_act, out=Act(sbuf, od, cs, 2 * cf))
                        _, pre = blk(s_act, out=Act(sbuf, od, 0, cs), chain=nxt0, a_pre=a_pre, x2=Act(sbuf, od, cs, 2 * cf))
                    else:
                        blk(s_act, out=Act(sbuf, od, 0, cs), a_pre=a_pre)
                        self.fuse[k + 1](f_act, out=Act(sbuf, od, cs, 2 * cf))
                    s_act = Act(sbuf, od, 0, cs + 2 * cf)
                else:
                    s_act = blk(s_act, a_pre=a_pre)
        # head (models.py:576-580 surgery): global average pool per pathway, concat slow | fast
        emb = torch.empty((b, s_act.C + f_act.C), dtype=torch.float32, device=self.dev)
        ops.mean_positions(s_act.ptr, b, s_act.buf.shape[0] // b, s_act.C, s_act.ld, emb, 0)
        ops.mean_positions(f_act.ptr, b, f_act.buf.shape[0] // b, f_act.C, f_act.ld, emb, s_act.C)
        return emb
