"""SuperSloMo interpolation at the jumps of the stitched video, on the MI355X.

The reference smooths every jump of the synthesised video with `SF - 1` intermediate frames from SuperSloMo
(contrastive_video_textures/interpolate.py:75-147 `interpolate`, models/slowmo.py:137-284 `UNet` / `backWarp`; created
in validate.py:179-185, called at :588-611; `--interpolation` is ON by default, main.py:95).  Here:

  * `UNet` is the weight container — a plain torch module with the reference's parameter names, so the two state dicts
    of SuperSloMo.ckpt (`state_dictFC`, `state_dictAT`, validate.py:183-185) load unchanged.  It also runs (torch ops) but
    the product never calls its forward.
  * `UNetX3` executes a UNet on the split-plane implicit-GEMM kernel (csrc/conv_x3.hip, fp16 planes, LeakyReLU(0.1)
    epilogue): NHWC rows, skip connections written straight into the second half of the concat buffer their `up` block
    reads, pooling / upsampling as plane-pair passes (csrc/interp.hip).
  * `Interpolator` is `interpolate.forward`: pack the frame pair, flowComp, then ALL `SF - 1` intermediate times as one
    batch through ArbTimeFlowIntrp (the reference loops over them), back-warps / blend / uint8 conversion fused in two
    passes.  Frames stay on the device from the source video to the output tensor.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from ._lib import AvtError
from .fused_slowfast import Act, FusedConv, new_act

MEAN = (0.429, 0.431, 0.397)  # interpolate.py:51 (std = 1)


class _Down(nn.Module):
    """avg-pool 2 -> conv k -> LeakyReLU(0.1) -> conv k -> LeakyReLU(0.1)   (slowmo.py:10-72)"""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, k, stride=1, padding=(k - 1) // 2)
        self.conv2 = nn.Conv2d(cout, cout, k, stride=1, padding=(k - 1) // 2)

    def forward(self, x):
        x = F.avg_pool2d(x, 2)
        x = F.leaky_relu(self.conv1(x), negative_slope=0.1)
        return F.leaky_relu(self.conv2(x), negative_slope=0.1)


class _Up(nn.Module):
    """bilinear x2 -> conv 3 -> LeakyReLU -> conv 3 over cat(x, skip) -> LeakyReLU   (slowmo.py:74-135)"""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride=1, padding=1)
        self.conv2 = nn.Conv2d(2 * cout, cout, 3, stride=1, padding=1)

    def forward(self, x, skip):
        x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
        x = F.leaky_relu(self.conv1(x), negative_slope=0.1)
        return F.leaky_relu(self.conv2(torch.cat((x, skip), 1)), negative_slope=0.1)


class UNet(nn.Module):
    """The SuperSloMo UNet (slowmo.py:137-208): 7x7, 7x7, five down blocks (5, 3, 3, 3, 3), five up blocks, 3x3 head."""

    WIDTHS = (32, 64, 128, 256, 512, 512)
    DOWN_K = (5, 3, 3, 3, 3)

    def __init__(self, cin, cout):
        super().__init__()
        w = self.WIDTHS
        self.conv1 = nn.Conv2d(cin, w[0], 7, stride=1, padding=3)
        self.conv2 = nn.Conv2d(w[0], w[0], 7, stride=1, padding=3)
        for i, k in enumerate(self.DOWN_K):
            setattr(self, "down%d" % (i + 1), _Down(w[i], w[i + 1], k))
        ups = ((512, 512), (512, 256), (256, 128), (128, 64), (64, 32))
        for i, (a, b) in enumerate(ups):
            setattr(self, "up%d" % (i + 1), _Up(a, b))
        self.conv3 = nn.Conv2d(w[0], cout, 3, stride=1, padding=1)

    def forward(self, x):
        x = F.leaky_relu(self.conv1(x), negative_slope=0.1)
        skips = [F.leaky_relu(self.conv2(x), negative_slope=0.1)]
        for i in range(5):
            skips.append(getattr(self, "down%d" % (i + 1))(skips[-1]))
        x = skips.pop()
        for i in range(5):
            x = getattr(self, "up%d" % (i + 1))(x, skips.pop())
        return F.leaky_relu(self.conv3(x), negative_slope=0.1)


def _pad8(n):
    return (n + 7) // 8 * 8


def _conv(conv, device, cin_pad=None, cout_pad=None):
    """nn.Conv2d -> FusedConv on the split-plane kernel (a [1, k, k] convolution over T = 1), channels zero-padded to the
    kernel's multiples of 8 (zero input taps / zero filters with zero bias: LeakyReLU(0) = 0)."""
    w = conv.weight.detach().float().cpu()
    b = conv.bias.detach().float().cpu()
    cout, cin, kh, kw = w.shape
    ci, co = cin_pad or _pad8(cin), cout_pad or _pad8(cout)
    wp = torch.zeros((co, ci, 1, kh, kw))
    wp[:cout, :cin, 0] = w
    bp = torch.zeros(co)
    bp[:cout] = b
    return FusedConv(None, None, 2, device, folded=(wp, bp, (1, 1, 1), (0, conv.padding[0], conv.padding[1])), x3=ops.X3_F16)


class UNetX3:
    """A `UNet` on the HIP kernels.  forward(x: Act [B*H*W, cin_pad]) -> Act [B*H*W, pad8(cout)]; H, W multiples of 32."""

    def __init__(self, net, device):
        self.dev = device
        self.cin = _pad8(net.conv1.in_channels)
        self.conv1 = _conv(net.conv1, device)
        self.conv2 = _conv(net.conv2, device)
        self.down = [(_conv(getattr(net, "down%d" % i).conv1, device), _conv(getattr(net, "down%d" % i).conv2, device)) for i in range(1, 6)]
        self.up = [(_conv(getattr(net, "up%d" % i).conv1, device), _conv(getattr(net, "up%d" % i).conv2, device)) for i in range(1, 6)]
        self.conv3 = _conv(net.conv3, device)

    def _pool(self, x):
        b, _, h, w = x.dims
        y = new_act(b * (h // 2) * (w // 2), x.C, (b, 1, h // 2, w // 2), self.dev, True)
        ops.avgpool2_x3(x.ptrs, (b, h, w), x.C, x.ld, y.ptrs, y.ld, ops.X3_F16)
        return y

    def _upsample(self, x):
        b, _, h, w = x.dims
        y = new_act(b * 4 * h * w, x.C, (b, 1, 2 * h, 2 * w), self.dev, True)
        ops.upsample2_bilinear_x3(x.ptrs, (b, h, w), x.C, x.ld, y.ptrs, y.ld, ops.X3_F16)
        return y

    def forward(self, x):
        b, _, h, w = x.dims
        if h % 32 or w % 32:
            raise AvtError("UNetX3: frame extents must be multiples of 32 (got %d x %d)" % (h, w))
        widths = UNet.WIDTHS
        # one concat buffer per resolution: [x_up | skip]; the down path writes its skip into the second half
        cats = []
        for lvl in range(5):
            c = widths[lvl]
            hh, ww = h >> lvl, w >> lvl
            cats.append(new_act(b * hh * ww, 2 * c, (b, 1, hh, ww), self.dev, True))

        def half(lvl, second):
            cat, c = cats[lvl], widths[lvl]
            return Act(cat.buf, cat.dims, c0=c if second else 0, C=c, lo=cat.lo)

        y = self.conv1(x)
        skip = self.conv2(y, out=half(0, True))
        for lvl in range(5):
            c1, c2 = self.down[lvl]
            y = c1(self._pool(skip))
            skip = c2(y, out=half(lvl + 1, True)) if lvl < 4 else c2(y)
        y = skip  # [b, h/32, w/32, 512]
        for i in range(5):
            lvl = 4 - i
            c1, c2 = self.up[i]
            c1(self._upsample(y), out=half(lvl, False))
            y = c2(cats[lvl])
        return self.conv3(y)


class Interpolator:
    """`interpolate(origDim, SF)` of the reference on the device.  __call__(frame0, frame1) with uint8 [H, W, 3] RGB device
    tensors -> uint8 [SF - 1, H, W, 3], the frames the reference appends at a jump (validate.py:601-611)."""

    def __init__(self, height, width, sf, device, flow_comp=None, arb_time=None):
        if not 2 <= sf <= 17:
            raise AvtError("Interpolator: slomo factor must be in 2..17 (got %d)" % sf)
        self.h, self.w, self.sf, self.dev = height, width, sf, device
        # interpolate.py:64-66, 90: the networks run at the frame size rounded DOWN to a multiple of 32
        self.nh, self.nw = height // 32 * 32, width // 32 * 32
        if self.nh == 0 or self.nw == 0:
            raise AvtError("Interpolator: frames smaller than 32 x 32 cannot be interpolated (%d x %d)" % (height, width))
        self.flow_comp = flow_comp if flow_comp is not None else UNet(6, 4)      # interpolate.py:79
        self.arb_time = arb_time if arb_time is not None else UNet(20, 5)        # interpolate.py:82
        self._fc = self._at = None

    def load_checkpoint(self, path):
        """SuperSloMo.ckpt: {'state_dictFC': ..., 'state_dictAT': ...}   (validate.py:183-185)"""
        ck = torch.load(path, map_location="cpu")
        self.arb_time.load_state_dict(ck["state_dictAT"])
        self.flow_comp.load_state_dict(ck["state_dictFC"])
        self._fc = self._at = None

    def _resize(self, frame, size, lanczos):
        """PIL's resize for frames whose extents are not multiples of 32 (interpolate.py:43, 137): host round trip, as rare
        as such videos (the networks' size equals the frame's for 128, 224, 256 ...)."""
        from PIL import Image
        img = Image.fromarray(frame.cpu().numpy())
        img = img.resize(size, Image.LANCZOS if lanczos else Image.BILINEAR)  # (ANTIALIAS = LANCZOS before Pillow 10)
        import numpy as np
        return torch.from_numpy(np.array(img.convert("RGB"))).to(self.dev)

    def __call__(self, frame0, frame1):
        if self._fc is None:
            self._fc, self._at = UNetX3(self.flow_comp, self.dev), UNetX3(self.arb_time, self.dev)
        if tuple(frame0.shape) != (self.h, self.w, 3) or tuple(frame1.shape) != (self.h, self.w, 3):
            raise AvtError("Interpolator: frames must be [%d, %d, 3] uint8" % (self.h, self.w))
        h, w, sf, pd = self.nh, self.nw, self.sf, ops.X3_F16
        if (h, w) != (self.h, self.w):
            frame0, frame1 = self._resize(frame0, (w, h), True), self._resize(frame1, (w, h), True)
        frame0, frame1 = frame0.contiguous(), frame1.contiguous()
        img = torch.empty((2, h, w, 4), dtype=torch.float32, device=self.dev)
        x = new_act(h * w, 8, (1, 1, h, w), self.dev, True)
        ops.interp_pack_pair(frame0, frame1, MEAN, img, x.ptrs, pd)
        flow = self._fc.forward(x)                                   # [h*w, 8]: F_0_1, F_1_0, pad
        nt = sf - 1
        xin = new_act(nt * h * w, 24, (nt, 1, h, w), self.dev, True)
        ft = torch.empty((nt, h, w, 4), dtype=torch.float32, device=self.dev)
        ops.interp_mid_input(img, flow.ptrs, h, w, sf, xin.ptrs, ft, pd)
        o = self._at.forward(xin)                                    # [nt*h*w, 8]: 5 outputs + pad
        out = torch.empty((nt, h, w, 3), dtype=torch.uint8, device=self.dev)
        ops.interp_final(img, ft, o.ptrs, h, w, sf, MEAN, out, pd)
        if (h, w) != (self.h, self.w):
            out = torch.stack([self._resize(f, (self.w, self.h), False) for f in out])
        return out


class IntpTimeline:
    """`new_frames_intp` of validate.py:588-650: the output frame sequence at (SF + 1) / 2 times the frame rate.  Every
    source frame is shown 1 + int((SF - 1) / 2) times; at a jump the copies of the last frame are taken back, the SF - 1
    interpolated frames go in, and the first frame after the jump is shown once.  Entries are source-frame indices (int)
    or interpolated frames (uint8 [H, W, 3] tensors); `frames(video)` materialises them.

    The reference's own consistency check (validate.py:812: len == int((SF + 1) / 2) * len(new_frames)) only holds for odd
    SF >= 3 — an even SF adds one frame too many per jump, SF <= 2 makes `[: -0]` wipe the list (validate.py:591) — so
    other values are refused here instead of asserting at the end of the run."""

    def __init__(self, sf):
        if sf < 3 or sf % 2 == 0:
            raise AvtError("interpolation: the slomo factor must be odd and >= 3 (got %d): the reference's frame bookkeeping "
                           "(validate.py:591, 812) fails its own assert otherwise" % sf)
        self.sf, self.dup, self.items = sf, int((sf - 1) / 2), []

    def jump(self, frames):
        del self.items[-self.dup:]
        self.items.extend(frames)

    def append(self, idx, first_after_jump=False):
        self.items.append(int(idx))
        if not first_after_jump:
            self.items.extend([int(idx)] * self.dup)

    def __len__(self):
        return len(self.items)

    def frames(self, video):
        """-> uint8 [n, H, W, 3] (on the video's device): one gather of the source frames, interpolated frames scattered in."""
        video = torch.as_tensor(video)
        src = [i for i, e in enumerate(self.items) if isinstance(e, int)]
        out = torch.empty((len(self.items),) + tuple(video.shape[1:]), dtype=torch.uint8, device=video.device)
        if src:
            out[torch.as_tensor(src, device=video.device)] = video[torch.as_tensor([self.items[i] for i in src], device=video.device)]
        for i, e in enumerate(self.items):
            if not isinstance(e, int):
                out[i] = e.to(video.device)
        return out
