"""VGGish feature stack on the hand-written MFMA convolution (csrc/conv_igemm.hip) and 2x2 max-pool (csrc/pool.hip).

Takes the `vggish.VGGish` plugin (the module the reference builds at main.py:336-338; forward at
audio_models/vggish.py:42-46: features -> permute(0,2,3,1) -> flatten, `fc` never applied) and runs its six
conv3x3+ReLU layers as implicit-GEMM launches on NHWC bf16 rows (a [n,1,H,W] NDHWC tensor with one frame), so the
flatten the reference needs a permute for is the native layout: the last pool's rows ARE the [n, 6*4*512] output.

The first layer has ONE input channel.  Eight adjacent mel bins (W axis) are read as eight channels of one 16-byte
chunk (fused_slowfast.group_weights_w): a block-Toeplitz [8*64, 8] x 3 x 3 filter over bin groups computes the same
convolution with full chunks and full MFMA tiles (structured zeros instead of padding).

Same contract as the module it wraps: forward([n,1,100,64]) -> [n, 12288] fp32.  precision = "bf16": bf16 activations (the
fast path); "f16x3" / "bf16x3": the contract-grade split-plane arithmetic of csrc/conv_x3.hip (every tensor two 16-bit planes,
three MFMA passes per product, plane-pair 2x2 max-pool) — within 5e-5 of the fp32 module, so config 3's default path
(validate.py, --enc_dtype fp32) has no MIOpen convolution left.
"""
import torch
import torch.nn as nn

from . import ops
from ._lib import AvtError
from .fused_slowfast import PRECISIONS, Act, FusedConv, group_weights_w, new_act, split_planes


class VGGishMFMA(nn.Module):
    out_dim = 12288

    def __init__(self, model, device, precision="bf16"):
        super().__init__()
        if precision not in PRECISIONS:
            raise AvtError("VGGishMFMA: precision must be one of %s" % sorted(PRECISIONS))
        self.precision, self.x3 = precision, PRECISIONS[precision]
        x3 = self.x3
        self.dev = torch.device(device)
        self._anchor = nn.Parameter(torch.zeros(1, dtype=torch.bfloat16, device=self.dev), requires_grad=False)
        self.layers = []  # (FusedConv, pool_after)
        mods = list(model.features)
        for i, m in enumerate(mods):
            if not isinstance(m, nn.Conv2d):
                continue
            if tuple(m.kernel_size) != (3, 3) or tuple(m.stride) != (1, 1) or tuple(m.padding) != (1, 1):
                raise AvtError("VGGishMFMA: expected 3x3 stride-1 pad-1 convolutions")
            pool = any(isinstance(n, nn.MaxPool2d) for n in mods[i + 1 : i + 3])
            w = m.weight.detach().float().unsqueeze(2)  # [Cout, Cin, 1, 3, 3]
            bias = m.bias.detach().float() if m.bias is not None else torch.zeros(w.shape[0])
            if w.shape[1] == 1:
                wg, rg = group_weights_w(w, 8)
                conv = FusedConv(None, None, True, self.dev, folded=(wg, bias.repeat(8), (1, 1, 1), (0, 1, rg)), x3=x3)
                conv._folded = None  # already grouped
                conv.alg_flops_per_row = 8 * 2.0 * 9 * w.shape[0]
                conv.bins = 8
            else:
                conv = FusedConv(None, None, True, self.dev, folded=(w, bias, (1, 1, 1), (0, 1, 1)), x3=x3)
                conv.bins = 1
            self.layers.append((conv, pool, w.shape[0]))

    @torch.no_grad()
    def forward(self, x):
        """x [n,1,H,W] log-mel examples (H=100 frames, W=64 bins) -> [n, (H/16)*(W/16)*512] fp32."""
        if x.dim() != 4 or x.shape[1] != 1:
            raise AvtError("VGGishMFMA: input must be [n,1,frames,bins], got %s" % (tuple(x.shape),))
        n, _, h, w = x.shape
        if w % 8:
            raise AvtError("VGGishMFMA: the number of mel bins must be a multiple of 8 (got %d)" % w)
        if self.x3 is not None:
            hi, lo = split_planes(x.to(self.dev, torch.float32).contiguous().view(n * h * (w // 8), 8), self.x3)
            act = Act(hi, (n, 1, h, w // 8), lo=lo)
        else:
            act = Act(x.to(self.dev, torch.bfloat16).contiguous().view(n * h * (w // 8), 8), (n, 1, h, w // 8))
        for conv, pool, cout in self.layers:
            if conv.bins > 1:
                y = conv(act)
                act = Act(y.buf.view(-1, cout), (n, 1, h, w), lo=None if y.lo is None else y.lo.view(-1, cout))
            else:
                act = conv(act)
            if pool:
                _, _, h, w = act.dims
                out = new_act(n * (h // 2) * (w // 2), cout, (n, 1, h // 2, w // 2), self.dev, self.x3 is not None)
                if self.x3 is not None:
                    ops.maxpool_hw2s2_x3(act.ptrs, out.ptrs, n, h, w, cout, act.ld, cout, self.x3)
                else:
                    ops.maxpool_hw2s2(act.ptr, out.ptr, n, h, w, cout, act.ld, cout)
                h, w = h // 2, w // 2
                act = out
        return act.float(self.x3).view(n, -1) if self.x3 is not None else act.buf.view(n, -1).float()
