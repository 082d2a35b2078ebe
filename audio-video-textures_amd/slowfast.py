"""SlowFast-8x8-R50 video encoder plugin: forward([slow [B,3,8,H,W], fast [B,3,32,H,W]]) -> [B,2304].

The reference builds this model through the third-party `slowfast` package, which it neither vendors
nor pins (contrastive_video_textures/models/models.py:565-580), then replaces the head's dropout /
projection / act by Identity so the head returns the pooled, concatenated (slow 2048 | fast 256)
features.  This file restates the architecture from the SlowFast paper / model-zoo config
SLOWFAST_8x8_R50 with PySlowFast's module names, so a converted reference
checkpoint (`q_encoder.s1.pathway0_stem.conv.weight`, ...) loads by key.  PARITY UNPINNED: no
reference test or weight file pins its arithmetic.

MI355X notes: this nn.Module is the weight container and the plugin surface.  At `-e` its weights run on the hand-written MFMA
convolutions (fused_slowfast.SlowFastMFMA: BatchNorm folded, split-plane fp16 arithmetic by default); in train mode its
convolutions / BatchNorms / pools run on the hand-written passes of train_ops (channels_last_3d).  Only `--enc_impl module`
and `--train_conv fp32` send it to MIOpen.
"""
import torch
import torch.nn as nn

from .train_ops import bn_act, conv3d, conv3d_fork, join_channels, max_pool_hw

ALPHA, BETA_INV, FUSION_RATIO, FUSION_KERNEL = 4, 8, 2, 7
PATHWAY_STREAMS = 1  # train mode: the fast pathway of a default-stream forward on a side stream (SlowFast._forward_two_streams:
#                      config 5 390 -> 411 clips/s, same losses; profiles/r05/train_pathway_streams_ab.log); tests set 0 for the reference
WIDTH = 64
DEPTHS = (3, 4, 6, 3)
# temporal kernel of conv `a` per stage, (slow, fast): SLOWFAST_8x8: slow 1,1,3,3 / fast 3,3,3,3
TKERNEL = ((1, 3), (1, 3), (3, 3), (3, 3))
STEM_TK = (1, 5)


class Stem(nn.Module):
    def __init__(self, cout, tk):
        super().__init__()
        self.conv = nn.Conv3d(3, cout, (tk, 7, 7), stride=(1, 2, 2), padding=(tk // 2, 3, 3), bias=False)
        self.bn = nn.BatchNorm3d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.pool_layer = nn.MaxPool3d((1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1))

    def forward(self, x, cat_extra=0):
        # (train mode on the GPU: fused HIP passes; cat_extra: the pooled output as the first channels of the lateral fusion's
        #  concatenation buffer, train_ops.join_channels)
        return max_pool_hw(bn_act(conv3d(x, self.conv), self.bn, relu=True), self.pool_layer, cat_extra=cat_extra)


class VideoModelStem(nn.Module):
    def __init__(self):
        super().__init__()
        self.pathway0_stem = Stem(WIDTH, STEM_TK[0])
        self.pathway1_stem = Stem(WIDTH // BETA_INV, STEM_TK[1])

    def forward(self, x, cat_extra=0):
        return [self.pathway0_stem(x[0], cat_extra), self.pathway1_stem(x[1])]


class FuseFastToSlow(nn.Module):
    def __init__(self, c_fast):
        super().__init__()
        self.conv_f2s = nn.Conv3d(c_fast, c_fast * FUSION_RATIO, (FUSION_KERNEL, 1, 1), stride=(ALPHA, 1, 1),
                                  padding=(FUSION_KERNEL // 2, 0, 0), bias=False)
        self.bn = nn.BatchNorm3d(c_fast * FUSION_RATIO)
        self.relu = nn.ReLU(inplace=True)
        self.extra = c_fast * FUSION_RATIO  # channels this fusion appends to the slow pathway

    def forward(self, x):
        # torch.cat([slow, lateral], 1); in train mode on the GPU the slow pathway's producer left room for the lateral channels
        # (cat_extra) and the lateral BatchNorm writes them in place: no copy either way (train_ops.join_channels)
        tag = getattr(x[0], "_avt_cat", None)
        lat = bn_act(conv3d(x[1], self.conv_f2s, stats=self.bn), self.bn, relu=True, cat_into=None if tag is None else (tag[0], x[0].shape[1]))
        return [join_channels(x[0], lat), x[1]]


class BottleneckTransform(nn.Module):
    def __init__(self, cin, cout, cinner, tk, stride):
        super().__init__()
        self.a = nn.Conv3d(cin, cinner, (tk, 1, 1), padding=(tk // 2, 0, 0), bias=False)
        self.a_bn = nn.BatchNorm3d(cinner)
        self.a_relu = nn.ReLU(inplace=True)
        self.b = nn.Conv3d(cinner, cinner, (1, 3, 3), stride=(1, stride, stride), padding=(0, 1, 1), bias=False)
        self.b_bn = nn.BatchNorm3d(cinner)
        self.b_relu = nn.ReLU(inplace=True)
        self.c = nn.Conv3d(cinner, cout, 1, bias=False)
        self.c_bn = nn.BatchNorm3d(cout)

    def forward(self, x):
        x = self.a_relu(self.a_bn(self.a(x)))
        x = self.b_relu(self.b_bn(self.b(x)))
        return self.c_bn(self.c(x))


class ResBlock(nn.Module):
    def __init__(self, cin, cout, cinner, tk, stride):
        super().__init__()
        if cin != cout or stride != 1:
            self.branch1 = nn.Conv3d(cin, cout, 1, stride=(1, stride, stride), bias=False)
            self.branch1_bn = nn.BatchNorm3d(cout)
        self.branch2 = BottleneckTransform(cin, cout, cinner, tk, stride)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x, cat_extra=0):
        t = self.branch2
        # (train mode on the GPU: the shortcut's gradient is summed into a's input gradient inside that kernel, train_ops)
        # (stats=bn: the convolution's epilogue leaves that BatchNorm's batch statistics behind — no statistics pass over its output)
        h, xs = conv3d_fork(x, t.a, stats=t.a_bn)
        sc = bn_act(conv3d(xs, self.branch1, stats=self.branch1_bn), self.branch1_bn, relu=False) if hasattr(self, "branch1") else xs
        h = bn_act(h, t.a_bn, relu=True)
        h = bn_act(conv3d(h, t.b, stats=t.b_bn), t.b_bn, relu=True)
        # c's BatchNorm, the shortcut add and the block's ReLU: one pass in train mode (csrc/bn_train.hip), the stock ops else
        return bn_act(conv3d(h, t.c, stats=t.c_bn), t.c_bn, res=sc, relu=True, cat_extra=cat_extra)


class ResStage(nn.Module):
    def __init__(self, cin, cout, cinner, tks, stride, depth):
        super().__init__()
        self.depth = depth
        for p in range(2):
            for i in range(depth):
                self.add_module("pathway%d_res%d" % (p, i),
                                ResBlock(cin[p] if i == 0 else cout[p], cout[p], cinner[p], tks[p],
                                         stride if i == 0 else 1))

    def pathway(self, p, y, cat_extra=0):
        """The blocks of pathway p (0 slow, 1 fast) of this stage."""
        for i in range(self.depth):
            y = getattr(self, "pathway%d_res%d" % (p, i))(y, cat_extra if (p == 0 and i == self.depth - 1) else 0)
        return y

    def forward(self, x, cat_extra=0):
        """cat_extra: channels the lateral fusion after this stage appends to the slow pathway (its last block writes into the
        concatenation buffer, train_ops.join_channels)."""
        return [self.pathway(0, x[0], cat_extra), self.pathway(1, x[1])]


class Head(nn.Module):
    """ResNetBasicHead after the reference's surgery (models.py:578-580): per-pathway global average
    pool, concat slow|fast, Identity dropout/projection/act -> [B, 2304]."""

    def __init__(self):
        super().__init__()
        self.pathway0_avgpool = nn.AdaptiveAvgPool3d(1)
        self.pathway1_avgpool = nn.AdaptiveAvgPool3d(1)
        self.dropout = nn.Identity()
        self.projection = nn.Identity()
        self.act = nn.Identity()

    def forward(self, x):
        z = torch.cat([self.pathway0_avgpool(x[0]), self.pathway1_avgpool(x[1])], 1)
        z = self.act(self.projection(self.dropout(z.permute(0, 2, 3, 4, 1))))
        return z.reshape(z.shape[0], -1)


class SlowFast(nn.Module):
    out_dim = WIDTH * 32 + WIDTH * 32 // BETA_INV  # 2048 + 256

    def __init__(self):
        super().__init__()
        w, wf = WIDTH, WIDTH // BETA_INV
        self.s1 = VideoModelStem()
        self.s1_fuse = FuseFastToSlow(wf)
        dims = []
        cin = [w + wf * FUSION_RATIO, wf]
        for k in range(4):
            cout = [w * 4 * 2 ** k, wf * 4 * 2 ** k]
            cinner = [w * 2 ** k, wf * 2 ** k]
            dims.append((cin, cout, cinner))
            cin = [cout[0] + cout[1] * FUSION_RATIO, cout[1]]
        self.s2 = ResStage(*dims[0], TKERNEL[0], 1, DEPTHS[0])
        self.s2_fuse = FuseFastToSlow(dims[0][1][1])
        self.s3 = ResStage(*dims[1], TKERNEL[1], 2, DEPTHS[1])
        self.s3_fuse = FuseFastToSlow(dims[1][1][1])
        self.s4 = ResStage(*dims[2], TKERNEL[2], 2, DEPTHS[2])
        self.s4_fuse = FuseFastToSlow(dims[2][1][1])
        self.s5 = ResStage(*dims[3], TKERNEL[3], 2, DEPTHS[3])
        self.head = Head()
        for m in self.modules():
            if isinstance(m, nn.Conv3d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        for m in self.modules():  # PySlowFast zero-initialises the last BN of every residual branch
            if isinstance(m, BottleneckTransform):
                nn.init.zeros_(m.c_bn.weight)

    def forward(self, x):
        if PATHWAY_STREAMS and self.training and x[0].is_cuda and torch.is_grad_enabled():
            cur = torch.cuda.current_stream(x[0].device)
            # the encoder whose forward runs on the STEP's own stream (the target encoder) forks its fast pathway; the one that
            # already runs on one of the package's side streams (the query encoder) does not.  (Round 5 asked for the default stream;
            # a step captured as a HIP graph runs on a capture stream, train_ops.GraphedStep)
            from . import ops

            if all(s.cuda_stream != cur.cuda_stream for lst in ops._SIDE.values() for s in lst):
                return self._forward_two_streams(x, cur)
        x = self.s1_fuse(self.s1(x, self.s1_fuse.extra))
        x = self.s2_fuse(self.s2(x, self.s2_fuse.extra))
        x = self.s3_fuse(self.s3(x, self.s3_fuse.extra))
        x = self.s4_fuse(self.s4(x, self.s4_fuse.extra))
        return self.head(self.s5(x))

    def _forward_two_streams(self, x, cur):
        """The training forward with the FAST pathway on a side stream (round 5 experiment, PATHWAY_STREAMS): the fast pathway depends
        on nothing of the slow one, so its stem and stages run ahead on their own stream; the slow pathway's stream waits for the
        fast features of a stage only where the lateral connection reads them.  autograd replays every backward node on its
        forward's stream and orders the streams itself.  Only for a forward that runs on the STEP's own stream (the target encoder of a
        training step; the query encoder already runs on a side stream of its own — three hardware queues in all): the default stream,
        or the stream a captured step runs on (train_ops.GraphedStep)."""
        from . import models

        dev = x[0].device
        fs = models._side_stream(dev, cur, role=1)  # (role 0 of this step stream is the query encoder's)
        fs.wait_stream(cur)
        stages, fuses = (self.s2, self.s3, self.s4, self.s5), (self.s1_fuse, self.s2_fuse, self.s3_fuse, self.s4_fuse)
        with torch.cuda.stream(fs):
            f = self.s1.pathway1_stem(x[1])
            ev = torch.cuda.Event()
            ev.record(fs)
        s = self.s1.pathway0_stem(x[0], self.s1_fuse.extra)
        for k, stage in enumerate(stages):
            cur.wait_event(ev)           # the fast features the lateral connection reads
            f.record_stream(cur)
            f_in = f
            with torch.cuda.stream(fs):  # the fast pathway's next stage is queued first: it runs under the slow pathway's
                f = stage.pathway(1, f_in)
                ev = torch.cuda.Event()
                ev.record(fs)
            s = fuses[k]([s, f_in])[0]
            s = stage.pathway(0, s, fuses[k + 1].extra if k + 1 < len(fuses) else 0)
        cur.wait_event(ev)
        f.record_stream(cur)
        return self.head([s, f])


def prepare_encoder(module, device, dtype=torch.bfloat16, channels_last=False):
    """Inference placement on MI355X: weights resident on the GPU in `dtype`.  Measured on MIOpen (ROCm 7.2,
    find mode on): bf16 NCDHW 756 clips/s, bf16 channels-last-3d 650, fp32 NCDHW 288 — so NCDHW is the default."""
    module = module.to(device=device, dtype=dtype).eval()
    if channels_last:
        module = module.to(memory_format=torch.channels_last_3d)
    return module
