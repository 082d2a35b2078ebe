"""The operator and the encoder factory, with the reference's names and signatures
(contrastive_video_textures/models/models.py:233-467, 536-584).

`ContrastivePredictionTemporal.forward` keeps the reference contract call-for-call so existing
callers drop in; the arithmetic between "embeddings exist" and "logits" (concat, L2-normalise,
similarity, /temp) runs in the gfx950 kernels behind include/avt.h.  The stitch loop does not call
this per chunk any more — it uses texture.TextureEngine, which encodes every window once — but the
operator stays for callers of `model(...)` and for training.
"""
import os
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops, resnet3d
from ._lib import AvtError
from .slowfast import SlowFast

FAST_T, SLOW_T, ALPHA = 32, 8, 4
SF_MEAN, SF_STD = 0.45, 0.225


def process_cv2_inputs(frames, cfg=None):
    """Float frames [W,H,W,3] (0-1, BGR) -> [slow [1,3,8,H,W], fast [1,3,32,H,W]].

    Stand-in for the third-party slowfast.visualization.utils.process_cv2_inputs the reference calls
    (models.py:365, validate.py:333, dataset.py:145): normalise by 0.45/0.225, THWC->CTHW,
    linspace(0,W-1,32).long() temporal sampling, slow = every 4th via linspace(0,31,8).long().
    PARITY UNPINNED (not in the reference repo).  Device torch ops; the fast path is ops.clip_pack."""
    x = (frames - SF_MEAN) / SF_STD
    x = x.permute(3, 0, 1, 2)
    fast = torch.index_select(x, 1, torch.linspace(0, x.shape[1] - 1, FAST_T).long().to(x.device))
    slow = torch.index_select(fast, 1, torch.linspace(0, fast.shape[1] - 1, fast.shape[1] // ALPHA).long().to(x.device))
    return [slow.unsqueeze(0), fast.unsqueeze(0)]


def similarity_logits(q_v, t_v, temp, q_a=None, t_a=None):
    """q_v [B,Dv], t_v [B,n,Dv] (+ audio parts) -> (logits [B,n], q_hat [B,1,D], t_hat [B,D,n]).

    models.py:347-351, 408-417 through the HIP kernels: two-source l2norm (the torch.cat is folded
    into the normalise), exact-fp32 MFMA similarity, true division by temp."""
    b, n, dv = t_v.shape
    qn, _, _ = ops.l2norm_rows(q_v.float().contiguous(), None if q_a is None else q_a.float().contiguous())
    tn, _, _ = ops.l2norm_rows(t_v.reshape(b * n, dv).float().contiguous(),
                               None if t_a is None else t_a.reshape(b * n, -1).float().contiguous())
    d = qn.shape[1]
    out = torch.empty((b, n), dtype=torch.float32, device=qn.device)
    tn = tn.view(b, n, d)
    for i in range(b):
        ops.sim_gemm_nt(qn[i : i + 1], tn[i], temp, "f32", out=out[i : i + 1])
    return out, qn.unsqueeze(1), tn.permute(0, 2, 1)


class _InfoNCELogits(torch.autograd.Function):
    """Training branch (models.py:385-417): normalise -> bmm -> /temp as ONE fused HIP kernel forward and one
    backward (csrc/infonce.hip).  CPU tensors (unit tests of the module without a GPU) use the stock ops."""

    @staticmethod
    def forward(ctx, q, t, temp):
        q, t = q.contiguous().float(), t.contiguous().float()
        logits, inv_q, inv_t = ops.infonce_fwd(q, t, temp)
        ctx.save_for_backward(q, t, logits, inv_q, inv_t)
        ctx.temp = temp
        return logits

    @staticmethod
    def backward(ctx, g):
        q, t, logits, inv_q, inv_t = ctx.saved_tensors
        dq, dt = ops.infonce_bwd(q, t, logits, g.float(), inv_q, inv_t, ctx.temp)
        return dq, dt, None

    @staticmethod
    def apply_ops(q, t, temp, want_unit=False):
        """-> (logits [B,n], q_hat [B,1,D] | None, t_hat [B,D,n] | None); the unit vectors are only materialised when the
        caller returns them (cam_viz) — the fused kernel never needs them."""
        if not q.is_cuda:
            raise AvtError("ContrastivePredictionTemporal training needs the model on the MI355X (model.cuda()); "
                           "the normalise/bmm/temperature branch runs on the HIP InfoNCE kernels, no CPU fallback")
        out = _InfoNCELogits.apply(q, t, temp)
        qh = th = None
        if want_unit:
            with torch.no_grad():
                qh = F.normalize(q, dim=1).unsqueeze(1)
                th = F.normalize(t, dim=2).permute(0, 2, 1)
        return out, qh, th


class InfoNCECriterion(nn.Module):
    """nn.CrossEntropyLoss(mean) over the [B,1+negs] logits (train.py:129-135) on the HIP softmax-CE
    kernels, with the analytic backward dlogits = (softmax - onehot)/B."""

    class _Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, logits, labels):
            loss, prob = ops.softmax_ce_fwd(logits.contiguous(), labels)
            ctx.save_for_backward(prob, labels)
            return loss.mean()

        @staticmethod
        def backward(ctx, g):
            prob, labels = ctx.saved_tensors
            d = ops.softmax_ce_bwd(prob, labels, scale=1.0 / prob.shape[0])
            return d * g, None

    def forward(self, logits, labels):
        return self._Fn.apply(logits.float(), labels)


_TRAIN_STREAMS = 1  # the query encoder on a side stream next to the target encoder (tests set 0 for the single-stream step)
_SIDE_STREAMS = {}
_SIDE_SKIP = 0  # (tools/experimental/probe_side_stream_index.py: leave the first k process-wide side streams to others)


def _side_stream(device, cur, role=0):
    """The side stream of `role` (0: the query encoder, 1: the target encoder's fast pathway) that belongs to stream `cur` of `device`
    (a training loop that runs its items on several streams — or a step captured as a HIP graph on a stream of its own — gets its own
    set per step stream: the side work of two steps in flight does not queue behind each other, and the roles of ONE step never share
    a stream: round 6 found the query encoder and the fast pathway of a captured step on one stream, 65 ms per step instead of 52).
    Taken from the process-wide list of ops.side_streams — the SAME streams the synthesis engine's two-stream mode uses: which hardware
    queue a torch stream lands on depends on how many streams the process has created before it, and a side stream created after
    bench.py's two-stream bf16 leg shared a queue with the step's own stream (the query encoder no longer overlapped the target
    encoder: the config-5 leg ran 3.5 % slower inside the default bench run than alone, profiles/r05/trainleg_order.log)."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device(), cur.cuda_stream, int(role))
    s = _SIDE_STREAMS.get(key)
    if s is None:
        taken = {v.cuda_stream for v in _SIDE_STREAMS.values()} | {cur.cuda_stream}
        taken |= {v.cuda_stream for v in ops.side_streams(device, _SIDE_SKIP)} if _SIDE_SKIP else set()
        n = 1
        while s is None:
            for cand in ops.side_streams(device, n):
                if cand.cuda_stream not in taken:
                    s = cand
                    break
            n += 1
        _SIDE_STREAMS[key] = s
    return s


def training_side_streams(device):
    """Every side stream a training forward may have used on `device` (main.wrap_ddp's gradient hook waits for them: a bucket can
    hold gradients written on the query encoder's stream, on the target encoder's fast-pathway stream and on the step's own):
    the whole process-wide list of ops.side_streams."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return list(ops._SIDE.get((device.type, idx), []))


class ContrastivePredictionTemporal(nn.Module):
    def __init__(self, q_image_enc_model, t_image_enc_model, audio_enc_model, model_type, fc_dim, temp=0.1,
                 window=20, stride=2, threshold=0.20, mini_batchsize=20, dropout=0.5, enc_arch="resnet",
                 img_size=224):
        super().__init__()
        if enc_arch != "slowfast":  # models.py:252-260
            self.q_encoder = nn.Sequential(q_image_enc_model, nn.AdaptiveAvgPool3d((1, 1, 1)))
            self.t_encoder = nn.Sequential(t_image_enc_model, nn.AdaptiveAvgPool3d((1, 1, 1)))
        else:
            self.q_encoder = q_image_enc_model
            self.t_encoder = t_image_enc_model
        if model_type == 2:  # models.py:265-284: one shared audio encoder, two never-called MLPs
            self.q_a_encoder = audio_enc_model
            self.q_a_mlp = self._mlp()
            self.t_a_encoder = audio_enc_model
            self.t_a_mlp = self._mlp()
        self.temp = temp
        self.fc_dim = fc_dim
        self.window = window
        self.stride = stride
        self.threshold = threshold
        self.mini_batchsize = mini_batchsize
        self.model_type = model_type
        self.enc_arch = enc_arch
        self.img_size = img_size
        self.cfg = SimpleNamespace(NUM_GPUS=1)
        self.criterion = nn.CrossEntropyLoss()

    @staticmethod
    def _mlp():
        # kept so reference checkpoints (q_a_mlp.*, t_a_mlp.*) load by key; never applied (models.py:267-284)
        return nn.Sequential(nn.Linear(512 * 48, 4096), nn.ReLU(inplace=True), nn.Linear(4096, 4096),
                             nn.ReLU(inplace=True), nn.Linear(4096, 128), nn.ReLU(inplace=True))

    # -- helpers -----------------------------------------------------------------
    def _enc_dtype(self, enc):
        p = next(enc.parameters(), None)
        return p.dtype if p is not None else torch.float32

    def _run_enc(self, enc, x):
        dt = self._enc_dtype(enc)
        x = [v.to(dt) for v in x] if isinstance(x, (list, tuple)) else x.to(dt)
        return enc(x).float()

    def forward(self, q_f, t_f, q_audio_eg=None, t_audio_eg=None, is_inference=False, driving_audio=None,
                da_model=None, da_feats=None, cam_viz=False):
        slowfast = self.enc_arch == "slowfast"
        if slowfast:
            C = q_f[0].shape[1]
            H, W = q_f[0].shape[3], q_f[0].shape[4]
            batch_size = q_f[0].shape[0]
        else:
            C = q_f.shape[2]
            H, W = q_f.shape[3], q_f.shape[4]
            batch_size = q_f.shape[0]
            q_f = q_f.permute(0, 2, 1, 3, 4).contiguous().view(-1, C, self.window, H, W)
        # Training on the device: the query encoder (ONE clip per item: launches of a few workgroups each) runs on a side
        # stream next to the target encoder's 15 clips — the two are independent until the logits, and autograd replays each
        # backward node on its forward's stream, so the backward passes overlap the same way (AVT_TRAIN_STREAMS=0: one stream)
        q_side = None
        if self.training and _TRAIN_STREAMS and slowfast and q_f[0].is_cuda and self.model_type != 2:
            cur = torch.cuda.current_stream(q_f[0].device)
            q_side = _side_stream(q_f[0].device, cur)
            q_side.wait_stream(cur)
            with torch.cuda.stream(q_side):
                q_v = self._run_enc(self.q_encoder, q_f).view(batch_size, -1)
        else:
            q_v = self._run_enc(self.q_encoder, q_f).view(batch_size, -1)

        q_a = t_a = None
        if self.model_type == 2:
            A_c, A_w, A_h = t_audio_eg.shape[2], t_audio_eg.shape[3], t_audio_eg.shape[4]
            q_a = self._run_enc(self.q_a_encoder, q_audio_eg.contiguous().view(-1, A_c, A_w, A_h)).view(batch_size, -1)

        if not self.training:  # models.py:355-383: re-window the chunk of replica row 0 at stride S
            dev = q_v.device
            wins = [t_f[0, i * self.stride : i * self.stride + self.window].to(dev) for i in range(self.mini_batchsize)]
            if not slowfast:
                t_f = torch.stack(wins).unsqueeze(0)
            else:
                packs = [[F.interpolate(item.squeeze(0), size=(self.img_size, self.img_size), mode="bilinear")
                          for item in process_cv2_inputs(w, self.cfg)] for w in wins]
                t_f = [torch.stack([p[k] for p in packs]).unsqueeze(0) for k in range(2)]

        if not slowfast:
            t_len = t_f.shape[1]
            t_f = t_f.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, C, self.window, H, W)
        else:
            t_len = t_f[0].shape[1]
            t_f = [t_f[0].reshape(-1, C, SLOW_T, t_f[0].shape[-2], t_f[0].shape[-1]),
                   t_f[1].reshape(-1, C, FAST_T, t_f[1].shape[-2], t_f[1].shape[-1])]
        t_v = self._run_enc(self.t_encoder, t_f).view(batch_size, t_len, -1)
        if q_side is not None:
            cur.wait_stream(q_side)
            q_v.record_stream(cur)  # (allocated on the side stream, consumed and freed from this one)
        if self.model_type == 2:
            t_a = self._run_enc(self.t_a_encoder, t_audio_eg.contiguous().view(-1, A_c, A_w, A_h)).view(
                batch_size, t_len, -1)

        if self.training:
            qc = q_v if q_a is None else torch.cat((q_v, q_a), dim=1)
            tc = t_v if t_a is None else torch.cat((t_v, t_a), dim=2)
            output, q, t = _InfoNCELogits.apply_ops(qc, tc, self.temp, want_unit=cam_viz)
        else:
            if not q_v.is_cuda:
                raise AvtError("ContrastivePredictionTemporal inference needs the model on the MI355X "
                               "(model.cuda()); the similarity path has no CPU fallback")
            output, q, t = similarity_logits(q_v, t_v, self.temp, q_a, t_a)

        if driving_audio is not None:
            A_c, A_w, A_h = t_audio_eg.shape[2], t_audio_eg.shape[3], t_audio_eg.shape[4]
            if da_feats == "VGG":  # models.py:424-439
                s_a = da_model.forward(t_audio_eg.contiguous().view(-1, A_c, A_w, A_h)).float().view(batch_size, t_len, -1)
                d_a = da_model.forward(driving_audio.contiguous().view(-1, A_c, A_w, A_h)).float().view(batch_size, -1)
            elif da_feats == "Contrastive":
                raise NotImplementedError("da_feats='Contrastive' needs the separately trained VideoForAudio "
                                          "checkpoint (validate.py:268-294); outside the hot-path scope")
            else:  # models.py:445-455: raw log-mel features
                s_a = t_audio_eg.contiguous().view(batch_size, t_audio_eg.shape[1], -1).float()
                d_a = driving_audio.contiguous().view(batch_size, -1).float()
            output_a, _, _ = similarity_logits(d_a.to(output.device), s_a.to(output.device), self.temp)
            output_a = output_a.unsqueeze(1)  # the reference keeps bmm's middle dim here (models.py:439)
            return (output, output_a, q, t) if cam_viz else (output, output_a)
        return (output, q, t) if cam_viz else output


class ModelBuilder3D(object):
    """Encoder factory (models.py:536-584).  `register` adds plugin encoders under new arch names."""

    _plugins = {}

    def __init__(self):
        pass

    @classmethod
    def register(cls, arch, factory):
        """factory(img_size, window, pretrained) -> nn.Module following the plugin contract."""
        cls._plugins[arch] = factory

    @staticmethod
    def build_network(arch="resnet18", img_size=224, window=20, pretrained=True):
        if arch in ModelBuilder3D._plugins:
            return ModelBuilder3D._plugins[arch](img_size, window, pretrained), 128
        assert arch in ["resnet10", "resnet18", "resnet34", "resnet50", "resnext50", "resnext101", "resnext152",
                        "densenet121", "slowfast"]
        if "resnet" in arch:
            model = resnet3d.build(arch, img_size, window)
        elif "slowfast" in arch:
            model = SlowFast()
        else:
            raise Exception("Architecture {} undefined: the reference's own ResNeXt/DenseNet factories reject the "
                            "arguments ModelBuilder3D passes (models.py:557-564)".format(arch))
        if pretrained:
            # the reference reads Kinetics weights from absolute paths outside its repo (models.py:568-573: the caffe2 pickle
            # SLOWFAST_8x8_R50.pkl through PySlowFast); here the path is named by an environment variable, or found at the
            # reference's own relative location, and both the caffe2 .pkl and a torch state dict are accepted
            path = os.environ.get("AVT_PRETRAINED_" + arch.upper())
            if not path and "slowfast" in arch and os.path.isfile(os.path.join("pretrained", "SLOWFAST_8x8_R50.pkl")):
                path = os.path.join("pretrained", "SLOWFAST_8x8_R50.pkl")
            if path and os.path.isfile(path):
                if "slowfast" in arch:
                    from .checkpoint import load_kinetics_slowfast

                    dropped = load_kinetics_slowfast(model, path, strict=True)
                    print("ModelBuilder3D: {} loaded into SlowFast ({} blobs without a place dropped)".format(path, len(dropped)))
                else:
                    sd = torch.load(path, map_location="cpu")
                    model.load_state_dict(sd.get("state_dict", sd), strict=False)
            else:
                print("ModelBuilder3D: no pretrained weights for '{}' (set AVT_PRETRAINED_{}); random init".format(
                    arch, arch.upper()))
        return model, 128  # fc_dim is hard-coded 128 in the reference for every arch (models.py:584)
