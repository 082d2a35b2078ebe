#!/usr/bin/env python3
"""Generates tests/golden/*.npz by IMPORTING AND RUNNING THE REFERENCE's own Python
(/root/reference, read-only) in the build container.  Fixtures are data only —
inputs and the reference's outputs; no reference source is copied.

Run:  python tools/gen_golden.py          (needs /root/reference; CPU only, ~2-3 min)

Stub recipe: the reference imports packages this image lacks
(torchvision, librosa, slowfast, ...).  They are replaced by empty modules; only
three stubs carry behaviour, all stated here:
  * torchvision.transforms.Compose/ToTensor/Normalize -> x/255 then (x-mean)/std, no resize
    (frames are generated at img_size already);
  * slowfast process_cv2_inputs -> the upstream PySlowFast arithmetic (NUM_FRAMES 32,
    ALPHA 4, mean .45, std .225) WITHOUT a second /255, as validate.py:121 implies.
    PARITY UNPINNED: that function is third-party and not in the reference repo;
  * torch.nn.DataParallel -> a single-process scatter/gather over dim 0 (what DataParallel
    does, minus threads), and Tensor.cuda()/Module.cuda() -> identity.
"""
import contextlib
import io as _io
import math
import os
import re
import sys
import tempfile
import types
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
CVT = os.path.join(REF, "contrastive_video_textures")
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from tiny_encoders import TinyR3D, TinySlowFast, checksum, seeded  # noqa: E402

STATE = SimpleNamespace(n_gpus=1, frame_lut=None, win_log=None, video=None, fps=20.0, wave=None, sr=16000,
                        wave_da=None)


# --------------------------------------------------------------------------- stubs
def process_cv2_inputs(frames, cfg):
    if STATE.win_log is not None and STATE.frame_lut is not None:
        ids = [STATE.frame_lut.get(f.numpy().tobytes(), -1) for f in frames]
        STATE.win_log.append(ids)
    x = (frames - 0.45) / 0.225
    x = x.permute(3, 0, 1, 2)
    fast = torch.index_select(x, 1, torch.linspace(0, x.shape[1] - 1, 32).long())
    slow = torch.index_select(fast, 1, torch.linspace(0, fast.shape[1] - 1, fast.shape[1] // 4).long())
    return [slow.unsqueeze(0), fast.unsqueeze(0)]


class _Compose:
    def __init__(self, ts):
        self.norm = [t for t in ts if isinstance(t, _Normalize)]

    def __call__(self, x):
        x = x.float() / 255
        for n in self.norm:
            x = n(x)
        return x


class _Normalize:
    def __init__(self, mean, std):
        self.mean = torch.tensor(mean).view(-1, 1, 1)
        self.std = torch.tensor(std).view(-1, 1, 1)

    def __call__(self, x):
        return (x - self.mean) / self.std


class _Marker:
    def __init__(self, *a, **k):
        pass


class FakeDP(nn.Module):
    """torch.nn.DataParallel without threads: scatter dim 0 over n_gpus replicas, gather on dim 0."""

    def __init__(self, module):
        super().__init__()
        self.module = module
        self.calls = []

    def cuda(self):
        return self

    def forward(self, *args, **kw):
        if not hasattr(self.module, "q_encoder"):  # da_model: called as da_model.forward(x) inside a replica (Q9)
            return self.module(*args, **kw)

        def cut(v, g):
            if isinstance(v, torch.Tensor):
                return v[g : g + 1]
            if isinstance(v, list):
                return [cut(u, g) for u in v]
            return v

        b = args[1].shape[0] if isinstance(args[1], torch.Tensor) else args[1][0].shape[0]
        outs = []
        for g in range(b):
            outs.append(self.module(*[cut(a, g) for a in args], **{k: cut(v, g) for k, v in kw.items()}))
        res = tuple(torch.cat([o[i] for o in outs], dim=0) for i in range(len(outs[0])))
        self.calls.append([r.detach().clone().numpy() for r in res[:-2]])  # raw logits (and audio logits)
        return res


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    librosa = mod("librosa", load=lambda path, *a, **k: (STATE.wave_da if "driving" in path else STATE.wave, STATE.sr))
    librosa.output = mod("librosa.output", write_wav=lambda *a, **k: None)
    for n in ("resampy", "soundfile", "ipdb", "cv2", "moviepy"):
        mod(n)
    mod("IPython", get_ipython=lambda: None)
    mod("IPython.display")
    mod("imageio", get_reader=None)
    mod("tensorboardX", SummaryWriter=_Marker)
    tv = mod("torchvision")
    tv.transforms = mod("torchvision.transforms", Compose=_Compose, Normalize=_Normalize, ToPILImage=_Marker,
                        Resize=_Marker, ToTensor=_Marker, CenterCrop=_Marker, ColorJitter=_Marker)
    tv.io = mod("torchvision.io", read_video=lambda fn, pts_unit="sec": (STATE.video, None, {"video_fps": STATE.fps}))
    tv.utils = mod("torchvision.utils", make_grid=lambda x: x)
    tv.models = mod("torchvision.models")
    mod("slowfast")
    mod("slowfast.utils")
    mod("slowfast.visualization")
    mod("slowfast.utils.parser", load_config=lambda *a, **k: SimpleNamespace(NUM_GPUS=1, TEST=SimpleNamespace()))
    mod("slowfast.visualization.predictor", ActionPredictor=_Marker)
    mod("slowfast.visualization.utils", process_cv2_inputs=process_cv2_inputs)
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.device_count = lambda: STATE.n_gpus
    torch.nn.DataParallel = FakeDP
    sys.path.insert(0, CVT)


class FakePlt:
    """Captures the rows validate() plots before (validate.py:548) and after (:677) thresholding."""

    def __init__(self):
        self.rows = []

    def figure(self, *a, **k):
        return self

    def add_subplot(self, *a, **k):
        return self

    def imshow(self, arr, **k):
        self.rows.append(np.array(arr[0], np.float32))
        return None

    def colorbar(self, *a, **k):
        return None

    def __getattr__(self, name):
        return lambda *a, **k: None


class FakeTB:
    def __getattr__(self, name):
        return lambda *a, **k: None


# --------------------------------------------------------------------------- helpers
def make_video(seed, n_frames, h, w):
    g = torch.Generator().manual_seed(seed)
    # smooth-ish video so that neighbouring segments are similar (non-trivial transition rows)
    base = torch.rand((n_frames // 8 + 2, h, w, 3), generator=g)
    t = torch.linspace(0, n_frames / 8, n_frames)
    i0 = t.floor().long()
    frac = (t - i0.float()).view(-1, 1, 1, 1)
    vid = (1 - frac) * base[i0] + frac * base[i0 + 1] + 0.05 * torch.rand((n_frames, h, w, 3), generator=g)
    return (vid.clamp(0, 1) * 255).to(torch.uint8)


def make_wave(seed, seconds, sr=16000):
    rng = np.random.RandomState(seed)
    t = np.arange(int(seconds * sr)) / sr
    f = 220 + 200 * np.sin(2 * np.pi * 0.5 * t)
    return (0.3 * np.sin(2 * np.pi * np.cumsum(f) / sr) + 0.05 * rng.randn(len(t))).astype(np.float32)


def base_args(**kw):
    a = SimpleNamespace(
        vdata="/nonexistent", adata="/tmp", dadata="/tmp/driving", subsample_rate=1, fps=20, stride=4, window=10,
        enc_arch="slowfast", img_size=24, size=24, model_type=2, mini_batchsize=10, threshold=0.3, alpha=0.5,
        temp=0.1, driving_audio=None, da_feats="VGG", daf_resume="", interpolation=False, SF=5, vcam=False,
        new_video_length=4, frames_bar=False, results_folder=None, logname="exp", batch_size=24)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


# --------------------------------------------------------------------------- G1
def gen_g1(utils):
    out = {}
    cases = [(200, 10, 20, 4), (57, 6, 10, 4), (31, 100, 20, 4), (120, 7, 15, 6), (20, 3, 20, 4)]
    for i, (n, mbs, W, S) in enumerate(cases):
        x = torch.arange(1, n + 1).view(n, 1).float()
        seg, nv = utils.split_into_overlapping_segments(x, mbs, W, S)
        out["ov%d_args" % i] = np.array([n, mbs, W, S])
        out["ov%d_out" % i] = seg.numpy()[..., 0]
        out["ov%d_nvalid" % i] = np.array(nv)
    for i, (n, mbs) in enumerate([(44, 10), (100, 100), (7, 3), (1, 5)]):
        x = torch.arange(1, n + 1).view(1, n, 1).float()
        b, nv = utils.split_into_batches(x, mbs)
        out["sb%d_args" % i] = np.array([n, mbs])
        out["sb%d_out" % i] = b.numpy()[..., 0]
        out["sb%d_nvalid" % i] = np.array(nv)
    np.savez_compressed(os.path.join(OUT, "g1_split.npz"), **out)
    print("G1 done")


# --------------------------------------------------------------------------- G3 / G4 / G7
def gen_g3_g4(models):
    from models import ContrastivePredictionTemporal as CPT, VGGish

    out = {}
    W, S, mbs, hw = 10, 4, 5, 24
    g = torch.Generator().manual_seed(77)
    # m=1, slowfast plugin, inference branch
    q_enc, t_enc = seeded(TinySlowFast, 11), seeded(TinySlowFast, 12)
    cpt = CPT(q_enc, t_enc, None, 1, 128, temp=0.1, window=W, stride=S, mini_batchsize=mbs, enc_arch="slowfast",
              img_size=hw).eval()
    qwin = torch.rand((W, hw, hw, 3), generator=g)
    chunk = torch.rand((1, mbs * S + W, hw, hw, 3), generator=g)
    qf = [x.squeeze(0) for x in process_cv2_inputs(qwin, None)]  # [C,T,H,W] each (validate.py:333-344 w/o resize)
    qf = [x.unsqueeze(0) for x in qf]
    with torch.no_grad():
        o, q, t = cpt(qf, chunk, is_inference=True, cam_viz=True)
    out.update(m1_qwin=qwin.numpy(), m1_chunk=chunk.numpy(), m1_out=o.numpy(), m1_q=q.numpy(), m1_t=t.numpy(),
               m1_seeds=np.array([11, 12]), m1_cfg=np.array([W, S, mbs, hw]),
               m1_ck=np.array([checksum(q_enc), checksum(t_enc)]))
    # m=1 non-slowfast plugin (TinyR3D + AdaptiveAvgPool3d), inference
    q2, t2 = seeded(TinyR3D, 13), seeded(TinyR3D, 14)
    cpt2 = CPT(q2, t2, None, 1, 128, temp=0.1, window=W, stride=S, mini_batchsize=mbs, enc_arch="resnet18",
               img_size=hw).eval()
    qv = torch.rand((1, W, 3, hw, hw), generator=g)
    tv = torch.rand((1, mbs * S + W, 3, hw, hw), generator=g)
    with torch.no_grad():
        o2, qq, tt = cpt2(qv, tv, is_inference=True, cam_viz=True)
    out.update(r3d_q=qv.numpy(), r3d_t=tv.numpy(), r3d_out=o2.numpy(), r3d_qe=qq.numpy(), r3d_te=tt.numpy(),
               r3d_seeds=np.array([13, 14]))
    # m=2 with the real VGGish (seeded weights) + driving-audio VGG branch
    vgg = seeded(VGGish, 21)
    cpt3 = CPT(seeded(TinySlowFast, 15), seeded(TinySlowFast, 16), vgg, 2, 128, temp=0.1, window=W, stride=S,
               mini_batchsize=mbs, enc_arch="slowfast", img_size=hw).eval()
    q_a = torch.randn((1, 1, 100, 64), generator=g)
    t_a = torch.randn((1, mbs, 1, 100, 64), generator=g)
    d_a = torch.randn((1, 1, 100, 64), generator=g)
    with torch.no_grad():
        o3, oa3, q3, t3 = cpt3(qf, chunk, q_audio_eg=q_a, t_audio_eg=t_a, is_inference=True, driving_audio=d_a,
                               da_model=vgg, da_feats="VGG", cam_viz=True)
        feats = vgg(t_a.view(-1, 1, 100, 64))  # G7: layout
        raw = vgg.features(t_a.view(-1, 1, 100, 64))
    out.update(m2_qa=q_a.numpy(), m2_ta=t_a.numpy(), m2_da=d_a.numpy(), m2_out=o3.numpy(), m2_out_a=oa3.numpy(),
               m2_q=q3.numpy(), m2_t=t3.numpy(), m2_seeds=np.array([15, 16, 21]), m2_vgg_ck=np.array(checksum(vgg)),
               g7_feats=feats.numpy(), g7_raw_nchw=raw.numpy())
    # G4: training branch (b=3, 1+negs=5) with logits, CE, dlogits
    cpt4 = CPT(seeded(TinySlowFast, 17), seeded(TinySlowFast, 18), None, 1, 128, temp=0.1, window=W, stride=S,
               enc_arch="slowfast", img_size=hw).train()
    b, nt = 3, 5
    q_s, q_f = torch.rand((b, 3, 8, hw, hw), generator=g), torch.rand((b, 3, 32, hw, hw), generator=g)
    t_s, t_f = torch.rand((b, nt, 3, 8, hw, hw), generator=g), torch.rand((b, nt, 3, 32, hw, hw), generator=g)
    logits = cpt4([q_s, q_f], [t_s.clone(), t_f.clone()])
    logits.retain_grad()
    loss = nn.CrossEntropyLoss()(logits, torch.zeros(b, dtype=torch.long))
    loss.backward()
    out.update(tr_qs=q_s.numpy(), tr_qf=q_f.numpy(), tr_ts=t_s.numpy(), tr_tf=t_f.numpy(),
               tr_logits=logits.detach().numpy(), tr_loss=np.array(loss.item()), tr_dlogits=logits.grad.numpy(),
               tr_seeds=np.array([17, 18]),
               tr_grad_fc_q=cpt4.q_encoder.fc.weight.grad.numpy(), tr_grad_fc_t=cpt4.t_encoder.fc.weight.grad.numpy())
    np.savez_compressed(os.path.join(OUT, "g3_g4_operator.npz"), **out)
    print("G3/G4/G7 done")


# --------------------------------------------------------------------------- G5 (+G2)
def run_validate(validate_mod, name, args, n_frames, seed_video, seeds_enc, n_gpus, with_da=False, arch_cls=TinySlowFast):
    from models import ContrastivePredictionTemporal as CPT, VGGish

    STATE.n_gpus = n_gpus
    STATE.fps = float(args.fps)
    STATE.video = make_video(seed_video, n_frames, args.img_size, args.img_size)
    STATE.wave = make_wave(seed_video + 1, n_frames / args.fps + 0.5)
    STATE.wave_da = make_wave(seed_video + 2, 6.0)
    fr = STATE.video.float() / 255
    fr = fr[:, :, :, [2, 1, 0]]
    STATE.frame_lut = {f.numpy().tobytes(): i for i, f in enumerate(fr)}
    STATE.frame_lut[torch.zeros_like(fr[0]).numpy().tobytes()] = -1
    STATE.win_log = [] if args.enc_arch == "slowfast" else None
    vgg_seed = seeds_enc[2]
    vgg = seeded(VGGish, vgg_seed)
    model = CPT(seeded(arch_cls, seeds_enc[0]), seeded(arch_cls, seeds_enc[1]), vgg, 2, 128, args.temp, args.window,
                args.stride, args.threshold, mini_batchsize=args.mini_batchsize, enc_arch=args.enc_arch,
                img_size=args.img_size)
    dp = FakeDP(model)
    fake_plt = FakePlt()
    validate_mod.plt = fake_plt
    validate_mod.save_videos = lambda *a, **k: None
    validate_mod.Image = SimpleNamespace(fromarray=lambda a: SimpleNamespace(save=lambda *x, **k: None))
    choices_log = []
    orig_choice = np.random.choice

    def rec_choice(a, *p, **k):
        r = orig_choice(a, *p, **k)
        choices_log.append((np.array(a).copy(), int(r)))
        return r

    if with_da:
        args.driving_audio = ["driving_clip"]
        orig_load = torch.load
        torch.load = lambda *a, **k: seeded(VGGish, vgg_seed).state_dict()
        os.makedirs("/tmp/driving", exist_ok=True)
        open("/tmp/driving/driving_clip.wav", "wb").close()
    open("/tmp/%s.wav" % name, "wb").close()
    tmp = tempfile.mkdtemp()
    args.results_folder = os.path.join(tmp, "results")
    np.random.seed(1234)
    torch.manual_seed(4321)
    np.random.choice = rec_choice
    buf = _io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            try:
                validate_mod.validate(dp, args, video_name=name, tb_logger=FakeTB(), model_type=2, itr=0)
            except UnboundLocalError as e:
                # [quirk Q11] with -nintp the reference dies in its final save_videos() call
                # (validate.py:861-872 reads output_audio_filename_intp, only set when interpolating);
                # the stitch ("Frames list", validate.py:787) is complete by then.
                assert "output_audio_filename_intp" in str(e) or "outfile_intp" in str(e), e
    finally:
        np.random.choice = orig_choice
        if with_da:
            torch.load = orig_load
    text = buf.getvalue()
    frames_list = [int(x) for x in re.findall(r"\d+", text.split("Frames list:")[1].split("\n")[0].replace("np.int64", ""))]
    queries = [int(x) for x in re.findall(r"Query frame:\s+(\d+)", text)]
    chosen = [int(x) for x in re.findall(r"Chosen next frame:\s*(\d+)", text)]
    n_steps = len(chosen)
    L = math.floor((n_frames - args.window) / args.stride)
    # raw logits per step, assembled exactly as validate.py:481-493 does (we re-read them from the plotted row instead)
    pre = fake_plt.rows[0::2][:n_steps]
    post = fake_plt.rows[1::2][:n_steps]
    if STATE.win_log is not None:
        per_step = len(STATE.win_log) // n_steps
        wins = np.array(STATE.win_log, np.int64).reshape(n_steps, per_step, args.window)
    else:
        wins = np.zeros(0, np.int64)
    calls_per_step = len(dp.calls) // n_steps
    raw = []
    for s in range(n_steps):
        cs = dp.calls[s * calls_per_step : (s + 1) * calls_per_step]
        raw.append([np.concatenate([c[i].reshape(-1) for c in cs]) for i in range(len(cs[0]))])
    out = dict(
        video=STATE.video.numpy(), wave=STATE.wave, wave_da=STATE.wave_da if with_da else np.zeros(0, np.float32),
        cfg=np.array([n_frames, args.window, args.stride, args.mini_batchsize, n_gpus, args.img_size, L,
                      args.new_video_length, args.fps, int(with_da)]),
        th_alpha_temp=np.array([args.threshold, args.alpha, args.temp], np.float64),
        seeds=np.array(list(seeds_enc) + [1234, 4321, seed_video]), arch=np.array(args.enc_arch),
        queries=np.array(queries), chosen=np.array(chosen), frames_list=np.array(frames_list),
        rows_pre=np.array(pre, dtype=object), rows_post=np.array(post, dtype=object),
        choices=np.array([c[0] for c in choices_log], dtype=object), rdm=np.array([c[1] for c in choices_log]),
        window_frames=wins,  # [step, 1 + n_calls*G*mbs, W] frame ids each encoder call saw (query first), -1 = padding
        raw_logits=np.array([r[0] for r in raw], dtype=object),
        raw_logits_a=np.array([r[1] for r in raw], dtype=object) if with_da else np.zeros(0),
        enc_ck=np.array([checksum(model.q_encoder), checksum(model.t_encoder), checksum(vgg)]))
    np.savez_compressed(os.path.join(OUT, "g5_validate_%s.npz" % name), **out)
    print("G5 %s done: steps=%d L=%d frames=%d first choices %s" % (name, n_steps, L, len(frames_list), chosen[:6]))


def gen_g5():
    import validate as validate_mod

    run_validate(validate_mod, "sf_th03", base_args(threshold=0.3), 120, 5, (31, 32, 33), 1)
    run_validate(validate_mod, "sf_th00", base_args(threshold=0.0), 120, 5, (31, 32, 33), 1)
    run_validate(validate_mod, "sf_g2", base_args(threshold=0.3, mini_batchsize=6), 131, 6, (34, 35, 36), 2)
    run_validate(validate_mod, "sf_da", base_args(threshold=0.0), 120, 7, (37, 38, 39), 1, with_da=True)
    # the non-SlowFast plugin path (enc_arch != "slowfast": transforms + [B,C,T,H,W] encoders + AdaptiveAvgPool3d)
    run_validate(validate_mod, "r3d_th03", base_args(threshold=0.3, enc_arch="resnet18"), 120, 8, (41, 42, 43), 1,
                 arch_cls=TinyR3D)


# --------------------------------------------------------------------------- G6
def gen_g6():
    from utils import waveform_to_examples

    w = make_wave(99, 3.0)
    ex = waveform_to_examples(w, 16000)
    np.savez_compressed(os.path.join(OUT, "g6_logmel.npz"), wave=w, examples=np.array(ex, np.float64))
    print("G6 done", ex.shape, ex.dtype)


# --------------------------------------------------------------------------- G8 classic
def gen_g8():
    sys.path.insert(0, os.path.join(REF, "baselines", "classic_video_textures"))
    import computeD1
    import computeD2

    g = torch.Generator().manual_seed(7)
    frames = torch.randint(0, 256, (20, 8, 8, 3), generator=g).float()
    d1, p1, sigma1 = computeD1.compute_D1(frames, 0.1, feats="RGB", slow=True, batch_size=7)
    d1b, _, _ = computeD1.compute_D1(frames, 0.1, feats="RGB", slow=False)
    d2, p2, sigma2, filt = computeD2.compute_D2(d1, 0.1, filter_size=4)
    import q_learning as ql  # (same stubbed third-party imports; prints its eps per sweep)

    with contextlib.redirect_stdout(_io.StringIO()):
        d3, p3, p3n, sigma3 = ql.q_learning(d2.clone(), 0.1)
    np.savez_compressed(os.path.join(OUT, "g8_classic.npz"), frames=frames.numpy(), d1=d1.numpy(), p1=p1.numpy(),
                        sigma1=np.array(float(sigma1)), d1_fast=d1b.numpy(), d2=d2.numpy(), p2=p2.numpy(),
                        sigma2=np.array(float(sigma2)), d3=d3.numpy(), p3=p3.numpy(), p3_thresholded=p3n.numpy(),
                        sigma3=np.array(float(sigma3)))
    print("G8 done")


# --------------------------------------------------------------------------- G9 dataset sampling
def gen_g9():
    from dataset import AudioVideoSegments

    STATE.video = make_video(3, 150, 16, 16)
    STATE.fps = 20.0
    args = SimpleNamespace(vdata="/tmp", adata=None, n_negs=10, img_size=16, enc_arch="slowfast", window=0, stride=0)
    open("/tmp/g9.mp4", "wb").close()
    torch.manual_seed(5)
    ds = AudioVideoSegments(args, "g9", split="train")
    rec = {}
    for idx in (0, 1, 3, 10, len(ds) - 1):
        np.random.seed(100 + idx)
        item = ds[idx]
        t_ae = item[5]  # [1+negs, 10] dummy audio rows identify the segment ids
        ids = [int((ds.audio_eg == row).all(dim=1).nonzero()[0, 0]) for row in t_ae]
        rec["idx%d" % idx] = np.array(ids)
    rec["len"] = np.array(len(ds))
    rec["window_stride"] = np.array([args.window, args.stride])
    np.savez_compressed(os.path.join(OUT, "g9_dataset.npz"), **rec)
    print("G9 done", {k: v.tolist() for k, v in rec.items()})


def gen_g10():
    """G10 — SuperSloMo at a jump: the reference's own UNet / backWarp (models/slowmo.py) and interpolate.forward
    (interpolate.py:93-147) on seeded weights (tests/interp_weights.py) and synthetic frame pairs.  torchvision is not
    installed here, so the three transforms around the call are restated from its published behaviour (ToTensor: x / 255;
    Normalize: (x - mean) / std; ToPILImage: x.mul(255).byte()); `interpolate.__init__` is bypassed because it places its
    grid on `device=0` — the members it would create are built here with the reference's classes."""
    from PIL import Image
    import interpolate as ref_intp
    from models import slowmo as ref_slowmo
    from interp_weights import frame_pair, unet_state

    mean = torch.tensor([0.429, 0.431, 0.397]).view(3, 1, 1)  # interpolate.py:51-52
    out = {}
    for name, (h, w, sf, seed) in {"a": (64, 96, 5, 3), "b": (32, 32, 3, 4), "c": (128, 128, 5, 5)}.items():
        fc, at = unet_state(6, 4, 10 + seed, head_gain=20.0), unet_state(20, 5, 20 + seed, head_gain=5.0)
        f0, f1 = frame_pair(seed, h, w)
        m = ref_intp.interpolate.__new__(ref_intp.interpolate)
        nn.Module.__init__(m)
        m.flowComp = ref_slowmo.UNet(6, 4)
        m.flowComp.load_state_dict(fc)
        m.ArbTimeFlowIntrp = ref_slowmo.UNet(20, 5)
        m.ArbTimeFlowIntrp.load_state_dict(at)
        m.SF, m.origDim, m.dim = sf, [w, h], (w, h)
        m.flowBackWarp = ref_slowmo.backWarp(w, h, device="cpu")
        floats = []

        def tp(x):  # revNormalize (mean -> -mean, std 1) + ToPILImage
            y = (x - (-mean)) / torch.ones(3).view(3, 1, 1)
            floats.append(y.clone())
            return Image.fromarray(y.mul(255).byte().permute(1, 2, 0).contiguous().numpy())

        def tt(f):  # ToTensor + Normalize
            return (f.permute(2, 0, 1).float().div(255) - mean) / torch.ones(3).view(3, 1, 1)

        frames = m.forward(tt(f0), tt(f1), tp)
        assert len(frames) == sf - 1
        with torch.no_grad():
            flow = m.flowComp(torch.cat((tt(f0), tt(f1)), 0).unsqueeze(0))
        out[name + "_dims"] = np.array([h, w, sf, seed], np.int64)
        out[name + "_frame0"] = f0.numpy()
        out[name + "_frame1"] = f1.numpy()
        out[name + "_out"] = np.stack([np.array(fr) for fr in frames]).astype(np.uint8)
        if h * w <= 64 * 96:
            out[name + "_float"] = torch.stack(floats).numpy().astype(np.float32)   # [sf-1, 3, H, W] before * 255
        out[name + "_flow"] = flow[0].numpy().astype(np.float32)                # flowComp's output [4, H, W]
        out[name + "_wsum"] = np.array([sum(float(v.double().abs().sum()) for v in fc.values()),
                                        sum(float(v.double().abs().sum()) for v in at.values())])
        print("g10", name, "flow |mean| %.3f max %.3f" % (float(flow.abs().mean()), float(flow.abs().max())),
              "out mean %.2f" % out[name + "_out"].mean())
    np.savez_compressed(os.path.join(OUT, "g10_interp.npz"), **out)


def gen_g11():
    """The reference CLI as data ("every flag in main.py:41-296 with same names, defaults, types"): one record per
    parser.add_argument call of contrastive_video_textures/main.py, read from its syntax tree (the module itself is not
    imported: its parser runs parse_args() at import) -> option strings, dest, default, type name, action, choices, nargs."""
    import ast
    import json

    src = open(os.path.join(REF, "contrastive_video_textures", "main.py")).read()
    rows = []
    for node in ast.walk(ast.parse(src)):
        if not (isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "add_argument"
                and isinstance(node.func.value, ast.Name) and node.func.value.id == "parser"):
            continue
        opts = [ast.literal_eval(a) for a in node.args]
        kw = {}
        for k in node.keywords:
            if k.arg == "type":
                kw["type"] = k.value.id if isinstance(k.value, ast.Name) else ast.unparse(k.value)
            elif k.arg in ("default", "action", "choices", "nargs", "dest", "required", "const"):
                kw[k.arg] = ast.literal_eval(k.value)
        longs = [o for o in opts if o.startswith("--")]
        dest = kw.get("dest") or (longs[0][2:] if longs else opts[0].lstrip("-")).replace("-", "_")
        action = kw.get("action", "store")
        default = kw.get("default", False if action == "store_true" else (True if action == "store_false" else None))
        rows.append({"line": node.lineno, "opts": opts, "dest": dest, "default": default, "type": kw.get("type"), "action": action,
                     "choices": kw.get("choices"), "nargs": kw.get("nargs"), "required": kw.get("required", False)})
    rows.sort(key=lambda r: r["line"])
    assert 41 <= rows[0]["line"] and rows[-1]["line"] <= 296, (rows[0]["line"], rows[-1]["line"])
    np.savez_compressed(os.path.join(OUT, "g11_cli.npz"), flags=np.array([json.dumps(r, sort_keys=True) for r in rows]))
    print("g11: %d add_argument calls, lines %d..%d" % (len(rows), rows[0]["line"], rows[-1]["line"]))


def main():
    os.makedirs(OUT, exist_ok=True)
    install_stubs()
    import models  # noqa: F401  (reference)
    import utils as ref_utils

    which = sys.argv[1:] or ["g1", "g3", "g5", "g6", "g8", "g9", "g10", "g11"]
    with torch.no_grad():
        if "g1" in which:
            gen_g1(ref_utils)
        if "g6" in which:
            gen_g6()
        if "g5" in which:
            gen_g5()
        if "g9" in which:
            gen_g9()
        if "g10" in which:
            gen_g10()
    if "g11" in which:
        gen_g11()
    if "g3" in which:
        gen_g3_g4(models)
    if "g8" in which:
        with torch.no_grad():
            gen_g8()


if __name__ == "__main__":
    main()
