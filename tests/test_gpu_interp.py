"""SuperSloMo interpolation at jumps on the HIP kernels (avtex.slowmo: conv_x3 with the LeakyReLU epilogue + csrc/interp.hip)
against the oracle (oracle/interp_ref.py) and G10, the reference's own outputs (interpolate.py:93-147, models/slowmo.py)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from interp_weights import frame_pair, unet_state
from oracle import interp_ref

pytestmark = pytest.mark.gpu
G10 = np.load(os.path.join(os.path.dirname(__file__), "golden", "g10_interp.npz"))
DEV = "cuda:0"


def _planes(x_nhwc):
    """fp32 [M, C] -> Act with fp16 plane pair"""
    from avtex import ops
    from avtex.fused_slowfast import Act, split_planes
    hi, lo = split_planes(x_nhwc, ops.X3_F16)
    return hi.to(DEV), lo.to(DEV)


def _act(x):  # x [B, C, H, W] fp32 cpu -> Act
    from avtex.fused_slowfast import Act
    b, c, h, w = x.shape
    hi, lo = _planes(x.permute(0, 2, 3, 1).reshape(-1, c))
    return Act(hi, (b, 1, h, w), lo=lo)


def _nchw(act, c=None):
    from avtex import ops
    b, _, h, w = act.dims
    y = act.float(ops.X3_F16).cpu().reshape(b, h, w, -1).permute(0, 3, 1, 2)
    return y if c is None else y[:, :c]


def test_pool_and_upsample_passes_match_torch():
    from avtex import ops
    from avtex.fused_slowfast import Act, new_act
    torch.manual_seed(0)
    x = torch.randn(3, 16, 8, 12)
    a = _act(x)
    y = new_act(3 * 4 * 6, 16, (3, 1, 4, 6), DEV, True)
    ops.avgpool2_x3(a.ptrs, (3, 8, 12), 16, a.ld, y.ptrs, y.ld, ops.X3_F16)
    xin = _nchw(a)  # what the planes hold (2^-22 from x)
    assert float((_nchw(y) - F.avg_pool2d(xin, 2)).abs().max()) < 1e-6
    # upsample into the first half of a wider buffer, as the UNet's concat does
    cat = new_act(3 * 16 * 24, 32, (3, 1, 16, 24), DEV, True)
    cat.buf.zero_()
    cat.lo.zero_()
    half = Act(cat.buf, cat.dims, c0=0, C=16, lo=cat.lo)
    ops.upsample2_bilinear_x3(a.ptrs, (3, 8, 12), 16, a.ld, half.ptrs, half.ld, ops.X3_F16)
    up = _nchw(cat)
    assert float((up[:, :16] - F.interpolate(xin, scale_factor=2, mode="bilinear", align_corners=False)).abs().max()) < 1e-6
    assert float(up[:, 16:].abs().max()) == 0.0


@pytest.mark.parametrize("k,cin,cout", [(7, 8, 32), (5, 32, 64), (3, 64, 32)])
def test_leaky_convolution_matches_torch(k, cin, cout):
    from avtex import slowmo
    torch.manual_seed(k)
    conv = torch.nn.Conv2d(cin, cout, k, padding=(k - 1) // 2)
    x = torch.randn(2, cin, 32, 32)
    fc = slowmo._conv(conv, DEV)
    y = _nchw(fc(_act(x)))
    with torch.no_grad():
        exp = F.leaky_relu(conv(x), negative_slope=0.1)
    assert float((y - exp).abs().max()) < 2e-5 * float(exp.abs().max())
    assert float(exp.min()) < 0  # the negative branch is exercised


def test_unet_matches_oracle():
    from avtex import slowmo
    sd = unet_state(6, 4, 13, head_gain=20.0)
    net = slowmo.UNet(6, 4)
    net.load_state_dict(sd)
    torch.manual_seed(1)
    x = torch.randn(2, 6, 64, 96) * 0.3
    xp = torch.cat((x, torch.zeros(2, 2, 64, 96)), 1)
    y = _nchw(slowmo.UNetX3(net, DEV).forward(_act(xp)), 4)
    exp = interp_ref.unet(sd, x)
    err = float((y - exp).abs().max()) / float(exp.abs().max())
    assert err < 1e-4, err


@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_interpolated_frames_match_reference(name):
    """uint8 frames against the reference's (G10).  The convolutions carry a 2^-22 split instead of fp32's 2^-24 and sum in
    another order, so a pixel whose value * 255 lies within ~1e-3 of an integer may land on the other side of the
    truncation: at most one grey level, on a small fraction of the pixels."""
    from avtex import slowmo
    h, w, sf, seed = (int(v) for v in G10[name + "_dims"])
    fc, at = unet_state(6, 4, 10 + seed, head_gain=20.0), unet_state(20, 5, 20 + seed, head_gain=5.0)
    f0, f1 = frame_pair(seed, h, w)
    it = slowmo.Interpolator(h, w, sf, DEV)
    it.flow_comp.load_state_dict(fc)
    it.arb_time.load_state_dict(at)
    out = it(f0.to(DEV), f1.to(DEV))
    torch.cuda.synchronize()
    exp = G10[name + "_out"]
    assert tuple(out.shape) == exp.shape and out.dtype == torch.uint8
    d = np.abs(out.cpu().numpy().astype(np.int32) - exp.astype(np.int32))
    frac = float((d > 0).mean())
    print("g10 %s: pixels differing %.5f, max |diff| %d" % (name, frac, int(d.max())))
    assert int(d.max()) <= 1
    assert frac < 1e-3, frac   # measured 1.1e-4 (a, c), 0 (b)


def test_interpolator_is_deterministic_and_handles_other_sizes():
    """A frame size that is not a multiple of 32 goes through the PIL resize of interpolate.py:43, 137."""
    from avtex import slowmo
    f0, f1 = frame_pair(9, 72, 100)
    it = slowmo.Interpolator(72, 100, 3, DEV)
    it.flow_comp.load_state_dict(unet_state(6, 4, 31, head_gain=20.0))
    it.arb_time.load_state_dict(unet_state(20, 5, 32, head_gain=5.0))
    a = it(f0.to(DEV), f1.to(DEV))
    b = it(f0.to(DEV), f1.to(DEV))
    assert tuple(a.shape) == (2, 72, 100, 3) and torch.equal(a, b)


def test_validate_writes_the_interpolated_video(tmp_path, capsys):
    """`main.py -e` with interpolation on (the reference's default, main.py:95): the Frames list is the one of the plain
    run, and the second video holds every source frame (SF + 1) / 2 times with SF - 1 new frames at each jump
    (validate.py:588-650, 809-872)."""
    from types import SimpleNamespace

    import avtex as avt
    from avtex.slowfast import SlowFast
    torch.manual_seed(0)
    W, S, L = 20, 4, 14
    g = torch.Generator().manual_seed(3)
    video = torch.randint(0, 256, (L * S + W + 1, 64, 64, 3), generator=g, dtype=torch.uint8)
    q_mod, t_mod = SlowFast().eval(), SlowFast().eval()
    with torch.no_grad():
        for m in list(q_mod.modules()) + list(t_mod.modules()):
            if isinstance(m, torch.nn.BatchNorm3d):
                m.weight.uniform_(0.5, 1.0)
    model = avt.ContrastivePredictionTemporal(q_mod, t_mod, None, 1, 128, 0.1, W, S, 0.3, mini_batchsize=8,
                                              enc_arch="slowfast", img_size=224).to(DEV).eval()

    def run(interpolation, folder):
        args = SimpleNamespace(vdata=None, adata=None, dadata=None, subsample_rate=1, fps=4, stride=S, window=W,
                               enc_arch="slowfast", img_size=224, model_type=1, mini_batchsize=8, threshold=0.3, alpha=0.5,
                               temp=0.1, driving_audio=None, da_feats="VGG", interpolation=interpolation, new_video_length=12,
                               results_folder=folder, logname="exp", batch_size=24, stitch_mode="aligned", enc_batch=8,
                               enc_impl="mfma", enc_dtype="bf16", SF=5, slomo_ckpt="random")
        np.random.seed(7)
        return avt.validate(model, args, video_name="x", model_type=1, video=(video, 4.0))

    plain = run(False, None)
    out = capsys.readouterr().out
    frames = run(True, str(tmp_path))
    out = capsys.readouterr().out
    assert frames == plain
    jumps = out.count("Added 4 intermediate frames.")
    assert jumps >= 1 and "Saving Interpolated Video." in out
    import glob
    import shutil
    if shutil.which("ffmpeg") is None:  # the writer's lossless fallback: <name>.npz (video, fps)
        files = glob.glob(os.path.join(str(tmp_path), "*_intp_True_*_SF_5", "*.npz"))
        assert len(files) == 1
        z = np.load(files[0])
        assert float(z["fps"]) == 3 * 4.0
        v = z["video"]
        assert v.shape[0] == 3 * len(frames) and v.shape[1:] == (64, 64, 3)
        # frames that are copies of source frames: all but 4 per jump
        src = {video[i].numpy().tobytes() for i in set(frames)}
        n_new = sum(1 for f in v if f.tobytes() not in src)
        assert n_new == 4 * jumps
