"""End-to-end `validate()` on a synthetic video (the user-facing -e run): N ~ 2048 windows at 128^2, aligned mode, contract-grade
encoders, 30 s of new video at 30 fps, SuperSloMo interpolation on (seeded weights) — wall time by stage."""
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, ".")
import avtex as avt  # noqa: E402
from avtex import synth  # noqa: E402
from avtex.slowfast import SlowFast  # noqa: E402

dev = torch.device("cuda:0")
W, S, fps = 15, 6, 30.0
n_frames = 2048 * S + W + 1
t0 = time.time()
video = synth.structured_video(11, n_frames, 128, 128)
print("video %s built in %.1f s" % (tuple(video.shape), time.time() - t0))
torch.manual_seed(0)
q_mod, t_mod = SlowFast().eval(), SlowFast().eval()
synth.randomise_bn(q_mod, 1, 0.0)
synth.randomise_bn(t_mod, 2, 0.0)
model = avt.ContrastivePredictionTemporal(q_mod, t_mod, None, 1, 128, 0.1, W, S, 0.3, mini_batchsize=100, enc_arch="slowfast",
                                          img_size=224).to(dev).eval()
args = SimpleNamespace(vdata=None, adata=None, dadata=None, subsample_rate=1, fps=fps, stride=S, window=W, enc_arch="slowfast",
                       img_size=224, model_type=1, mini_batchsize=100, threshold=0.3, alpha=0.5, temp=0.1, driving_audio=None,
                       da_feats="VGG", interpolation=True, SF=5, slomo_ckpt="random", new_video_length=30, results_folder="/tmp/avt_e2e",
                       logname="e2e", batch_size=8, stitch_mode="aligned", enc_batch=249, enc_impl="mfma", enc_dtype="fp32", frames_bar=False)
np.random.seed(0)
torch.cuda.synchronize()
t0 = time.time()
import contextlib, io
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    frames = avt.validate(model, args, video_name="synthetic", model_type=1, video=(video, fps))
torch.cuda.synchronize()
wall = time.time() - t0
out = buf.getvalue()
print("validate(): %.1f s wall for %d output frames (%d jumps interpolated); windows encoded line: %s" % (
    wall, len(frames), out.count("Added 4 intermediate frames."), [l for l in out.splitlines() if l.startswith("Windows encoded")]))
