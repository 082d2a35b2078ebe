"""The drop-in entry points on the MI355X: `main.py -e` (CLI -> main() -> validate() with the real SlowFast on the MFMA
convolutions) and one epoch of `train()` (dataset -> operator training branch -> HIP InfoNCE criterion -> SGD)."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tiny_encoders import TinySlowFast, seeded  # noqa: E402

pytestmark = pytest.mark.gpu


def _video(n=70, hw=48, seed=5):
    g = torch.Generator().manual_seed(seed)
    base = torch.rand((n // 6 + 2, hw, hw, 3), generator=g)
    t = torch.linspace(0, n / 6, n)
    i0 = t.floor().long()
    fr = (t - i0.float()).view(-1, 1, 1, 1)
    return (((1 - fr) * base[i0] + fr * base[i0 + 1]).clamp(0, 1) * 255).to(torch.uint8)


def test_cli_evaluate_end_to_end(avt, dev, tmp_path, capsys, monkeypatch):
    """python main.py -vdata DIR -vl clip -ea slowfast -e --resume CKPT ...: reference flags, .npz media."""
    from avtex.main import cli
    from avtex.slowfast import SlowFast

    vdir = tmp_path / "videos"
    vdir.mkdir()
    np.savez(vdir / "clip.npz", video=_video().numpy(), fps=10.0)
    torch.manual_seed(0)
    model = avt.ContrastivePredictionTemporal(SlowFast(), SlowFast(), avt.VGGish(), 1, 128, enc_arch="slowfast")
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm3d):
                m.weight.uniform_(0.5, 1.0)
    ckpt = tmp_path / "ckpt.pth.tar"
    torch.save({"epoch": 3, "arch": "slowfast", "state_dict": model.state_dict(), "best_loss": 0.5}, ckpt)
    monkeypatch.chdir(tmp_path)
    np.random.seed(11)
    cli(["-vdata", str(vdir), "-vl", "clip", "-ea", "slowfast", "-m", "1", "-e", "-nintp", "-th", "0.3", "-temp", "0.1",
         "-mbs", "6", "-nvl", "3", "--resume", str(ckpt), "--stitch_mode", "compat", "--ref_num_gpus", "1",
         "--enc_batch", "8", "--logdir", str(tmp_path / "logs")])
    out = capsys.readouterr().out
    assert "Stride 2 Window 5" in out  # fps 10 -> W = ceil(10/2), S = ceil(10/5): the reference's override (Q10)
    assert "=> loaded checkpoint" in out and "Frames list: " in out
    frames = [int(x) for x in out.split("Frames list: ")[1].split("]")[0].strip(" [").split(",")]
    assert len(frames) >= 30 and max(frames) < 70 and frames[:5] == list(range(frames[0], frames[0] + 5))


def test_train_one_epoch(avt, dev):
    """config 5 in miniature: AudioVideoSegments -> DataLoader -> train(): finite, decreasing InfoNCE loss."""
    args = SimpleNamespace(vdata="/tmp", adata=None, n_negs=10, img_size=32, enc_arch="slowfast", window=0, stride=0,
                           print_freq=100, log_freq=100)
    torch.manual_seed(1)
    ds = avt.AudioVideoSegments(args, "x", split="train", video=(_video(90, 32), 10.0))
    assert (args.window, args.stride) == (5, 2)
    loader = torch.utils.data.DataLoader(ds, batch_size=4, shuffle=True, num_workers=0, drop_last=True)
    model = avt.ContrastivePredictionTemporal(seeded(TinySlowFast, 1), seeded(TinySlowFast, 2), None, 1, 128, temp=0.1,
                                              window=5, stride=2, enc_arch="slowfast", img_size=32).to(dev)
    opt = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9)
    np.random.seed(0)
    losses = [avt.train(loader, model, opt, args, epoch) for epoch in range(3)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    assert losses[0] < np.log(11) * 1.5  # starts near log(1 + negs); n_negs >= 8 as in the reference (dataset.py:190 needs room for the hard negatives)


def test_train_one_epoch_as_a_replayed_graph(avt, dev):
    """`--train_graph 1`: train() captures the device side of a step once per batch shape (train_ops.GraphedStep) and replays it — the
    same loop, the same meters; the loss falls like the eager loop's (the first batch of a shape also serves the two warm-up steps)."""
    args = SimpleNamespace(vdata="/tmp", adata=None, n_negs=10, img_size=32, enc_arch="slowfast", window=0, stride=0,
                           print_freq=100, log_freq=100, train_graph=1)
    torch.manual_seed(1)
    ds = avt.AudioVideoSegments(args, "x", split="train", video=(_video(90, 32), 10.0))
    loader = torch.utils.data.DataLoader(ds, batch_size=4, shuffle=True, num_workers=0, drop_last=True)
    model = avt.ContrastivePredictionTemporal(seeded(TinySlowFast, 1), seeded(TinySlowFast, 2), None, 1, 128, temp=0.1,
                                              window=5, stride=2, enc_arch="slowfast", img_size=32).to(dev)
    opt = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9)
    np.random.seed(0)
    losses = [avt.train(loader, model, opt, args, epoch) for epoch in range(3)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0] and losses[0] < np.log(11) * 1.5


def test_validate_m2_driving_audio_on_mfma_encoders(avt, dev, capsys):
    """Config 3 wiring with the production encoders: real SlowFast x2 AND VGGish run on the hand-written MFMA
    convolutions (validate.py swaps both in), source + driving audio tables are built once, aligned N x N mode."""
    from avtex.fused_vggish import VGGishMFMA
    from avtex.slowfast import SlowFast
    from avtex.texture import TextureEngine
    from avtex.audio_frontend import waveform_to_examples

    torch.manual_seed(0)
    model = avt.ContrastivePredictionTemporal(SlowFast(), SlowFast(), avt.VGGish(), 2, 128, temp=0.1, window=5, stride=2,
                                              threshold=0.3, mini_batchsize=8, enc_arch="slowfast", img_size=64)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm3d):
                m.weight.uniform_(0.5, 1.0)
    model = model.to(dev).eval()
    rng = np.random.default_rng(3)
    wave, wave_da = (0.1 * rng.standard_normal(7 * 16000)).astype(np.float32), (0.1 * rng.standard_normal(3 * 16000)).astype(np.float32)
    args = SimpleNamespace(vdata=None, adata=None, dadata=None, subsample_rate=1, fps=10, stride=2, window=5,
                           enc_arch="slowfast", img_size=64, model_type=2, mini_batchsize=8, threshold=0.3, alpha=0.5,
                           temp=0.1, driving_audio=None, da_feats="VGG", interpolation=False, new_video_length=2,
                           results_folder=None, logname="exp", batch_size=8, stitch_mode="aligned", ref_num_gpus=1,
                           enc_batch=8, enc_impl="auto", enc_dtype="bf16")
    np.random.seed(5)
    frames = avt.validate(model, args, video_name="x", model_type=2, video=(_video(70, 48).numpy(), 10.0),
                          audio=(wave, 16000), driving_audio=(wave_da, 16000))
    out = capsys.readouterr().out
    assert "Frames list: " in out and len(frames) >= 10 and 0 <= min(frames) and max(frames) < 70
    # the audio tables of the engine are the MFMA VGGish's, within bf16 of the fp32 module's
    eg = torch.from_numpy(np.asarray(waveform_to_examples(wave, 16000), dtype=np.float32)).unsqueeze(1)
    fused = VGGishMFMA(model.t_a_encoder, dev)
    tiny = seeded(TinySlowFast, 1).to(dev)
    eng = TextureEngine(tiny, tiny, fused, window=5, stride=2, temp=0.1, img_size=64, model_type=2, device=dev)
    eng.N = 32
    eng.set_audio(eg)
    with torch.no_grad():
        ref = model.t_a_encoder(eg[:32].to(dev))
    assert eng.A.shape == (32, 12288)
    assert torch.nn.functional.cosine_similarity(eng.A, ref, dim=1).min() > 0.999


def test_four_streams_unjoined_equals_one_stream(avt, dev):
    """run_encoders(join=False) with n_streams=4 (clip batch split in halves): the halves are concatenated on an encoder
    stream, not on the caller's un-joined stream — tables bit-identical to the single-stream run."""
    from avtex.fused_slowfast import SlowFastMFMA
    from avtex.slowfast import SlowFast
    from avtex.texture import TextureEngine

    torch.manual_seed(0)
    q, t = SlowFastMFMA(SlowFast(), dev), SlowFastMFMA(SlowFast(), dev)
    video = _video(20 + 4 * 24, 64, seed=2)
    tabs = []
    for n_streams in (1, 4):
        eng = TextureEngine(q, t, None, window=20, stride=4, temp=0.1, img_size=224, model_type=1, device=dev, enc_batch=6)
        eng.n_streams = n_streams
        eng.set_video(video)
        tabs.append([x.clone() for x in eng.build_tables()])
        torch.cuda.synchronize()
    assert torch.equal(tabs[0][0], tabs[1][0]) and torch.equal(tabs[0][1], tabs[1][1])


def test_ddp_training_with_unused_audio_mlps(avt, dev, tmp_path):
    """model_type 2 under DistributedDataParallel (one-rank RCCL group): q_a_mlp / t_a_mlp are never called
    (models.py:267-284), nor is VGGish's fc stack (vggish.py:45): main.wrap_ddp freezes them so DDP's reducer does not stall."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import avtex
from tiny_encoders import TinySlowFast, seeded
from avtex import dist as adist
rank, world, local = adist.init_from_env()
dev = torch.device("cuda", local)
model = avtex.ContrastivePredictionTemporal(seeded(TinySlowFast, 1), seeded(TinySlowFast, 2), avtex.VGGish(), 2, 128,
                                            temp=0.1, window=5, stride=2, enc_arch="slowfast", img_size=32).to(dev)
from avtex.main import wrap_ddp
model = wrap_ddp(model, dev, local)
opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=0.01)
crit = avtex.InfoNCECriterion()
g = torch.Generator().manual_seed(0)
for step in range(3):
    qf = [torch.randn(2, 3, 8, 32, 32, generator=g).to(dev), torch.randn(2, 3, 32, 32, 32, generator=g).to(dev)]
    tf = [torch.randn(2, 4, 3, 8, 32, 32, generator=g).to(dev), torch.randn(2, 4, 3, 32, 32, 32, generator=g).to(dev)]
    qa, ta = torch.randn(2, 1, 100, 64, generator=g).to(dev), torch.randn(2, 4, 1, 100, 64, generator=g).to(dev)
    model.train()
    out = model(qf, tf, q_audio_eg=qa, t_audio_eg=ta)
    loss = crit(out, torch.zeros(2, dtype=torch.long, device=dev))
    opt.zero_grad(); loss.backward(); opt.step()
    assert torch.isfinite(loss)
print("DDP_OK", float(loss))
''' % (root, root)
    env = dict(os.environ, AVT_FORCE_PG="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", HSA_ENABLE_IPC_MODE_LEGACY="0",
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DDP_OK" in r.stdout, r.stderr[-3000:]


def test_ddp_gradients_with_the_query_encoder_on_its_side_stream(avt, dev):
    """Training under DistributedDataParallel with the query encoder on a side stream (models.ContrastivePredictionTemporal.
    forward): main.wrap_ddp's hook starts a bucket's all-reduce after BOTH streams — the gradients equal the single-stream,
    unwrapped step's."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import copy, os, sys, torch
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import avtex
from avtex import models
from tiny_encoders import TinySlowFast, seeded
from avtex import dist as adist
rank, world, local = adist.init_from_env()
dev = torch.device("cuda", local)
base = avtex.ContrastivePredictionTemporal(seeded(TinySlowFast, 1), seeded(TinySlowFast, 2), None, 1, 128, temp=0.1, window=5,
                                           stride=2, enc_arch="slowfast", img_size=32).to(dev).train()
g = torch.Generator().manual_seed(0)
qf = [torch.randn(2, 3, 8, 32, 32, generator=g).to(dev), torch.randn(2, 3, 32, 32, 32, generator=g).to(dev)]
tf = [torch.randn(2, 4, 3, 8, 32, 32, generator=g).to(dev), torch.randn(2, 4, 3, 32, 32, 32, generator=g).to(dev)]
crit = avtex.InfoNCECriterion()
lab = torch.zeros(2, dtype=torch.long, device=dev)
def grads(net, streams):
    models._TRAIN_STREAMS = streams
    for _ in range(3):  # (several steps: the side stream and the bucket hooks are exercised while earlier work is in flight)
        net.zero_grad(set_to_none=True)
        crit(net(qf, tf), lab).backward()
    torch.cuda.synchronize()
    mod = net.module if hasattr(net, "module") else net
    return {k: p.grad.detach().clone() for k, p in mod.named_parameters() if p.grad is not None}
want = grads(copy.deepcopy(base), 0)
from avtex.main import wrap_ddp
got = grads(wrap_ddp(copy.deepcopy(base), dev, local), 1)
assert models.training_side_streams(dev), "the side stream was not used"
assert set(got) == set(want)
worst = max(float((got[k] - want[k]).abs().max()) / (float(want[k].abs().max()) + 1e-30) for k in want)
assert worst < 1e-4, worst
print("DDP_STREAMS_OK", worst)
''' % (root, root)
    env = dict(os.environ, AVT_FORCE_PG="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0",
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DDP_STREAMS_OK" in r.stdout, r.stderr[-3000:]
