"""End-to-end `validate()` AT SIZE on the MI355X: BASELINE configs 2 and 3 on ~2k windows of a structured
128^2 video with the real SlowFast-8x8-R50 encoders in the default (contract-grade f16x3) arithmetic, aligned mode — the frames
list against the CPU oracle's normalise / similarity / select / walk over the SAME embedding tables (reference:
validate.py:324-685, the m=2 blend :524-527); and, on boxes with two or more GPUs, the same runs over real RCCL
(reference: DataParallel, main.py:420)."""
import os
import subprocess
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import cref, ref_py

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, S, FPS, L = 20, 4, 30.0, 2048


def _encoders(dev, video):
    from avtex import ops, synth
    from avtex.slowfast import SlowFast

    torch.manual_seed(0)
    q_mod = synth.randomise_bn(SlowFast().eval(), 10, 2.0, 0.1).to(dev)
    t_mod = synth.perturbed_copy(q_mod, 11, 0.05)
    cal = np.linspace(0, L - 1, 8).astype(np.int64) * S
    slow, fast = ops.clip_pack(video.to(dev), cal, W, out_hw=224, dtype=torch.float32)
    synth.calibrate_bn(q_mod, slow, fast)
    synth.calibrate_bn(t_mod, slow, fast)
    return q_mod.eval(), t_mod.eval()


def _args(model_type, threshold, nvl=10):
    return SimpleNamespace(vdata=None, adata=None, dadata=None, subsample_rate=1, fps=FPS, stride=S, window=W, enc_arch="slowfast",
                           img_size=224, model_type=model_type, mini_batchsize=100, threshold=threshold, alpha=0.5, temp=0.1,
                           driving_audio=None, da_feats="VGG", interpolation=False, new_video_length=nvl, results_folder=None,
                           logname="exp", batch_size=8, stitch_mode="aligned", enc_batch=166, enc_impl="auto", enc_dtype="fp32")


def _clamp_rows(a, n):
    return a[np.minimum(np.arange(n), a.shape[0] - 1)]


def test_config2_validate_at_size_equals_the_oracle_walk(avt, dev, capsys):
    """BASELINE config 2: contrastive synthesis (-e) of one ~2k-window video, SlowFast encoders, w=20 stride=4, one MI355X."""
    from avtex import synth
    from avtex.validate import validate

    video = synth.structured_video(11, L * S + W + 1, 128, 128, variety=1)
    q_mod, t_mod = _encoders(dev, video)
    model = avt.ContrastivePredictionTemporal(q_mod, t_mod, None, 1, 128, 0.1, W, S, 0.3, mini_batchsize=100,
                                              enc_arch="slowfast", img_size=224).to(dev).eval()
    args = _args(1, 0.3)
    np.random.seed(7)
    frames = validate(model, args, video_name="synthetic", model_type=1, video=(video, FPS))
    out = capsys.readouterr().out
    assert "precision f16x3 (contract grade)" in out and "Windows encoded: %d " % (2 * L) in out
    eng = validate.last_engine
    assert eng.N == L and tuple(eng.Qv.shape) == (L, 2304)
    qn, _, _ = cref.l2norm_rows(eng.Qv.cpu().numpy(), want_split=False)
    tn, _, _ = cref.l2norm_rows(eng.Tv.cpu().numpy(), want_split=False)
    sim = cref.sim_f32(qn, tn, 0.1)
    assert np.array_equal(eng.sim.cpu().numpy(), sim)  # the whole 2048^2 matrix, bit for bit
    kept = []

    def row_fn(q):
        o = cref.row_transition(sim[q : q + 1], q_ids=np.array([q]), n_seg=L, threshold=0.3, cap=L)
        kept.append(int(o["cnt"][0]))
        return o["idx"][0, : o["cnt"][0]], ref_py.target_segment_ids(q, L)

    ref_frames, steps, jumps = ref_py.stitch_walk(row_fn, len(video), W, S, int(np.ceil(FPS)) * 10, q_id=10, rng=np.random.RandomState(7))
    assert frames == ref_frames and len(frames) >= 300
    assert 1 <= np.mean(kept) < 0.5 * L, np.mean(kept)  # rows are neither degenerate nor all-surviving
    print("config 2 at size: %d steps, %d jumps, %.1f survivors per visited row" % (len(steps), jumps, np.mean(kept)))


def test_config3_validate_m2_driving_audio_at_size_equals_the_oracle_walk(avt, dev, capsys):
    """BASELINE config 3: audio-conditioned synthesis (m=2): SlowFast + VGGish, D = 2304 + 12288 = 14592 jointly normalised,
    driving audio blended at alpha = 0.5, th 0.0 (argmax) — every stitch step against the oracle on the same tables."""
    from avtex import synth
    from avtex.validate import audio_start_segment, validate

    video = synth.structured_video(12, L * S + W + 1, 128, 128, variety=1)
    q_mod, t_mod = _encoders(dev, video)
    torch.manual_seed(3)
    vgg = avt.VGGish()
    model = avt.ContrastivePredictionTemporal(q_mod, t_mod, vgg, 2, 128, 0.1, W, S, 0.0, mini_batchsize=100,
                                              enc_arch="slowfast", img_size=224).to(dev).eval()
    rng = np.random.default_rng(5)
    n_in = len(video)
    wave = (0.1 * rng.standard_normal(int(n_in / FPS * 16000) + 16000)).astype(np.float32)
    wave_da = (0.1 * rng.standard_normal(12 * 16000)).astype(np.float32)
    args = _args(2, 0.0)
    np.random.seed(9)
    frames = validate(model, args, video_name="synthetic", model_type=2, video=(video, FPS), audio=(wave, 16000),
                      driving_audio=(wave_da, 16000))
    capsys.readouterr()
    eng = validate.last_engine
    assert tuple(eng.A.shape)[1] == 12288 and eng.Ad is not None
    a = _clamp_rows(eng.A.cpu().numpy(), L)
    qn, _, _ = cref.l2norm_rows(eng.Qv.cpu().numpy(), a, want_split=False)
    tn, _, _ = cref.l2norm_rows(eng.Tv.cpu().numpy(), a, want_split=False)
    assert qn.shape == (L, 14592)
    sim = cref.sim_f32(qn, tn, 0.1)
    assert np.array_equal(eng.sim.cpu().numpy(), sim)
    dn, _, _ = cref.l2norm_rows(eng.Ad.cpu().numpy(), want_split=False)
    an, _, _ = cref.l2norm_rows(_clamp_rows(eng.A_da.cpu().numpy(), L), want_split=False)
    sim_a = cref.sim_f32(dn, an, 0.1)
    step = [1]  # validate()'s iter_count starts at 1 (validate.py:257) and indexes the driving example of the step

    def row_fn(q):
        o = cref.row_transition(sim[q : q + 1], q_ids=np.array([q]), n_seg=L, sim_a=sim_a[step[0] : step[0] + 1], alpha=0.5,
                                threshold=0.0, cap=L)
        step[0] += 1
        return o["idx"][0, : o["cnt"][0]], ref_py.target_segment_ids(q, L)

    # start segment: validate.py:223-240 on the log-mel examples (host logic of the product, pinned by fixture G5 sf_da)
    from avtex.audio_frontend import waveform_to_examples

    apf = int(np.floor(16000 / FPS))
    aeg = torch.from_numpy(waveform_to_examples(wave[: n_in * apf], 16000)).float()[:L]
    deg = torch.from_numpy(waveform_to_examples(wave_da, 16000)).float()
    q0 = audio_start_segment(aeg, deg[0])
    ref_frames, steps, _ = ref_py.stitch_walk(row_fn, n_in, W, S, int(np.ceil(FPS)) * 10, q_id=q0, rng=np.random.RandomState(9))
    assert frames == ref_frames and len(frames) >= 300


def _run(cmd, env_extra=None, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), **(env_extra or {}))
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


needs_two = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (real RCCL over xGMI)")


@needs_two
def test_bench_self_launch_two_ranks_over_rccl(dev):
    """`python bench.py --gpus 2` with NO launcher: bench starts its own ranks (a child process of torch.distributed.run),
    shards 2 x 256 windows, all-gathers T_hat over RCCL and prints one line that says so itself."""
    import json

    r = _run([sys.executable, "bench.py", "--gpus", "2", "--windows", "256", "--steps", "1", "--warmup", "1", "--enc-batch", "64",
              "--no-fast", "--no-train-leg", "--no-cpu-baseline", "--no-nxn-legs", "--no-precision-block"])
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["config"]["windows_total"] == 512
    assert line["allgather_ms"] is not None and line["allgather_ms"] > 0 and line["value"] > 0


_WORLD_SCRIPT = r"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, %(root)r)
import avtex as avt
from avtex import dist as adist, synth
from avtex.validate import validate
sys.path.insert(0, os.path.join(%(root)r, "tests"))
from test_gpu_e2e import _args, _encoders, W, S, FPS
rank, world, local = adist.init_from_env(backend=os.environ.get("AVT_TEST_BACKEND") or None)
dev = torch.device("cuda", local %% torch.cuda.device_count())  # (gloo ranks may share one GPU; RCCL ranks have one each)
torch.cuda.set_device(dev)
n = int(os.environ.get("AVT_TEST_WINDOWS", "256"))
video = synth.structured_video(11, n * S + W + 1, 128, 128, variety=1)
import test_gpu_e2e
test_gpu_e2e.L = n
q_mod, t_mod = _encoders(dev, video)
model = avt.ContrastivePredictionTemporal(q_mod, t_mod, None, 1, 128, 0.1, W, S, 0.3, mini_batchsize=100, enc_arch="slowfast",
                                          img_size=224).to(dev).eval()
args = _args(1, 0.3, nvl=5)
args.enc_batch = 64
np.random.seed(7)
frames = validate(model, args, video_name="synthetic", model_type=1, video=(video, FPS))
if rank == 0:
    print("FRAMES " + json.dumps(frames))
if torch.distributed.is_initialized():
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
"""


@needs_two
def test_validate_world2_over_rccl_equals_world1(dev, tmp_path):
    """validate() sharded over two GPUs (index blocks -> RCCL all-gather of T_hat -> row blocks -> survivors to rank 0) walks
    the same frames list as one GPU."""
    import json

    script = tmp_path / "world.py"
    script.write_text(_WORLD_SCRIPT % {"root": ROOT})

    def frames_of(r):
        assert r.returncode == 0, r.stderr[-3000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("FRAMES ")][-1][7:])

    one = frames_of(_run([sys.executable, str(script)]))
    two = frames_of(_run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", "29533", str(script)]))
    assert one == two and len(one) >= 150


def test_validate_two_ranks_sharing_one_gpu_equal_one_rank(dev, tmp_path):
    """The sharded validate() path with the REAL kernels on a one-GPU box: two ranks under a gloo group share cuda:0 (RCCL
    refuses two ranks on a device; gloo carries the exchange through the host, avtex.dist.all_gather_rows) — index blocks,
    the all-gather of T_hat, row blocks of the similarity + select, survivors to rank 0 — and rank 0 walks the same frames
    list as a single process.  What the two-GPU RCCL test above checks, minus RCCL itself, where a second GPU is absent
    (reference: the DataParallel scatter / gather of validate.py:320, 349-363, 442-445)."""
    import json

    script = tmp_path / "world.py"
    script.write_text(_WORLD_SCRIPT % {"root": ROOT})

    def frames_of(r):
        assert r.returncode == 0, r.stderr[-3000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("FRAMES ")][-1][7:])

    small = {"AVT_TEST_WINDOWS": "96"}
    one = frames_of(_run([sys.executable, str(script)], small))
    two = frames_of(_run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", "29541", str(script)], dict(small, AVT_TEST_BACKEND="gloo")))
    assert one == two and len(one) >= 150


def test_bench_self_launch_two_ranks_sharing_one_gpu(dev):
    """`python bench.py --gpus 2 --dist-backend gloo` on a one-GPU box: bench starts its own two ranks (children of
    torch.distributed.run), both on cuda:0, shards 2 x 128 windows, exchanges T_hat (staged through the host under gloo),
    takes the max-over-ranks time and rank 0 prints ONE line that names the world it ran in — the N > 1 control flow of the
    bench contract with the real kernels, where the RCCL form above needs a second GPU."""
    import json

    r = _run([sys.executable, "bench.py", "--gpus", "2", "--dist-backend", "gloo", "--windows", "128", "--steps", "1", "--warmup", "1",
              "--enc-batch", "64", "--no-fast", "--no-train-leg", "--no-cpu-baseline", "--no-nxn-legs", "--no-precision-block"])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # rank 0 alone prints
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["gloo_ranks"] == 2 and "rccl_ranks" not in line
    assert line["config"]["windows_total"] == 256 and line["config"]["windows_per_gpu"] == 128
    assert line["allgather_ms"] is not None and line["allgather_ms"] > 0 and line["value"] > 0 and line["scaling"] == "weak"


def test_bench_eight_ranks_sharing_one_gpu_headline_and_one_item_training(dev):
    """First-contact readiness for the driver's 8-GPU run, on a one-GPU box: `bench.py --gpus 8 --dist-backend
    gloo` starts EIGHT ranks on cuda:0 — (1) the headline leg: 8 x 32 windows sharded by index, one all-gather of T_hat, row blocks
    of the similarity + select, max-over-ranks time, one JSON line from rank 0; (2) `--mode train`: BASELINE config 5 at its real
    per-rank shape — batch 8 over 8 ranks = ONE item (16 clips) per rank, DistributedDataParallel through main.wrap_ddp (bucketed
    all-reduce behind the comm hook that orders it after every stream of the forward), strong scaling."""
    import json

    r = _run([sys.executable, "bench.py", "--gpus", "8", "--dist-backend", "gloo", "--windows", "32", "--steps", "1", "--warmup", "1",
              "--enc-batch", "32", "--no-fast", "--no-train-leg", "--no-cpu-baseline", "--no-nxn-legs", "--no-precision-block"])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["gloo_ranks"] == 8 and line["scaling"] == "weak"
    assert line["config"]["windows_total"] == 256 and line["config"]["windows_per_gpu"] == 32
    assert line["allgather_ms"] > 0 and line["value"] > 0 and line["ms_per_step_rank_max"] >= line["ms_per_step_rank_min"] > 0

    r = _run([sys.executable, "bench.py", "--mode", "train", "--gpus", "8", "--dist-backend", "gloo", "--steps", "1", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["scaling"] == "strong" and line["unit"] == "clips/s"
    assert line["config"]["items_per_rank"] == 1 and line["config"]["clips_per_step"] == 128
    assert "DistributedDataParallel" in line["config"]["parallelism"]
    assert all(np.isfinite(v) for v in line["loss_first_last"]) and line["value"] > 0


@pytest.mark.gpu
def test_bench_two_ranks_train_as_replayed_graphs_with_the_exchange_outside(dev):
    """Round 6: `bench.py --mode train --gpus 2 --train-graph 1` — every rank replays its forward + backward as ONE HIP graph (the plain
    module, not DistributedDataParallel) and the mean gradient crosses the ranks in two all-reduces AFTER the replay (the convolutions'
    gradient arena in place, one flat buffer for everything else), then SGD.  Two gloo ranks share cuda:0 here; what the test holds: the
    capture works next to a process group, the ranks hold the SAME parameters after the steps (checksum spread 0), the loss is finite
    and the line says which form ran."""
    import json

    r = _run([sys.executable, "bench.py", "--mode", "train", "--gpus", "2", "--dist-backend", "gloo", "--train-graph", "1", "--train-items", "2",
              "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["items_per_rank"] == 1 and line["config"]["hip_graph"] == "captured"
    assert "replayed as a HIP graph" in line["config"]["parallelism"]
    assert all(np.isfinite(v) for v in line["loss_first_last"]) and line["value"] > 0
    assert line["ranks_param_checksum_spread"] == 0.0
