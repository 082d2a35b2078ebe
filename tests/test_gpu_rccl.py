"""The N>1 launch contract on a 1-GPU box: bench.py under torch.distributed.run with a one-rank RCCL process group
(AVT_FORCE_PG=1), so init_process_group("nccl"), all_gather_into_tensor, all_reduce(MAX) and barrier of the sharded
path run on the device exactly as they do at N=2,4,8 (reference: DataParallel, main.py:420)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_under_torchrun_one_rank_rccl(dev):
    env = dict(os.environ, AVT_FORCE_PG="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--windows", "128",
           "--enc-batch", "64",  # (two batches: the bf16 leg's two encoder streams have something to overlap)
           "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-precision-block", "--no-nxn-legs", "--no-train-leg"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    assert len(line.encode()) < 4096  # the driver's capture holds ~8 KB: one compact line (bench.LINE_LIMIT)
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["unit"] == "clip-windows/s"
    assert d["roofline"]["kernel"].startswith("conv_x3") and 0 < d["roofline"]["frac"] < 1
    assert set(d["roofline"]) >= {"kernel", "bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert d["dtype"] == "f16x3" and d["fast_mode_value"] > d["value"] and d["config"]["workload"]
    assert os.path.exists(os.path.join(ROOT, "bench_detail.json"))  # the per-kernel tables live beside the script


_VALIDATE_SCRIPT = r'''
import json, os, sys
from types import SimpleNamespace
import numpy as np, torch
root = %r
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import avtex
from avtex import dist as adist
from tiny_encoders import TinySlowFast, seeded
rank, world, local = adist.init_from_env()   # AVT_FORCE_PG=1: a one-rank RCCL group; unset: no process group at all
dev = torch.device("cuda", local)
g = torch.Generator().manual_seed(5)
n, hw = 150, 40
base = torch.rand((n // 6 + 2, hw, hw, 3), generator=g)
t = torch.linspace(0, n / 6, n); i0 = t.floor().long(); fr = (t - i0.float()).view(-1, 1, 1, 1)
video = (((1 - fr) * base[i0] + fr * base[i0 + 1]).clamp(0, 1) * 255).to(torch.uint8)
rng = np.random.default_rng(3)
wave = (0.1 * rng.standard_normal(16 * 16000)).astype(np.float32)
wave_da = (0.1 * rng.standard_normal(4 * 16000)).astype(np.float32)
out = {}
for m_type, driving in ((1, False), (2, False), (2, True)):
    model = avtex.ContrastivePredictionTemporal(seeded(TinySlowFast, 1), seeded(TinySlowFast, 2), seeded(avtex.VGGish, 3),
                                                m_type, 128, temp=0.1, window=5, stride=2, threshold=0.3,
                                                mini_batchsize=8, enc_arch="slowfast", img_size=32).to(dev).eval()
    args = SimpleNamespace(vdata=None, adata=None, dadata=None, subsample_rate=1, fps=10, stride=2, window=5,
                           enc_arch="slowfast", img_size=32, model_type=m_type, mini_batchsize=8, threshold=0.3,
                           alpha=0.5, temp=0.1, driving_audio=None, da_feats="VGG", interpolation=False,
                           new_video_length=3, results_folder=None, logname="exp", batch_size=8,
                           stitch_mode="aligned", ref_num_gpus=1, enc_batch=16, enc_impl="auto")
    np.random.seed(5)
    frames = avtex.validate(model, args, video_name="x", model_type=m_type, video=(video.numpy(), 10.0),
                            audio=(wave, 16000) if m_type == 2 else None,
                            driving_audio=(wave_da, 16000) if driving else None)
    out["m%%d_da%%d" %% (m_type, int(driving))] = frames
print("FRAMES " + json.dumps(out))
''' % ROOT


@pytest.mark.gpu
def test_validate_sharded_path_on_one_rank_rccl_equals_plain(dev):
    """validate() in aligned mode under a (one-rank) RCCL process group takes the sharded route — encode own block,
    all-gather of the normalised target table, row-block similarity + select, survivors to rank 0 (dist.sharded_survivors;
    reference: validate.py:320, 349-363, 442-445, 481-493) — and must print the frames list of the plain single-process
    route, for m=1, m=2 (audio columns) and m=2 with driving audio (matrix gathered, per-step blend on rank 0)."""
    outs = []
    for pg in ("1", ""):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0", RANK="0",
                   WORLD_SIZE="1", LOCAL_RANK="0")
        env.pop("AVT_FORCE_PG", None)
        if pg:
            env["AVT_FORCE_PG"] = "1"
        r = subprocess.run([sys.executable, "-c", _VALIDATE_SCRIPT], env=env, cwd=ROOT, capture_output=True, text=True,
                           timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(json.loads([l for l in r.stdout.splitlines() if l.startswith("FRAMES ")][-1][7:]))
    assert outs[0] == outs[1]
    assert all(len(v) >= 30 for v in outs[0].values()) and outs[0]["m2_da0"] != outs[0]["m1_da0"]
