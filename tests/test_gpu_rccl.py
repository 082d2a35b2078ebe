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
           "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-precision-block"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["unit"] == "clip-windows/s"
    assert d["roofline"]["kernel"].startswith("conv_x3_kernel") and 0 < d["roofline"]["frac"] < 1
    assert d["dtype"] == "f16x3" and d["fast_mode"]["dtype"] == "bf16" and d["fast_mode"]["value"] > d["value"]
