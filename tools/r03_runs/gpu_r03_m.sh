#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03m
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -4
for N in 2048 4096 16384; do
  echo "v2 N=$N"; python tools/probe_sim.py $N 2>&1 | grep "^f32"
  echo "v1 N=$N"; AVT_SIM_F32_V2=0 python tools/probe_sim.py $N 2>&1 | grep "^f32"
done | tee $OUT/sim.log
python - <<'PY' | tee $OUT/select.log
import torch, sys
sys.path.insert(0, ".")
import avtex
from avtex import ops
dev = torch.device("cuda:0")
for nq, nt in ((4096, 4096), (2048, 16384)):
    sim = (torch.rand((nq, nt), device=dev) * 4 + 0.5)
    q_ids = torch.arange(nq, device=dev, dtype=torch.int64)
    for name, fn in (("row_transition th0.3", lambda: ops.row_transition(sim, q_ids=q_ids if nq == nt else None, threshold=0.3, cap=64)),
                     ("row_transition th0.0", lambda: ops.row_transition(sim, q_ids=q_ids if nq == nt else None, threshold=0.0, cap=64)),
                     ("row_topk 8", lambda: ops.row_topk(sim, 8))):
        for _ in range(3): fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): fn()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 20
        print("%d x %d %-22s %.4f ms  %.0f GB/s" % (nq, nt, name, ms, nq * nt * 4 / ms / 1e6))
PY
