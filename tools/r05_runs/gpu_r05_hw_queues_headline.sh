#!/bin/bash
# does GPU_MAX_HW_QUEUES=8 (the package's default since the end of round 5) cost the single-stream headline run anything?
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
L=gpurun_out/r05_hw_queues_headline.log; : > $L
for rep in 1 2 3; do for q in 4 8; do
  GPU_MAX_HW_QUEUES=$q python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast --no-nxn-legs --no-train-leg --no-inputs-r03-leg --no-precision-block 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('queues=$q headline', d['value'], d['ms_per_step'], d['roofline']['achieved'])" | tee -a $L
done; done
