#!/bin/bash
# why is the config-5 leg ~4 % slower as the last leg of the default bench run than alone?  (see gpurun_out/r05_trainleg_order.log)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default legs: value', round(d['value'],1), 'train', round(d['train_clips_per_s'],1), d['train_ms_per_step'])" | tee -a gpurun_out/r05_trainleg_order.log
python bench.py --mode train --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('alone: train', round(d['value'],1), d['ms_per_step'])" | tee -a gpurun_out/r05_trainleg_order.log
