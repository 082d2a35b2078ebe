#!/bin/bash
# the in-code default (GPU_MAX_HW_QUEUES=8 set by the package before HIP initialises) against an explicit 4: two more streams in the process
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
L=gpurun_out/r05_hw_queues_default.log; : > $L
run() { python bench.py --mode train --steps 6 --warmup 2 "${@:2}" 2> gpurun_out/r05_hw_queues.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])" | tee -a $L; }
for rep in 1 2; do
  run "extra_streams=2 in-code default"  --train-extra-streams 2
  GPU_MAX_HW_QUEUES=4 run "extra_streams=2 GPU_MAX_HW_QUEUES=4" --train-extra-streams 2
done
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-nxn-legs --no-inputs-r03-leg --no-precision-block 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default run (short)', d['value'], 'fast', d.get('fast_mode_value'), 'train', d.get('train_clips_per_s'))" | tee -a $L
