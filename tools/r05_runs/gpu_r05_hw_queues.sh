#!/bin/bash
# round 5: how does the config-5 step (three streams of its own) take other streams in the process — an RCCL communicator's, another
# engine's — and does GPU_MAX_HW_QUEUES (default 4) change it?
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
L=gpurun_out/r05_hw_queues.log; : > $L
run() {  # $1 = label, rest = bench flags; env from the caller
  python bench.py --mode train --steps 6 --warmup 2 "${@:2}" 2> gpurun_out/r05_hw_queues.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])" | tee -a $L
}
for rep in 1 2; do
  run "extra_streams=0 queues=default"
  run "extra_streams=2 queues=default" --train-extra-streams 2
  GPU_MAX_HW_QUEUES=8 run "extra_streams=2 queues=8" --train-extra-streams 2
  GPU_MAX_HW_QUEUES=8 run "extra_streams=0 queues=8"
done
