#!/bin/bash
# round 5, final code (BatchNorm statistics on the convolution epilogues in both directions, the 256 x 128 weight-gradient tile, the
# fast pathway on a side stream): the 600-step x3 run of gpu_r05_convergence.sh again, to be held against the curves recorded in
# profiles/r05/train_convergence.json (x3 before those changes, MIOpen fp32) + the round trip through the checkpoint
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 1500 python tools/train_convergence.py --steps ${1:-600} --lr 0.1 --init default --modes x3 --roundtrip --against profiles/r05/train_convergence.json \
  --out gpurun_out/train_convergence_final_code.json 2> gpurun_out/train_convergence_final_code.err | tail -c 3000
