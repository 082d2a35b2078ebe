#!/bin/bash
# round 5 (VERDICT r4 items 2 + 3): 600 steps of config 5 with the x3 training arithmetic and 600 with MIOpen fp32 from the same seed /
# batches; the x3-trained pair -> checkpoint -> main.py -e --resume -> the contract on its own weights
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 2400 python tools/train_convergence.py --steps ${1:-600} --lr 0.1 --init default --modes x3 fp32 --roundtrip \
  --out gpurun_out/train_convergence.json 2> gpurun_out/train_convergence.err | tail -c 3000
