#!/bin/bash
# round 5, call 1: same-box baseline of the config-5 leg + a short learning-rate sweep of the x3 training arithmetic
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
python bench.py --mode train --steps 5 --warmup 2 > gpurun_out/r05_train_base.json 2> gpurun_out/r05_train_base.err
tail -c 600 gpurun_out/r05_train_base.json
for lr in 1e-3 1e-2 5e-2; do
  timeout 600 python tools/train_convergence.py --steps 60 --lr $lr --modes x3 --out gpurun_out/r05_sweep_lr$lr.json 2> gpurun_out/r05_sweep_lr$lr.err | tail -c 400
done
