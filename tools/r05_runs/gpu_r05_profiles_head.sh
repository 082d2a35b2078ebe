#!/bin/bash
# the profile collection of gpu_r05_final.sh again on the round's last code commit (no suite, no trained-weights leg)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd)
mkdir -p gpurun_out
bash tools/gpu_profile_round.sh r05 > gpurun_out/r05_profile_round.log 2>&1
cp bench_detail.json gpurun_out/profile_r05/bench_detail.json 2>/dev/null
bash tools/r05_runs/gpu_r05_train_prof.sh final > gpurun_out/r05_train_prof.log 2>&1
cd $R
tail -c 1500 gpurun_out/profile_r05/bench.json
