#!/bin/bash
# the end-of-round collection on the final code: the whole GPU suite + smoke, bench + rocprofv3 stats + PMC passes
# (tools/gpu_profile_round.sh r05), the config-5 profile, bench --weights trained
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd)
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) 2>&1 | tee gpurun_out/r05_suite_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r05_smoke.log
bash tools/gpu_profile_round.sh r05 > gpurun_out/r05_profile_round.log 2>&1
cp bench_detail.json gpurun_out/profile_r05/bench_detail.json 2>/dev/null
bash tools/r05_runs/gpu_r05_train_prof.sh final > gpurun_out/r05_train_prof.log 2>&1
cd $R
python bench.py --weights trained --trained-steps 300 --steps 4 --warmup 1 --no-train-leg --no-nxn-legs --no-cpu-baseline --no-fast > gpurun_out/r05_bench_trained.json 2> gpurun_out/r05_bench_trained.err
python tools/probe_sim_xl.py 2>&1 | tail -4 > gpurun_out/r05_probe_sim_xl.log
tail -c 2500 gpurun_out/profile_r05/bench.json; echo; tail -c 900 gpurun_out/r05_bench_trained.json
