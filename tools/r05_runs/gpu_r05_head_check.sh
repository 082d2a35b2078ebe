#!/bin/bash
# the GPU suite, smoke and the default bench run on the round's last commit (after gpu_r05_final.sh's collection: the hardware-queue default)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) 2>&1 | tee gpurun_out/r05_head_suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r05_head_smoke.log
python bench.py > gpurun_out/r05_head_bench.json 2> gpurun_out/r05_head_bench.err; tail -c 1600 gpurun_out/r05_head_bench.json
