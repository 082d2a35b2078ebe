#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03n
mkdir -p $OUT
cd $R
for CFG in "83 2" "83 1" "166 1" "166 2" "125 1" "104 1"; do
  set -- $CFG
  python bench.py --steps 1 --warmup 1 --no-fast --no-cpu-baseline --no-precision-block --no-nxn-legs --no-train-leg --enc-batch $1 --streams $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('enc-batch $1 streams $2', d['value'], d['ms_per_step'])" | tee -a $OUT/bench_sweep.log
done
