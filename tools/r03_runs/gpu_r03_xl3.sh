#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03g
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_x3.py -x -q -k "conv_x3_matches or encoder_matches" 2>&1 | tail -15 > $OUT/tests.log
tail -4 $OUT/tests.log
for B in 83 64; do
python tools/probe_x3.py f16x3 $B > $OUT/probe_xl_b$B.log 2>&1
echo XL b$B; sed -n 2,3p $OUT/probe_xl_b$B.log; grep "xl" $OUT/probe_xl_b$B.log | head -14
done
for EB in 64 83; do
  python bench.py --steps 1 --warmup 1 --no-fast --no-cpu-baseline --no-precision-block --no-nxn-legs --no-train-leg --enc-batch $EB 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('enc-batch $EB', d['value'], d['ms_per_step'])" | tee -a $OUT/bench_sweep.log
done
