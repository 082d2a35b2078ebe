#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03f
mkdir -p $OUT
cd $R
for B in 83 80; do
python tools/probe_x3.py f16x3 $B > $OUT/probe_xl_b$B.log 2>&1
AVT_CONV_X3_XL=0 python tools/probe_x3.py f16x3 $B > $OUT/probe_noxl_b$B.log 2>&1
echo XL b$B; sed -n 2,3p $OUT/probe_xl_b$B.log; grep "xl" $OUT/probe_xl_b$B.log | head -8
echo NOXL b$B; sed -n 2,3p $OUT/probe_noxl_b$B.log; grep "xl" $OUT/probe_noxl_b$B.log | head -8
done
for XL in 1 0; do for EB in 64 83; do
  AVT_CONV_X3_XL=$XL python bench.py --steps 1 --warmup 1 --no-fast --no-cpu-baseline --no-precision-block --no-nxn-legs --no-train-leg --enc-batch $EB 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('XL=$XL enc-batch $EB', d['value'], d['ms_per_step'])" | tee -a $OUT/bench_sweep.log
done; done
