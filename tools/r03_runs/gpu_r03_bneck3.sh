#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03d
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_x3.py -x -q -k "bneck or encoder_matches" 2>&1 | tail -25 > $OUT/tests.log
tail -5 $OUT/tests.log
python tools/probe_x3.py f16x3 64 > $OUT/probe_b64.log 2>&1
sed -n 2,3p $OUT/probe_b64.log; grep "fused bottleneck" $OUT/probe_b64.log
for EB in 60 62 64 83 125; do
  python bench.py --steps 1 --warmup 1 --no-fast --no-cpu-baseline --no-precision-block --no-nxn-legs --no-train-leg --enc-batch $EB 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('enc-batch $EB', d['value'], d['ms_per_step'])" | tee -a $OUT/bench_sweep.log
done
