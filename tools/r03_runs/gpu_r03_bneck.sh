#!/bin/bash
# round 3: fused x3 bottleneck — parity, per-layer probe (fused on / off), small-batch sweep, encoder-batch sweep
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03b
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_x3.py -x -q -k "bneck or encoder_matches" 2>&1 | tail -25 > $OUT/tests.log
tail -5 $OUT/tests.log
for B in 64; do
  python tools/probe_x3.py f16x3 $B > $OUT/probe_b${B}_fused.log 2>&1
  AVT_FUSE_BLOCK_X3=0 python tools/probe_x3.py f16x3 $B > $OUT/probe_b${B}_plain.log 2>&1
done
head -3 $OUT/probe_b64_fused.log | tail -2; head -3 $OUT/probe_b64_plain.log | tail -2
grep "fused bottleneck" $OUT/probe_b64_fused.log
for B in 8 16 32; do
  AVT_FUSE_BLOCK_X3=0 python tools/probe_x3.py f16x3 $B 2>&1 | sed -n 2,3p > $OUT/probe_b${B}_plain_head.log
  cat $OUT/probe_b${B}_plain_head.log
done
for EB in 128 125 62; do
  python bench.py --steps 1 --warmup 1 --no-fast --no-cpu-baseline --no-precision-block --no-nxn-legs --no-train-leg --enc-batch $EB 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('enc-batch $EB', d['value'], d['ms_per_step'])" | tee -a $OUT/bench_sweep.log
done
