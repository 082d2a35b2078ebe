#!/bin/bash
# config 5: the query encoder on a side stream (AVT_TRAIN_STREAMS) — parity test + A/B of the training step
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03i
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_train_step.py -x -q 2>&1 | tail -5 > $OUT/tests_train.log
tail -3 $OUT/tests_train.log
for round in 1 2; do for V in 0 1; do
AVT_TRAIN_STREAMS=$V python bench.py --mode train --steps 3 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('AVT_TRAIN_STREAMS=$V', d['value'], d['ms_per_step'])" | tee -a $OUT/train_streams.log
done; done
