#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03l
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_train_conv.py -x -q -k "micro_batch" 2>&1 | tail -12
for round in 1 2; do
python bench.py --mode train --steps 3 --warmup 2 --no-grad-accumulator 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('autograd accumulation', d['value'], d['ms_per_step'], d['loss_first_last'])" | tee -a $OUT/train_acc.log
python bench.py --mode train --steps 3 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MicroBatchGradients', d['value'], d['ms_per_step'], d['loss_first_last'])" | tee -a $OUT/train_acc.log
done
