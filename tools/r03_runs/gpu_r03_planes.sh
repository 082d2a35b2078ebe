#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03u
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_train_conv.py -x -q 2>&1 | tail -3
for round in 1 2; do for V in 0 1; do
AVT_TRAIN_PLANES_HIP=$V python bench.py --mode train --steps 3 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('AVT_TRAIN_PLANES_HIP=$V', d['value'], d['ms_per_step'], d['loss_first_last'])" | tee -a $OUT/train_planes_ab.log
done; done
timeout 900 python -m pytest tests/test_gpu_train_step.py -x -q 2>&1 | tail -3
