#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03k
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_bn_train.py -x -q 2>&1 | tail -3
python tools/probe_train_layers.py > $OUT/train_layers_bn_mask.log 2>&1
grep -E "^item|hand-written|by kind" $OUT/train_layers_bn_mask.log | cut -c1-160
python bench.py --mode train --steps 3 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train', d['value'], d['ms_per_step'])" | tee -a $OUT/train_bn.log
timeout 900 python -m pytest tests/test_gpu_train_step.py -x -q 2>&1 | tail -3
