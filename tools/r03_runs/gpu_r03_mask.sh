#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03r
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_bn_train.py -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_train_step.py -x -q -k "one_batch" 2>&1 | tail -2
python bench.py --mode train --steps 3 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train', d['value'], d['ms_per_step'], d['loss_first_last'], d.get('max_memory_allocated_gb'))" | tee -a $OUT/train_mask.log
python tools/probe_train_layers.py 8 > $OUT/train_layers_8items_mask.log 2>&1; grep -E "^pass|hand-written|by kind" $OUT/train_layers_8items_mask.log
