#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03r
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_bn_train.py -x -q 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_train_step.py -x -q -s -k "one_batch" 2>&1 | tail -6
for P in 1 8 4; do
python bench.py --mode train --steps 3 --warmup 2 --train-pass-items $P 2>$OUT/train_pass$P.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('items per pass $P', d['value'], d['ms_per_step'], d['loss_first_last'], d.get('max_memory_allocated_gb'))" | tee -a $OUT/train_pass.log
tail -2 $OUT/train_pass$P.err | cut -c1-300
done
