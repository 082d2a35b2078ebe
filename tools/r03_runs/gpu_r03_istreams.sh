#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03l
mkdir -p $OUT
cd $R
for round in 1 2; do for S in 1 2; do
python bench.py --mode train --steps 3 --warmup 2 --item-streams $S 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('item streams $S', d['value'], d['ms_per_step'], d['loss_first_last'])" | tee -a $OUT/train_istreams.log
done; done
