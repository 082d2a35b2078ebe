#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03j
mkdir -p $OUT
cd $R
python tools/probe_train_layers.py > $OUT/train_layers_stem_patch.log 2>&1
grep -E "^item|hand-written|by kind|stem|planes" $OUT/train_layers_stem_patch.log | cut -c1-160
for V in 0 1; do
AVT_TRAIN_STEM_PATCH=$V python bench.py --mode train --steps 3 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('AVT_TRAIN_STEM_PATCH=$V', d['value'], d['ms_per_step'])" | tee -a $OUT/train_stem_patch.log
done
timeout 900 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_train_conv.py -x -q 2>&1 | tail -3
