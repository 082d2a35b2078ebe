#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03t
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_train_conv.py -x -q -s -k "match_fp32_autograd or fork" 2>&1 | tail -16
for round in 1 2; do for V in 0 1; do
AVT_CONV_X3_XL_IO32=$V python bench.py --mode train --steps 3 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('AVT_CONV_X3_XL_IO32=$V', d['value'], d['ms_per_step'], d['loss_first_last'])" | tee -a $OUT/train_xl_io32_ab.log
done; done
python tools/probe_train_layers.py 8 > $OUT/train_layers_xl32.log 2>&1; grep -E "^pass|hand-written|by kind" $OUT/train_layers_xl32.log; grep -E "fwd  |dgrad" $OUT/train_layers_xl32.log | head -12 | cut -c1-140
