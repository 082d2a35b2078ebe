#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03q
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_x3.py tests/test_gpu_train_conv.py -x -q 2>&1 | tail -4
for round in 1 2; do for V in 0 1; do
AVT_STEM_FM_X3=$V python tools/probe_x3.py f16x3 83 > $OUT/probe_fm${V}_$round.log 2>&1
echo "AVT_STEM_FM_X3=$V round $round"; sed -n 2,3p $OUT/probe_fm${V}_$round.log; grep "stem" $OUT/probe_fm${V}_$round.log | head -3
done; done
