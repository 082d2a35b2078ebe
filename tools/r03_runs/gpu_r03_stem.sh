#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03j
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_train_conv.py -x -q -s -k "clip_planes or stems_run or stem_patch" 2>&1 | tail -30 > $OUT/tests_stem.log
tail -25 $OUT/tests_stem.log
