#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03m
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_x3.py -x -q 2>&1 | tail -3
for round in 1 2; do for V in 0 1; do
AVT_WBLK_X3=$V python tools/probe_x3.py f16x3 83 > $OUT/probe_wblk${V}_$round.log 2>&1
echo "WBLK=$V round $round"; sed -n 2,3p $OUT/probe_wblk${V}_$round.log; grep "xl" $OUT/probe_wblk${V}_$round.log | head -5
done; done
