#!/bin/bash
# A/B of the XL tile's K loop: AVT_CONV_X3_XL_V=0 (barrier -> reads -> both slices), 1 (rotated by one k-slice), 2 (16x16x32 MFMAs)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03h
mkdir -p $OUT
cd $R
AVT_CONV_X3_XL_V=2 timeout 900 python -m pytest tests/test_gpu_x3.py -x -q 2>&1 | tail -5 > $OUT/tests_v2.log
tail -3 $OUT/tests_v2.log
for round in 1 2; do for V in 0 2; do
AVT_CONV_X3_XL_V=$V python tools/probe_x3.py f16x3 83 > $OUT/probe_xl_v${V}_$round.log 2>&1
echo "XL_V=$V round $round"; sed -n 2,3p $OUT/probe_xl_v${V}_$round.log; grep "xl" $OUT/probe_xl_v${V}_$round.log | head -6
done; done
