#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03e
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_x3.py -x -q -k "conv_x3_matches or encoder_matches" 2>&1 | tail -15 > $OUT/tests.log
tail -4 $OUT/tests.log
python tools/probe_x3.py f16x3 64 > $OUT/probe_xl.log 2>&1
AVT_CONV_X3_XL=0 python tools/probe_x3.py f16x3 64 > $OUT/probe_noxl.log 2>&1
echo XL; sed -n 2,3p $OUT/probe_xl.log; grep "tile\|xl" $OUT/probe_xl.log | head -24
echo NOXL; sed -n 2,3p $OUT/probe_noxl.log; grep "tile" $OUT/probe_noxl.log | head -22
