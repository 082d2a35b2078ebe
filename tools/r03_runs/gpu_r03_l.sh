#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03l
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_x3.py -x -q -k "conv_x3_matches or encoder_matches" 2>&1 | tail -3
python tools/probe_x3.py f16x3 83 > $OUT/probe.log 2>&1
sed -n 2,3p $OUT/probe.log; grep "xl" $OUT/probe.log | head -14
bash tools/probe_stamps_xl.sh 2>&1 | grep -v amdgpu.ids > $OUT/stamps.log; cat $OUT/stamps.log
