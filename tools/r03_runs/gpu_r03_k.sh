#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03k
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_x3.py -x -q -k "encoder_matches or contract_on" 2>&1 | tail -5
python tools/probe_x3.py f16x3 83 > $OUT/probe.log 2>&1
AVT_FUSE_KCAT=0 python tools/probe_x3.py f16x3 83 > $OUT/probe_nokcat.log 2>&1
AVT_CONV_X3_XL_MINK=256 python tools/probe_x3.py f16x3 83 > $OUT/probe_mink256.log 2>&1
for f in probe probe_nokcat probe_mink256; do echo $f; sed -n 2,3p $OUT/$f.log; grep "fused bottleneck cin128 c128\|cin144\|cin80 cout256\|cin64 cout256 rows 2082304\|cin320 cout512" $OUT/$f.log; done
