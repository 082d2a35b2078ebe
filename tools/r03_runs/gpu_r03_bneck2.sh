#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03c
mkdir -p $OUT
cd $R
for TC in 4 8 16 32; do
  AVT_FUSE_TCHUNK_X3=$TC python tools/probe_x3.py f16x3 64 > $OUT/probe_tc$TC.log 2>&1
  echo "tchunk $TC"; sed -n 2,3p $OUT/probe_tc$TC.log; grep "fused bottleneck" $OUT/probe_tc$TC.log
done
