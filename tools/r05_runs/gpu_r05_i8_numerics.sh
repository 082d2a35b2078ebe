#!/bin/bash
# the int8 cross-term numerics experiment (emulation only) on the three input sets
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
for inp in r04 r03 trained:300; do
  timeout 1500 python tools/experimental/probe_f16f8_numerics.py 256 $inp 2>&1 | grep -v amdgpu.ids | tail -9
done | tee gpurun_out/r05_i8_numerics.log
