#!/bin/bash
# layer-selective pass count, numerics emulation only, three input sets
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
for inp in r04 r03 trained:300; do
  timeout 1200 python tools/experimental/probe_layer_npass_numerics.py 256 $inp 2>&1 | grep -v amdgpu.ids | tail -24
done | tee gpurun_out/r06_npass_numerics.log
