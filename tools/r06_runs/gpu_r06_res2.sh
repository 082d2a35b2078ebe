#!/bin/bash
# round 6: the fused slow-res2 bottleneck (csrc/res2_x3.hip): parity tests, then per-layer A/B at the production batch
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r06_res2
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_x3.py -x -q -m gpu -k "res2_x3" 2>&1 | grep -v amdgpu.ids | tail -15 > $O/tests_res2.log
cat $O/tests_res2.log
timeout 300 python tools/probe_x3.py f16x3 249 table res2=1 2>&1 | grep -v amdgpu.ids > $O/probe_x3_b249_res2_fused.log
timeout 300 python tools/probe_x3.py f16x3 249 table res2=0 2>&1 | grep -v amdgpu.ids > $O/probe_x3_b249_res2_unfused.log
head -3 $O/probe_x3_b249_res2_fused.log; grep -i "res2" $O/probe_x3_b249_res2_fused.log; head -3 $O/probe_x3_b249_res2_unfused.log
