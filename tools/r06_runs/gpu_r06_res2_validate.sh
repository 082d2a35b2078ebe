#!/bin/bash
# round 6: the shipped form of the fused slow-res2 kernel: parity tests, the contract on the same frames, per-layer probe, a short bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r06_res2
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_x3.py -x -q -m gpu -k "res2_x3 or contract_on_the_same_frames or pw_chain_x3 or encoder_matches" 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/tests_res2_final.log
timeout 300 python tools/probe_x3.py f16x3 249 table res2=1 2>&1 | grep -v amdgpu.ids > $O/probe_x3_b249_res2_fused.log
timeout 300 python tools/probe_x3.py f16x3 249 table res2=0 2>&1 | grep -v amdgpu.ids > $O/probe_x3_b249_res2_unfused.log
head -2 $O/probe_x3_b249_res2_fused.log; grep -i "res2" $O/probe_x3_b249_res2_fused.log; head -2 $O/probe_x3_b249_res2_unfused.log
timeout 900 python bench.py --steps 3 --warmup 1 --no-fast --no-train-leg --no-cpu-baseline --no-nxn-legs > $O/bench_short.json 2> $O/bench_short.err
tail -c 1800 $O/bench_short.json
