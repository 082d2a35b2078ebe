#!/bin/bash
# round 6, second GPU call: new tests (ADVICE r5 fixes, th-0.0 tie pin), short A/B probes, the stop-rule convergence run
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
O=gpurun_out/r06_batch2
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_conv.py -x -q -m gpu -k "pixel_grouped_statistics or releases_what or epilogue" 2>&1 | tail -5 > $O/tests_advice.log
timeout 600 python -m pytest tests/test_gpu_x3.py -x -q -m gpu -s -k "th0_exact_ties" 2>&1 | grep -v amdgpu.ids | tail -6 > $O/tests_th0.log
timeout 300 python tools/probe_select_wide.py 2>&1 | grep -v amdgpu.ids > $O/probe_select_wide.log
timeout 300 python tools/probe_x3.py f16x3 249 table 2>&1 | grep -v amdgpu.ids > $O/probe_x3_b249.log
timeout 300 python tools/probe_x3.py f16x3 249 table pwskip=256x1024 2>&1 | grep -v amdgpu.ids > $O/probe_x3_b249_pw256x1024_on_xl.log
timeout 300 python tools/probe_x3.py f16x3 241 table 2>&1 | grep -v amdgpu.ids | head -3 > $O/probe_x3_b241.log
timeout 1500 python tools/train_convergence.py --epochs 60 --frames 260 --lr 0.1 --init default --modes x3 fp32 \
   --out $O/train_stop_rule.json 2> $O/train_stop_rule.log | tail -2 > $O/train_stop_rule_brief.log
