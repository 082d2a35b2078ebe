#!/bin/bash
# round 6, third scripted GPU call: th-0.0 tie pin (with its report), wide-row select timing, the 8-rank gloo readiness test,
# is the one-item config-5 step host-bound?, the stop-rule convergence runs (x3 to the rule or 250 epochs; fp32 60 epochs)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r06_batch3
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_x3.py -x -q -m gpu -s -k "th0_exact_ties" > $O/tests_th0_full.log 2>&1
grep -a "TH0-TIES\|passed\|failed" $O/tests_th0_full.log | cut -c1-1800 > $O/tests_th0.log
timeout 300 python tools/probe_select_wide.py 2>&1 | grep -v amdgpu.ids > $O/probe_select_wide.log
timeout 1500 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -k "eight_ranks" 2>&1 | tail -15 > $O/tests_eight_ranks.log
timeout 600 python tools/experimental/probe_train_host_bound.py --train-items 1 --steps 12 --warmup 4 2>&1 | tail -2 > $O/host_bound_one_item.log
timeout 600 python tools/experimental/probe_train_host_bound.py --steps 4 --warmup 3 2>&1 | tail -2 > $O/host_bound_eight_items.log
timeout 1200 python tools/train_convergence.py --epochs 250 --frames 260 --lr 0.1 --init default --modes x3 \
   --out $O/train_stop_rule_x3.json 2> $O/train_stop_rule_x3.log | tail -2 > $O/train_stop_rule_x3_brief.log
timeout 900 python tools/train_convergence.py --epochs 60 --frames 260 --lr 0.1 --init default --modes fp32 \
   --out $O/train_stop_rule_fp32.json 2> $O/train_stop_rule_fp32.log | tail -2 > $O/train_stop_rule_fp32_brief.log
rm -f $O/tests_th0_full.log
