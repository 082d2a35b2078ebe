#!/bin/bash
# round 6, VERDICT r5 item 6, second run: x3 arithmetic, lr 0.1 -> 0.01 at epoch 250, until the reference's stop rule or 700 epochs
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r06_convergence
mkdir -p $O
timeout 1500 python tools/train_convergence.py --epochs 700 --lr-decay-epochs 250 --frames 260 --lr 0.1 --init default --modes x3 --roundtrip \
   --workdir /tmp/avt_stop_rule --out $O/train_stop_rule_x3_to_the_rule.json 2> $O/train_stop_rule_x3_to_the_rule.log | tail -2 > $O/x3_rule_brief.log
grep -a "epoch" $O/train_stop_rule_x3_to_the_rule.log | tail -3; cat $O/x3_rule_brief.log | cut -c1-600
