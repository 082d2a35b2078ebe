#!/bin/bash
# round 6, VERDICT r5 item 6: the reference's stop rule (epoch loss < 0.07, main.py:475-477) on a 40-segment video: x3 arithmetic with
# the reference's StepLR shape (x 0.1 at epoch 250), up to 400 epochs; MIOpen fp32 for the first 60 epochs from the same seed
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r06_convergence
mkdir -p $O
timeout 1500 python tools/train_convergence.py --epochs 400 --lr-decay-epochs 250 --frames 260 --lr 0.1 --init default --modes x3 \
   --out $O/train_stop_rule_x3_lr_decay.json 2> $O/train_stop_rule_x3_lr_decay.log | tail -2 > $O/x3_brief.log
timeout 1500 python tools/train_convergence.py --epochs 60 --frames 260 --lr 0.1 --init default --modes fp32 --no-miopen-find \
   --out $O/train_stop_rule_fp32_60_epochs.json 2> $O/train_stop_rule_fp32.log | tail -2 > $O/fp32_brief.log
grep -a "epoch" $O/train_stop_rule_x3_lr_decay.log | tail -4; grep -a "epoch" $O/train_stop_rule_fp32.log | tail -3
