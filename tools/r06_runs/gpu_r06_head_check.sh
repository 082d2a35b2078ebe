#!/bin/bash
# round 6: after the grouped-planes gather: the failing convergence test with its output, the new tests, the config-5 legs
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r06_head
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_convergence.py -x -q -m gpu 2>&1 | tail -40 > $O/tests_convergence.log
timeout 900 python -m pytest tests/test_gpu_train_conv.py tests/test_gpu_train_step.py tests/test_gpu_bn_train.py tests/test_gpu_cli_train.py -x -q -m gpu 2>&1 | tail -5 > $O/tests_train.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "bf16x3" 2>&1 | tail -3 > $O/tests_sim.log
timeout 900 python bench.py --mode train --steps 6 --warmup 3 > $O/train_eight_items.json 2> $O/train8.err
timeout 600 python bench.py --mode train --train-items 1 --steps 12 --warmup 4 > $O/train_one_item_eager.json 2> $O/train1.err
timeout 600 python bench.py --mode train --train-items 1 --steps 12 --warmup 4 --train-graph 1 > $O/train_one_item_graph.json 2> $O/train1g.err
cat $O/tests_convergence.log | tail -30; cat $O/tests_train.log $O/tests_sim.log
for f in train_eight_items train_one_item_eager train_one_item_graph; do python -c "
import json,sys
d=json.load(open('$O/$f.json')); print('$f', round(d['value'],1), 'clips/s', round(d['ms_per_step'],2), 'ms', d['config'].get('hip_graph'))"; done
