#!/bin/bash
# round 6: every weight's planes of a training step by ONE launch (avt_weight_planes_multi) + the graphed step's loss read one step late:
# the training tests, then config 5 at one item (eager, replayed graph) and at 8 items
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_planes_multi
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_train_conv.py tests/test_gpu_train_step.py tests/test_gpu_bn_train.py tests/test_gpu_cli_train.py tests/test_gpu_train_convergence.py -x -q -m gpu 2>&1 | tail -8 | tee $O/tests.log
timeout 600 python bench.py --mode train --train-items 1 --steps 12 --warmup 4 > $O/train_one_item_eager.json 2> $O/eager.err
timeout 600 python bench.py --mode train --train-items 1 --steps 12 --warmup 4 --train-graph 1 > $O/train_one_item_graph.json 2> $O/graph.err
timeout 600 python bench.py --mode train --steps 3 --warmup 2 > $O/train_eight_items.json 2> $O/eight.err
for f in train_one_item_eager train_one_item_graph train_eight_items; do python3 -c "
import json
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', d.get('value'), d.get('ms_per_step'), d.get('train_hip_graph'))"; done
tail -3 $O/graph.err
