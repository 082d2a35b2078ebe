#!/bin/bash
# round 6: single-pass gradient arena + torch's fused SGD in bench.py's training legs: the accumulator test, then config 5 at one item
# (replayed graph, eager) and at 8 items, with --train-fused-sgd 1 / 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_arena_sgd
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_train_conv.py tests/test_gpu_train_step.py -x -q -m gpu -k "micro_batch or graphed or stale or config5" 2>&1 | tail -5 | tee $O/tests.log
for f in 1 0; do
  timeout 600 python bench.py --mode train --train-items 1 --steps 12 --warmup 4 --train-graph 1 --train-fused-sgd $f > $O/graph_fused$f.json 2> $O/graph_fused$f.err
done
timeout 600 python bench.py --mode train --train-items 1 --steps 12 --warmup 4 > $O/eager_fused1.json 2> $O/eager_fused1.err
timeout 600 python bench.py --mode train --steps 3 --warmup 2 > $O/train_eight_items.json 2> $O/eight.err
for f in graph_fused1 graph_fused0 eager_fused1 train_eight_items; do python3 -c "
import json
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', d.get('value'), d.get('ms_per_step'), d.get('loss_first_last'))"; done
tail -3 $O/graph_fused1.err
