#!/bin/bash
# round 6: config 5 at its real per-rank shape (one item = 16 clips per step): eager against ONE replayed HIP graph
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r06_graph
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_train_step.py -x -q -m gpu -k "graphed_step" 2>&1 | tail -12 | tee $O/tests_graph.log
timeout 600 python bench.py --mode train --train-items 1 --steps 12 --warmup 4 > $O/train_one_item_eager.json 2> $O/train_one_item_eager.err
timeout 600 python bench.py --mode train --train-items 1 --steps 12 --warmup 4 --train-graph 1 > $O/train_one_item_graph.json 2> $O/train_one_item_graph.err
tail -c 600 $O/train_one_item_eager.json; echo; tail -c 900 $O/train_one_item_graph.json; echo; grep -a "train-graph\|Error\|error" $O/train_one_item_graph.err | head -5
