#!/bin/bash
# round 6: the IO32 convolutions' 64-row tile at small batches (csrc/conv_x3.hip io32_tile_rows): conv / BatchNorm / step tests, then config 5
# at one item with the tile on and off (AVT_SMALL_TILE=0 -> bench.py switches it off), and at 8 items (unaffected by construction)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_small_tile
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_train_conv.py tests/test_gpu_train_step.py tests/test_gpu_bn_train.py tests/test_gpu_train_convergence.py -x -q -m gpu 2>&1 | tail -8 | tee $O/tests.log
for st in 1 0; do
  AVT_SMALL_TILE=$st timeout 600 python bench.py --mode train --train-items 1 --steps 12 --warmup 4 --train-graph 1 > $O/graph_small$st.json 2> $O/graph_small$st.err
  AVT_SMALL_TILE=$st timeout 600 python bench.py --mode train --train-items 1 --steps 12 --warmup 4 > $O/eager_small$st.json 2> $O/eager_small$st.err
done
timeout 600 python bench.py --mode train --steps 3 --warmup 2 > $O/train_eight_items.json 2> $O/eight.err
for f in graph_small1 graph_small0 eager_small1 eager_small0 train_eight_items; do python3 -c "
import json
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', d.get('value'), d.get('ms_per_step'))"; done
