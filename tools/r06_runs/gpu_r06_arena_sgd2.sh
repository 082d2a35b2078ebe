#!/bin/bash
# round 6: the replayed one-item step with torch's fused SGD on / off, three processes each, interleaved (run-to-run spread of the graph's
# queue placement against the effect), after the accumulator test
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_arena_sgd2
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_train_conv.py -x -q -m gpu -k "micro_batch" 2>&1 | tail -3 | tee $O/tests.log
for i in 1 2 3; do for f in 1 0; do
  timeout 600 python bench.py --mode train --train-items 1 --steps 12 --warmup 4 --train-graph 1 --train-fused-sgd $f > $O/graph_fused${f}_$i.json 2> $O/err.log
  python3 -c "
import json
d=json.loads(open('$O/graph_fused${f}_$i.json').read().strip().splitlines()[-1]); print('graph fused=$f run $i', d.get('value'), d.get('ms_per_step'))"
done; done
