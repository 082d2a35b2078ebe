#!/bin/bash
# round 6: config 5 across ranks with every rank's forward + backward as a replayed HIP graph and the gradient exchange outside the capture
# (bench.py --mode train --gpus N --train-graph 1): the two-rank gloo test, then 2 ranks sharing the GPU, graph form against DDP
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_graph_ranks
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -k "two_ranks_train" 2>&1 | tail -15 | tee $O/tests.log
for g in 1 0; do
  timeout 900 python bench.py --mode train --gpus 2 --dist-backend gloo --train-items 2 --steps 8 --warmup 3 --train-graph $g > $O/two_ranks_graph$g.json 2> $O/two_ranks_graph$g.err
  python3 -c "
import json
d=json.loads(open('$O/two_ranks_graph$g.json').read().strip().splitlines()[-1]); print('2 gloo ranks on one GPU, graph=$g:', d.get('value'), d.get('ms_per_step'), d.get('loss_first_last'), d.get('ranks_param_checksum_spread'), d['config']['parallelism'][:60])"
done 2>&1 | tee $O/ab.log
tail -5 $O/two_ranks_graph1.err
