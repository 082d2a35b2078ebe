#!/bin/bash
# round 6: the weight gradient's round-aware split over positions (csrc/wgrad_x3.hip pick_chunks) against rounds 3-5's rule (AVT_WGRAD_ROUNDS=0):
# weight-gradient tests, then config 5 at 8 items and at one item (replayed), A/B interleaved, two processes each
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_wgrad_rounds
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_train_conv.py tests/test_gpu_train_step.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.log
for i in 1 2; do for r in 1 0; do
  AVT_WGRAD_ROUNDS=$r timeout 600 python bench.py --mode train --steps 4 --warmup 2 > $O/tmp.json 2> $O/err.log
  python3 -c "
import json
d=json.loads(open('$O/tmp.json').read().strip().splitlines()[-1]); print('8 items, rounds=$r run $i:', d.get('value'), d.get('ms_per_step'), d.get('loss_first_last'))"
  AVT_WGRAD_ROUNDS=$r timeout 600 python bench.py --mode train --train-items 1 --steps 12 --warmup 4 --train-graph 1 > $O/tmp.json 2> $O/err.log
  python3 -c "
import json
d=json.loads(open('$O/tmp.json').read().strip().splitlines()[-1]); print('one item graph, rounds=$r run $i:', d.get('value'), d.get('ms_per_step'))"
done; done | tee $O/ab.log
