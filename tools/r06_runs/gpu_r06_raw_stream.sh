#!/bin/bash
# round 6: the launch paths take the current stream's raw handle (torch._C._cuda_getCurrentRawStream) instead of building a Stream object per
# launch: stream-sensitive tests, then the eager one-item step (host-bound) twice, the replayed one, and the headline leg
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_raw_stream
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_train_conv.py tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.log
for i in 1 2; do
  timeout 600 python bench.py --mode train --train-items 1 --steps 12 --warmup 4 > $O/eager$i.json 2> $O/err.log
  python3 -c "
import json
d=json.loads(open('$O/eager$i.json').read().strip().splitlines()[-1]); print('eager one item run $i:', d.get('value'), d.get('ms_per_step'))"
done | tee $O/ab.log
timeout 600 python bench.py --mode train --train-items 1 --steps 12 --warmup 4 --train-graph 1 > $O/graph.json 2> $O/err.log
timeout 600 python bench.py --steps 2 --warmup 1 --no-fast --no-r03-leg --no-train-leg --no-cpu-baseline --no-precision-block --no-nxn-legs > $O/headline.json 2> $O/err.log
for f in graph headline; do python3 -c "
import json
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f:', d.get('value'), d.get('ms_per_step'))"; done | tee -a $O/ab.log
