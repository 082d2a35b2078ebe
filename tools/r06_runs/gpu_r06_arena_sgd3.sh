#!/bin/bash
# round 6: after the optimizer-step hook (fused optimizers do not move Tensor._version) and the arena views with the weights' own strides:
# tests, then the replayed one-item step with {arena on, off} x {fused SGD, foreach SGD}, two processes each; eager one item; 8 items
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_arena_sgd3
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_train_conv.py tests/test_gpu_train_step.py -x -q -m gpu -k "micro_batch or optimizer or graphed or stale" 2>&1 | tail -4 | tee $O/tests.log
python tools/experimental/debug_sgd_arena.py 2>&1 | grep -v amdgpu.ids | tail -9 | tee $O/debug_sgd_arena.log
for i in 1 2; do for f in 1 0; do for a in "" "--no-grad-accumulator"; do
  timeout 600 python bench.py --mode train --train-items 1 --steps 12 --warmup 4 --train-graph 1 --train-fused-sgd $f $a > $O/tmp.json 2> $O/err.log
  python3 -c "
import json
d=json.loads(open('$O/tmp.json').read().strip().splitlines()[-1]); print('graph fused=$f arena=%s run $i' % ('off' if '$a' else 'on'), d.get('value'), d.get('ms_per_step'), d.get('loss_first_last'))"
done; done; done 2>&1 | tee $O/ab.log
timeout 600 python bench.py --mode train --train-items 1 --steps 12 --warmup 4 > $O/eager.json 2> $O/err.log
timeout 600 python bench.py --mode train --steps 3 --warmup 2 > $O/eight.json 2> $O/err.log
timeout 600 python bench.py --mode train --steps 3 --warmup 2 --train-fused-sgd 0 > $O/eight_foreach.json 2> $O/err.log
for f in eager eight eight_foreach; do python3 -c "
import json
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', d.get('value'), d.get('ms_per_step'), d.get('loss_first_last'))"; done | tee -a $O/ab.log
