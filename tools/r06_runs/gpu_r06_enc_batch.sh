#!/bin/bash
# round 6: clips per encoder launch — 249 (the default since round 4) against 332 and 415 (4 and 5 rounds of 83), headline leg alone
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_enc_batch
mkdir -p $O
for b in 249 332 415 249; do
  timeout 600 python bench.py --steps 2 --warmup 1 --enc-batch $b --no-fast --no-r03-leg --no-train-leg --no-cpu-baseline --no-precision-block --no-nxn-legs > $O/b$b.json 2> $O/err.log
  python3 -c "
import json
d=json.loads(open('$O/b$b.json').read().strip().splitlines()[-1]); print('enc-batch $b:', d.get('value'), d.get('ms_per_step'), d['roofline']['frac'])"
done | tee $O/enc_batch.log
