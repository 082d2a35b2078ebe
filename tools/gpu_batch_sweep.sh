#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for B in; do
  python bench.py --steps 1 --warmup 1 --no-cpu-baseline --enc-batch $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('enc_batch', $B, 'windows/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],1))"
done
for S in 1 2 4; do for B in 64 128; do
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --enc-batch $B --streams $S 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams', $S, 'enc_batch', $B, 'windows/s', round(d['value'],1))"
done; done
