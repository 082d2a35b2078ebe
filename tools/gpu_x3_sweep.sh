#!/bin/bash
# On the GPU box: the contract-grade bench leg alone (no fast leg, no CPU baseline / precision / NxN blocks) over clip batch and
# encoder streams — one box, so the lines compare.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for CFG in "--enc-batch 128 --streams 2" "--enc-batch 64 --streams 2" "--enc-batch 96 --streams 2" "--enc-batch 160 --streams 2" "--enc-batch 128 --streams 1" "--enc-batch 128 --streams 4"; do
  python bench.py --no-fast --no-cpu-baseline --no-precision-block --no-nxn-legs --steps 2 --warmup 1 $CFG 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$CFG', round(d['value'],1), 'windows/s', round(d['ms_per_step'],1), 'ms/step')"
done
