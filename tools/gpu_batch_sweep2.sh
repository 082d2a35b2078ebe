#!/bin/bash
# Encoder batch sizes chosen for whole rounds of XL tiles: res4 has 6.125*B tiles, res5 3.06*B.
mkdir -p gpurun_out
for b in ${BATCHES:-83 125 128 167}; do
  echo "== enc-batch $b" >> gpurun_out/batch_sweep2.log
  python bench.py --no-cpu-baseline --enc-batch $b 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l); print(j['value'], j['ms_per_step'], j['roofline']['achieved'])
" >> gpurun_out/batch_sweep2.log
done
cat gpurun_out/batch_sweep2.log
