#!/bin/bash
# A/B on ONE box (box-to-box spread is +-3 %): bench.py under each environment setting given as arguments, twice, interleaved.
mkdir -p gpurun_out; : > gpurun_out/ab.log
for rep in 1 2; do
  for cfg in "$@"; do
    v=$(env $cfg python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): print(round(json.loads(l)['value'], 1))")
    echo "$cfg -> $v" >> gpurun_out/ab.log
  done
done
cat gpurun_out/ab.log
