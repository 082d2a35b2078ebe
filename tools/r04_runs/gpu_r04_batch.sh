# encoder batch sweep on one box (the frame table lifts the 4 GB limit of the packed fast clips)
for eb in 166 249 332 166 249 332; do
  python bench.py --no-cpu-baseline --no-train-leg --no-nxn-legs --no-precision-block --no-fast --enc-batch $eb 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('enc-batch $eb: %.1f clip-windows/s, %.1f ms/step' % (d['value'], d['ms_per_step']))"
done
