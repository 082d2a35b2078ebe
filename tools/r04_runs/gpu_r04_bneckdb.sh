# bneck_x3 with double-buffered a strips (one barrier per frame): parity tests, then the per-layer view
python -m pytest tests/test_gpu_x3.py -x -q -m gpu -k "bneck or encoder_matches or contract" 2>&1 | tail -2
for rep in 1 2; do python tools/probe_x3.py f16x3 166 table 2>&1 | grep -E "batch=|fused bottleneck"; done
