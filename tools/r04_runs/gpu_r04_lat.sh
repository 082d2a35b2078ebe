# lateral connections on the streaming kernel's temporal-tap form: parity, then per-layer A/B (fused_slowfast._LATERAL_X3 1 | 0)
python -m pytest tests/test_gpu_x3.py -x -q -m gpu -k "encoder_matches or contract or pw_x3" 2>&1 | tail -3
for flag in 1 0; do
python - $flag <<'PY' 2>&1 | grep -E "== lateral|batch=|lateral|k\(7, 1, 1\)"
import sys, runpy
sys.path.insert(0, ".")
import avtex.fused_slowfast as f
f._LATERAL_X3 = int(sys.argv[1])
print("== lateral on the streaming kernel:", f._LATERAL_X3)
sys.argv = ["probe_x3.py", "f16x3", "166", "table"]
runpy.run_path("tools/probe_x3.py", run_name="__main__")
PY
done
