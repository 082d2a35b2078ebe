# frame-table packing: parity test, then bench A/B on one box (texture.FRAME_TABLE on / off through a tiny driver)
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_x3.py -x -q -m gpu -k "frame_table or stem_x3 or contract" > gpurun_out/r04/tests_table.log 2>&1
tail -3 gpurun_out/r04/tests_table.log
for flag in 1 0 1 0; do
  python - $flag <<'PY' 2>/dev/null | tail -1
import sys, runpy
sys.path.insert(0, ".")
import avtex.texture as t
t.FRAME_TABLE = bool(int(sys.argv[1]))
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-train-leg", "--no-nxn-legs", "--no-precision-block", "--no-fast"]
import json, io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path("bench.py", run_name="__main__")
d = json.loads([l for l in buf.getvalue().splitlines() if l.startswith("{")][-1])
print("FRAME_TABLE=%d: %.1f clip-windows/s, %.1f ms/step" % (t.FRAME_TABLE, d["value"], d["ms_per_step"]))
PY
done
python tools/probe_x3.py f16x3 166 2>&1 | grep -E "batch=|stem|maxpool"; python tools/probe_x3.py f16x3 166 table 2>&1 | grep -E "batch=|stem|maxpool|Error|error"
