# same-box 2 x 2: bench inputs (round 3's | round 4's sharper ones) x packing (dense per-window clips | frame table), twice
mkdir -p gpurun_out/r04
for rep in 1 2; do for inp in r03 r04; do for flag in 0 1; do
  python - $flag $inp <<'PY' 2>/dev/null | tail -1
import sys, runpy, json, io, contextlib
sys.path.insert(0, ".")
import avtex.texture as t
t.FRAME_TABLE = bool(int(sys.argv[1]))
inp = sys.argv[2]
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-train-leg", "--no-nxn-legs", "--no-precision-block", "--no-fast", "--inputs", inp]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path("bench.py", run_name="__main__")
d = json.loads([l for l in buf.getvalue().splitlines() if l.startswith("{")][-1])
print("inputs %s, frame table %d: %.1f clip-windows/s, %.1f ms/step, XL tile %.0f TFLOP/s" % (inp, t.FRAME_TABLE, d["value"], d["ms_per_step"], d["roofline"]["achieved"]))
PY
done; done; done
