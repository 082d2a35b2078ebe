# training step: few-channel convolutions pixel-grouped (train_ops._GROUP 1 | 0): parity tests, then bench --mode train A/B on one box
python -m pytest tests/test_gpu_train_conv.py tests/test_gpu_train_step.py -x -q -m gpu 2>&1 | tail -3
for flag in 1 0 1 0; do
python - $flag <<'PY' 2>/dev/null | tail -1
import sys, runpy, json, io, contextlib
sys.path.insert(0, ".")
import avtex.train_ops as t
t._GROUP = int(sys.argv[1])
sys.argv = ["bench.py", "--mode", "train", "--steps", "3", "--warmup", "2"]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path("bench.py", run_name="__main__")
d = json.loads([l for l in buf.getvalue().splitlines() if l.startswith("{")][-1])
print("pixel-grouped few-channel layers %d: %.1f clips/s, %.1f ms/step, loss %s" % (t._GROUP, d["value"], d["ms_per_step"], d["loss_first_last"]))
PY
done
