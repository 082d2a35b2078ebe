# fast stem with merged frame taps: parity test, per-layer probe, bench A/B (fused_slowfast._STEM_MERGE 1 | 0) on one box
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_x3.py -x -q -m gpu -k "frame_table or stem_x3 or contract" > gpurun_out/r04/tests_merge.log 2>&1
tail -3 gpurun_out/r04/tests_merge.log
python tools/probe_x3.py f16x3 166 table 2>&1 | grep -E "batch=|stem"
for flag in 1 0 1 0; do
  python - $flag <<'PY' 2>/dev/null | tail -1
import sys, runpy, json, io, contextlib
sys.path.insert(0, ".")
import avtex.fused_slowfast as f
f._STEM_MERGE = int(sys.argv[1])
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-train-leg", "--no-nxn-legs", "--no-fast"]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path("bench.py", run_name="__main__")
d = json.loads([l for l in buf.getvalue().splitlines() if l.startswith("{")][-1])
print("merged taps %d: %.1f clip-windows/s, %.1f ms/step; precision max |dscore| %.2e, frames lists %s" % (
    f._STEM_MERGE, d["value"], d["ms_per_step"], d["precision_max_abs_dscore"], d["frames_lists_identical"]))
PY
done
