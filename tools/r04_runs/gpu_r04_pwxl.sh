for skip in "" "320,128" "144,256" "256,64" "80,64" "512,128"; do
python - "$skip" <<'PY' 2>&1 | grep -E "== skip|batch=|cin320 cout128|cin144 cout256|cin256 cout64|cin80 cout64|cin512 cout128"
import sys, runpy
sys.path.insert(0, ".")
import avtex.fused_slowfast as f
if sys.argv[1]:
    a, b = sys.argv[1].split(",")
    f._PW_X3_SKIP = {(int(a), int(b))}
print("== skip", sys.argv[1])
sys.argv = ["probe_x3.py", "f16x3", "166", "table"]
runpy.run_path("tools/probe_x3.py", run_name="__main__")
PY
done
