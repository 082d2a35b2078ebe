# regenerates the experiment logs profiles/r04/README.md cites (one box)
O=gpurun_out/r04logs; mkdir -p $O
( echo "# tools/probe_survivors.py 512 <sparsity ...>: round 3's synthetic encoders (two independent random-init SlowFasts)"; python tools/probe_survivors.py 512 0.5 1.0 2.0 3.0 2>&1 | grep -v amdgpu.ids
  echo "# t encoder = perturbed copy of q (REL), scenes with their own colour layout (VARIETY=1), weak residual branches (BRANCH)"
  VARIETY=1 NCAL=8 BRANCH=0.1 REL=0.05 python tools/probe_survivors.py 512 1.0 2.0 2.5 2>&1 | grep -v amdgpu.ids
  VARIETY=0 NCAL=8 BRANCH=1.0 REL=0.05 python tools/probe_survivors.py 512 0.5 2.0 2>&1 | grep -v amdgpu.ids ) > $O/probe_survivors_sweep.log
timeout 300 python tools/experimental/probe_cumask.py 166 4 2>&1 | grep -v amdgpu.ids > $O/cu_mask_streams_ab.log
( python tools/experimental/probe_f16f8_numerics.py 256 r04 2>&1 | grep -v amdgpu.ids; python tools/experimental/probe_f16f8_numerics.py 256 r03 2>&1 | grep -v amdgpu.ids ) > $O/f16f8_numerics_experiment.log
bash tools/r04_runs/gpu_r04_merge.sh 2>&1 | grep -v "passed\|^\.\." > $O/stem_merged_taps_ab.log
( for skip in "" "512,128" "320,128" "144,256" "256,64" "80,64" "256,1024" "128,512"; do
python - "$skip" <<'PY' 2>&1 | grep -E "== on|batch=|pointwise|cin512 cout128|cin320 cout128|cin144 cout256|cin256 cout64 k\(1|cin80 cout64|cin256 cout1024|cin128 cout512"
import sys, runpy
sys.path.insert(0, ".")
import avtex.fused_slowfast as f
f._PW_X3_SKIP = set()
if sys.argv[1]:
    a, b = sys.argv[1].split(",")
    f._PW_X3_SKIP = {(int(a), int(b))}
print("== on the general / 256 x 256 tile instead of pw_x3:", sys.argv[1] or "(none)")
sys.argv = ["probe_x3.py", "f16x3", "166", "table"]
runpy.run_path("tools/probe_x3.py", run_name="__main__")
PY
done ) > $O/pw_general_tile_sweep.log
ls -la $O
