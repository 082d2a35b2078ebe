# pw_x3 with the weight-fragment reads PF steps ahead of their MFMA triple: parity tests, then per-layer A/B on one box
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_x3.py -x -q -m gpu -k "pw_x3 or encoder_matches or pw_chain" > gpurun_out/r04/tests_pw.log 2>&1
tail -2 gpurun_out/r04/tests_pw.log
: > gpurun_out/r04/probe_pw_prefetch_ab.log
for cfg in "AVT_PW_PF=0" "AVT_PW_PF=1" "AVT_PW_PF=2" "AVT_PW_PF=3" "AVT_PW_PF=0" "AVT_PW_PF=2"; do
  echo "== $cfg" >> gpurun_out/r04/probe_pw_prefetch_ab.log
  env $cfg python tools/probe_x3.py f16x3 166 2>&1 | grep -E "batch=|pointwise|fused bottleneck cin128" >> gpurun_out/r04/probe_pw_prefetch_ab.log
done
cat gpurun_out/r04/probe_pw_prefetch_ab.log
