# bneck_x3 fragment-major / four-wave tilings: parity tests, then per-layer A/B on one box (AVT_BNECK_V = 3: round 3's tilings)
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_x3.py -x -q -m gpu -k "bneck or encoder_matches or contract" > gpurun_out/r04/tests_bneck.log 2>&1
tail -3 gpurun_out/r04/tests_bneck.log
for cfg in "AVT_BNECK_V=3 AVT_FUSE_TCHUNK_X3=8" "AVT_BNECK_V=4 AVT_FUSE_TCHUNK_X3=8" "AVT_BNECK_V=4 AVT_FUSE_TCHUNK_X3=11" "AVT_BNECK_V=4 AVT_FUSE_TCHUNK_X3=16" "AVT_BNECK_V=5 AVT_FUSE_TCHUNK_X3=11"; do
  echo "== $cfg" >> gpurun_out/r04/probe_bneck_ab.log
  env $cfg python tools/probe_x3.py f16x3 166 2>&1 | grep -E "batch=|fused bottleneck" >> gpurun_out/r04/probe_bneck_ab.log
done
cat gpurun_out/r04/probe_bneck_ab.log
