# bneck_x3 with the next frame requested after the first third of the a stage: tests, then per-layer A/B on one box
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_x3.py -x -q -m gpu -k "bneck" > gpurun_out/r04/tests_bneck.log 2>&1
tail -2 gpurun_out/r04/tests_bneck.log
: > gpurun_out/r04/probe_bneck_ab2.log
for cfg in "AVT_HIP_LIB=$PWD/audio-video-textures_amd/libavt_hip_base.so" "AVT_BNECK_V=3" "AVT_BNECK_V=4" "AVT_BNECK_V=6" "AVT_BNECK_V=4 AVT_FUSE_TCHUNK_X3=11" "AVT_BNECK_V=3 AVT_FUSE_TCHUNK_X3=11"; do
  echo "== $cfg" >> gpurun_out/r04/probe_bneck_ab2.log
  env $cfg python tools/probe_x3.py f16x3 166 2>&1 | grep -E "batch=|fused bottleneck" >> gpurun_out/r04/probe_bneck_ab2.log
done
cat gpurun_out/r04/probe_bneck_ab2.log
