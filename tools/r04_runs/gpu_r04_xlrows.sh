# XL tile epilogue in 64-row slabs staged by four waves (one per SIMD) instead of 64-column slabs staged by two waves of one SIMD:
# parity tests, phase stamps, per-layer probe and bench A/B on one box against a library with the previous conv_x3.hip
python -m pytest tests/test_gpu_x3.py tests/test_gpu_conv.py tests/test_gpu_train_conv.py tests/test_gpu_interp.py -x -q -m gpu 2>&1 | tail -2
mkdir -p gpurun_out/xlrows
for lib in new old new old; do
  if [ $lib = old ]; then export AVT_HIP_LIB=$PWD/audio-video-textures_amd/libavt_hip_old.so; else unset AVT_HIP_LIB; fi
  python tools/probe_x3.py f16x3 249 table > gpurun_out/xlrows/probe_${lib}_$RANDOM.log 2>&1
done
for f in gpurun_out/xlrows/probe_*.log; do echo "== $f"; sed -n 2,3p $f | cut -c1-150; grep -E "conv_x3_xl_kernel" $f | cut -c1-150; done
for lib in new old new old; do
  if [ $lib = old ]; then export AVT_HIP_LIB=$PWD/audio-video-textures_amd/libavt_hip_old.so; else unset AVT_HIP_LIB; fi
  echo "== bench $lib"; python bench.py --no-fast --no-train-leg --no-cpu-baseline --no-nxn-legs --no-precision-block 2>/dev/null | cut -c1-200
done
unset AVT_HIP_LIB
PRECISION=f16x3 SHAPES="512 2048 1 1 1 249 8 7 7 res;2048 512 3 1 1 249 8 7 7" bash tools/probe_stamps.sh 2>&1 | grep -v amdgpu.ids
