# conv_x3 tiles: bias / scale of a tile staged in LDS by the prologue instead of fetched per accumulator group in the epilogue.
# Parity tests, then the per-layer probe and the bench A/B on one box against a library built with the previous conv_x3.hip.
python -m pytest tests/test_gpu_x3.py tests/test_gpu_conv.py tests/test_gpu_train_conv.py -x -q -m gpu 2>&1 | tail -2
mkdir -p gpurun_out/epicf
for lib in new old new old; do
  if [ $lib = old ]; then export AVT_HIP_LIB=$PWD/audio-video-textures_amd/libavt_hip_old.so; else unset AVT_HIP_LIB; fi
  python tools/probe_x3.py f16x3 249 table > gpurun_out/epicf/probe_${lib}_$RANDOM.log 2>&1
done
for f in gpurun_out/epicf/probe_*.log; do echo "== $f"; sed -n 2,3p $f | cut -c1-150; grep -E "conv_x3_xl_kernel|tile<" $f | cut -c1-150; done
for lib in new old new old; do
  if [ $lib = old ]; then export AVT_HIP_LIB=$PWD/audio-video-textures_amd/libavt_hip_old.so; else unset AVT_HIP_LIB; fi
  echo "== bench $lib"; python bench.py --no-fast --no-train-leg --no-cpu-baseline --no-nxn-legs --no-precision-block 2>/dev/null | cut -c1-200
done
