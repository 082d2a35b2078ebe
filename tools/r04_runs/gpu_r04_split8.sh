# split8 / split4: the fp16-plane split with ONE wave-wide range test per 8 / 4 values and the plain convert-subtract-convert form when it
# passes.  The x3 parity suite (incl. the NaN / inf / range-edge tests), then per-layer probe, bench and the training bench, new / old
# library on one box (old = audio-video-textures_amd/libavt_hip_old.so built from the previous commit's csrc)
python -m pytest tests/test_gpu_x3.py tests/test_gpu_conv.py tests/test_gpu_kernels.py tests/test_gpu_train_conv.py -x -q -m gpu 2>&1 | tail -2
mkdir -p gpurun_out/split8b
for lib in new old new old; do
  if [ $lib = old ]; then export AVT_HIP_LIB=$PWD/audio-video-textures_amd/libavt_hip_old.so; else unset AVT_HIP_LIB; fi
  python tools/probe_x3.py f16x3 249 table > gpurun_out/split8b/probe_${lib}_$RANDOM.log 2>&1
done
for f in gpurun_out/split8b/probe_*.log; do echo "== $f"; sed -n 2,3p $f | cut -c1-150; grep -E "bottleneck|maxpool" $f | cut -c1-140; done
for lib in new old new old; do
  if [ $lib = old ]; then export AVT_HIP_LIB=$PWD/audio-video-textures_amd/libavt_hip_old.so; else unset AVT_HIP_LIB; fi
  echo "== bench $lib"; python bench.py --no-fast --no-train-leg --no-cpu-baseline --no-nxn-legs --no-precision-block 2>/dev/null | cut -c1-200
done
for lib in new old new old; do
  if [ $lib = old ]; then export AVT_HIP_LIB=$PWD/audio-video-textures_amd/libavt_hip_old.so; else unset AVT_HIP_LIB; fi
  echo "== train $lib"; python bench.py --mode train --steps 3 --warmup 2 2>/dev/null | cut -c1-200
done
