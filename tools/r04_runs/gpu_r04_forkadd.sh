# training convolutions (IO32 tiles): the fp32 `add` operand of conv3d_fork requested up front instead of chunk by chunk between the stores.
# Parity tests, then bench --mode train, new / old library alternating on one box
python -m pytest tests/test_gpu_train_conv.py tests/test_gpu_train_step.py -x -q -m gpu 2>&1 | tail -1
for lib in new old new old new old; do
  if [ $lib = old ]; then export AVT_HIP_LIB=$PWD/audio-video-textures_amd/libavt_hip_old.so; else unset AVT_HIP_LIB; fi
  echo "== train $lib"; python bench.py --mode train --steps 3 --warmup 2 2>/dev/null | cut -c1-215 | cut -c125-215
done
