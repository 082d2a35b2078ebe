#!/bin/bash
# round 5: the general conv tile's K loop peeled (loads under a compile-time `more`: no s_waitcnt in front of each load pair of the
# fp32 form) + the 256 x 128 weight-gradient tile's loads really two steps ahead.  Parity tests, then old / new library interleaved
# on ONE box: the config-5 step stand-alone and a short headline run (tools/build_old_lib.sh HEAD built libavt_hip_old.so).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
L=gpurun_out/r05_peeled_ab.log; : > $L
timeout 900 python -m pytest tests/test_gpu_train_conv.py tests/test_gpu_x3.py tests/test_gpu_train_step.py -x -q 2>&1 | tail -4 | tee -a $L
OLD=$PWD/audio-video-textures_amd/libavt_hip_old.so
for rep in 1 2; do for lib in old new; do
  if [ $lib = old ]; then export AVT_HIP_LIB=$OLD; else unset AVT_HIP_LIB; fi
  timeout 600 python bench.py --mode train --steps 6 --warmup 2 2> gpurun_out/r05_peeled_train_$lib.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train $lib', d['value'], d['ms_per_step'], d['loss_first_last'])" | tee -a $L
done; done
for rep in 1 2; do for lib in old new; do
  if [ $lib = old ]; then export AVT_HIP_LIB=$OLD; else unset AVT_HIP_LIB; fi
  timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-fast --no-nxn-legs --no-train-leg --no-inputs-r03-leg --no-precision-block 2> gpurun_out/r05_peeled_head_$lib.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline $lib', d['value'], d['ms_per_step'], d['roofline']['achieved'])" | tee -a $L
done; done
unset AVT_HIP_LIB
tail -3 gpurun_out/r05_peeled_train_new.err
