#!/bin/bash
# round 5: the pipelined weight-gradient tile with buffer loads (hardware zero fill instead of mask registers and selects), the
# pointwise fast path of its position decode, its 64-wide form — old (HEAD's csrc: tools/build_old_lib.sh) / new library on ONE box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
L=gpurun_out/r05_wgrad_valu_ab.log; : > $L
timeout 900 python -m pytest tests/test_gpu_train_conv.py tests/test_gpu_train_step.py -x -q 2>&1 | tail -3 | tee -a $L
OLD=$PWD/audio-video-textures_amd/libavt_hip_old.so
for rep in 1 2; do for lib in old new; do
  if [ $lib = old ]; then export AVT_HIP_LIB=$OLD; else unset AVT_HIP_LIB; fi
  timeout 600 python bench.py --mode train --steps 6 --warmup 2 2> gpurun_out/r05_wgrad_valu_$lib.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train $lib', d['value'], d['ms_per_step'], d['loss_first_last'])" | tee -a $L
done; done
unset AVT_HIP_LIB
