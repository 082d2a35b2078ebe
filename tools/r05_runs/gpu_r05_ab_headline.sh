#!/bin/bash
# generic same-box A/B of the headline leg: libavt_hip_old.so (tools/build_old_lib.sh HEAD) against the working tree's library
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
L=gpurun_out/r05_ab_headline.log; : > $L
[ -n "$1" ] && timeout 900 python -m pytest $1 -x -q 2>&1 | tail -2 | tee -a $L
OLD=$PWD/audio-video-textures_amd/libavt_hip_old.so
for rep in 1 2 3; do for lib in old new; do
  if [ $lib = old ]; then export AVT_HIP_LIB=$OLD; else unset AVT_HIP_LIB; fi
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast --no-nxn-legs --no-train-leg --no-inputs-r03-leg --no-precision-block 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline $lib', d['value'], d['ms_per_step'], d['roofline']['achieved'])" | tee -a $L
done; done
unset AVT_HIP_LIB
