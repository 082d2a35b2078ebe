#!/bin/bash
# Phase-skip diagnostic of the contract-grade pointwise kernel (csrc/pw_x3.hip): one library per setting with the stores (1),
# the MFMAs (2) or the residual loads (4) compiled out.   build: bash tools/probe_pw_phases.sh build   run (GPU box): bash tools/probe_pw_phases.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
C=audio-video-textures_amd/csrc
if [ "$1" = build ]; then
  make -C $C -j8 > /dev/null 2>&1 || exit 1
  for N in 0 1 2 4 7; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -DAVT_PW_DBG_CONST=$N -c $C/pw_x3.hip -o /tmp/pw_x3_dbg$N.o || exit 1
    OBJS=$(ls $C/*.o | grep -v "pw_x3.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/pw_x3_dbg$N.o -o audio-video-textures_amd/libavt_hip_pwdbg$N.so || exit 1
  done
  exit 0
fi
for L in "64 256 1 1 1 64 8 56 56 res" "256 64 1 1 1 64 8 56 56" "128 512 1 1 1 64 8 28 28 res" "256 1024 1 1 1 64 8 14 14 res"; do
  for N in 0 1 2 4 7; do
    echo -n "skip=$N  "; AVT_HIP_LIB=$PWD/audio-video-textures_amd/libavt_hip_pwdbg$N.so PRECISION=f16x3 python tools/conv_layer_bench.py $L 2>/dev/null | tail -1
  done
done
