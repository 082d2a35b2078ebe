#!/bin/bash
# Phase-skip diagnostic of the contract-grade conv tile: the same layer with the LDS stage (1), the MFMAs (2) or the global
# loads (4) or the LDS fragment reads (8) compiled OUT — what the time hangs on is what it gets faster without.  One diagnostic library per setting
# (libavt_hip_dbg<n>.so: the stamp build with its stamps off and -DAVT_DBG_CONST=n; the shipped library has no hooks).
#   build (here or on the box):  bash tools/probe_conv_phases.sh build
#   run on the GPU box:          bash tools/probe_conv_phases.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
C=audio-video-textures_amd/csrc
if [ "$1" = build ]; then
  rm -rf $C/stamp && make -C $C stamp STAMP_EXTRA=-DAVT_STAMP_OFF -j8 > /dev/null 2>&1 || exit 1
  for N in 0 1 2 4 6 7 8 13; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -DAVT_CONV_STAMP -DAVT_STAMP_OFF -DAVT_DBG_CONST=$N \
      -c $C/conv_x3.hip -o $C/stamp/conv_x3_dbg$N.o || exit 1
    OBJS=$(ls $C/stamp/*.o | grep -v "conv_x3")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $C/stamp/conv_x3_dbg$N.o -o audio-video-textures_amd/libavt_hip_dbg$N.so || exit 1
  done
  exit 0
fi
for L in "1024 256 3 1 1 64 8 14 14" "256 256 1 3 3 64 8 14 14" "64 64 1 3 3 64 8 56 56"; do
  for N in 0 1 2 4 6 7 8 13; do
    echo -n "skip=$N  "; AVT_HIP_LIB=$PWD/audio-video-textures_amd/libavt_hip_dbg$N.so PRECISION=f16x3 python tools/conv_layer_bench.py $L 2>/dev/null | tail -1
  done
done
