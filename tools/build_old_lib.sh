#!/bin/bash
# A/B helper: build audio-video-textures_amd/libavt_hip_old.so from the csrc/ of another revision (default HEAD), next to the working
# tree's libavt_hip.so — the same ABI, selected at run time with AVT_HIP_LIB=$PWD/audio-video-textures_amd/libavt_hip_old.so.
# The tools/r04_runs/*_ab scripts (gpu_r04_epicf.sh, gpu_r04_xlrows.sh, gpu_r04_split8.sh) alternate the two libraries on ONE box.
# usage (here, before gpurun; the built .so travels with the snapshot and is git-ignored):  bash tools/build_old_lib.sh [rev]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
REV=${1:-HEAD}
D=$R/audio-video-textures_amd/csrc_old
rm -rf "$D" && mkdir "$D"
git -C "$R" archive "$REV" audio-video-textures_amd/csrc | tar -x -C "$D" --strip-components=2
make -C "$D" -j8 OUT=../libavt_hip_old.so > /dev/null
rm -rf "$D"
ls -la "$R/audio-video-textures_amd/libavt_hip_old.so"
