#!/bin/bash
# TIMING-ONLY experiment (garbage results): what full-cache-line operand fetches would buy the XL tile.
# V=0 baseline, V=3 weights fetched as contiguous 1 KB per wave-instruction, V=4 + activations as 8 rows x 128 B
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03h
mkdir -p $OUT
cd $R
export AVT_HIP_LIB=$R/audio-video-textures_amd/libavt_hip_exp.so
for round in 1 2; do for V in 0 3 4; do
AVT_CONV_X3_XL_V=$V python tools/probe_x3.py f16x3 83 > $OUT/probe_xlexp_v${V}_$round.log 2>&1
echo "XL_V=$V round $round"; sed -n 2,3p $OUT/probe_xlexp_v${V}_$round.log; grep "xl" $OUT/probe_xlexp_v${V}_$round.log | head -6
done; done
