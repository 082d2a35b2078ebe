#!/bin/bash
# per-layer A/B of the 256x256 LDS-DMA tile (AVT_CONV_XL=1 forces it on every cout >= 256, nk >= 4 layer)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for SHAPE in "1024 256 3 1 1 64 8 14 14" "256 256 1 3 3 64 8 14 14" "256 1024 1 1 1 64 8 14 14 res" "2048 512 3 1 1 64 8 7 7" \
             "512 512 1 3 3 64 8 7 7" "512 2048 1 1 1 64 8 7 7 res" "640 256 3 1 1 64 8 28 28" "1280 512 3 1 1 64 8 14 14"; do
  for V in 0 1; do
    echo -n "XL=$V "; AVT_CONV_XL=$V python tools/conv_layer_bench.py $SHAPE 2>&1 | tail -1
  done
done
