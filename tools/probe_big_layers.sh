#!/bin/bash
# per-layer A/B of the 256x128 tile (AVT_CONV_BIG=1 forces it on every cout > 64 layer) at the bench batch
cd ${GRAFT_REPO_ROOT:-/root/repo}
for SHAPE in "64 256 1 1 1 64 8 56 56 res" "128 512 1 1 1 64 8 28 28 res" "256 1024 1 1 1 64 8 14 14 res" "512 2048 1 1 1 64 8 7 7 res" \
             "1024 256 3 1 1 64 8 14 14" "256 256 1 3 3 64 8 14 14" "128 128 1 3 3 64 8 28 28" "2048 512 3 1 1 64 8 7 7" \
             "512 512 1 3 3 64 8 7 7" "512 128 1 1 1 64 8 28 28" "320 128 1 1 1 64 8 56 56" "80 256 1 1 1 64 8 56 56" \
             "640 256 3 1 1 64 8 28 28" "1280 512 3 1 1 64 8 14 14"; do
  for BIG in 0 1; do
    echo -n "BIG=$BIG "; AVT_CONV_BIG=$BIG python tools/conv_layer_bench.py $SHAPE 2>&1 | tail -1
  done
done
