#!/bin/bash
# per-layer A/B of the 64x64-per-wave XL forms (AVT_CONV_XLS bit 0 = <512,64>, bit 1 = <256,128>)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for SHAPE in "64 64 1 3 3 80 8 56 56" "128 128 1 3 3 80 8 28 28" "512 128 1 1 1 80 8 28 28" "256 64 1 1 1 80 8 56 56"; do
  for V in 0 3; do
    echo -n "XLS=$V "; AVT_CONV_XLS=$V python tools/conv_layer_bench.py $SHAPE 2>&1 | tail -1
  done
done
