#!/bin/bash
# isolated timings of the fast-pathway temporal convolutions under different pixel-group caps
cd ${GRAFT_REPO_ROOT:-/root/repo}
for CAP in 0 32 64; do
  echo "== AVT_GROUP_KW1_CAP=$CAP"
  for SHAPE in "64 16 3 1 1 64 32 28 28" "32 8 3 1 1 64 32 56 56" "8 8 3 1 1 64 32 56 56" "128 32 3 1 1 64 32 14 14" "64 32 3 1 1 64 32 28 28" "256 64 3 1 1 64 32 7 7"; do
    AVT_GROUP_KW1_CAP=$CAP python tools/conv_layer_bench.py $SHAPE 2>&1 | tail -1
  done
  AVT_GROUP_KW1_CAP=$CAP python tools/probe_fused.py 64 2>&1 | grep -E "^fused"
done
