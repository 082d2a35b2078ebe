#!/bin/bash
# encoder throughput vs clip batch with and without the XL tile (tile-count quantisation on 256 CUs)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for XL in 0 1 150; do for B in 64 80 96; do
  echo -n "AVT_CONV_XL=$XL batch $B: "; AVT_CONV_XL=$XL python tools/probe_fused.py $B 2>&1 | grep -E "^fused"
done; done
