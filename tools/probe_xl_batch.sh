#!/bin/bash
# encoder throughput vs clip batch and XL dispatch rule (tile-count quantisation on 256 CUs)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for CFG in "0 8" "1 8" "1 10" "1 16" "1 30"; do set -- $CFG; for B in 64 80; do
  echo -n "AVT_CONV_XL=$1 XL_NK=$2 batch $B: "; AVT_CONV_XL=$1 AVT_CONV_XL_NK=$2 python tools/probe_fused.py $B 2>&1 | grep -E "^fused"
done; done
