#!/bin/bash
# A/B of the LDS-DMA staging per tile shape on the fused encoder (same process order, same box)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for G in 0 1 3 7 5; do
  echo "AVT_CONV_GLDS=$G"; AVT_CONV_GLDS=$G python tools/probe_fused.py 64 2>&1 | grep -E "^fused"
done
