#!/bin/bash
# generic A/B of an environment switch on the fused encoder: probe_ab.sh VAR v1 v2 ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
VAR=$1; shift
for round in 1 2; do for V in "$@"; do
  echo -n "$VAR=$V: "; env $VAR=$V python tools/probe_fused.py 64 2>&1 | grep -E "^fused"
done; done
