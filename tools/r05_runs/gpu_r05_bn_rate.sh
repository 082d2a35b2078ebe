#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/bn_rate; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bn_rate -- python3 tools/experimental/probe_bn_bwd_rate.py ${1:-256} ${2:-56} > gpurun_out/bn_rate/run.log 2>&1
tail -3 gpurun_out/bn_rate/run.log
f=$(find gpurun_out/bn_rate -name "*kernel_stats.csv" | head -1); find gpurun_out/bn_rate -name "*kernel_trace.csv" -delete; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r['TotalDurationNs']) > 2e5:
        print("%-90s calls %4s avg %9.1f us min %9.1f max %9.1f" % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
