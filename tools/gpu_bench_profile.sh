#!/bin/bash
# Runs on the GPU box: plain bench, then the same command under rocprofv3 --kernel-trace --stats.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01}
shift
cd $R
python bench.py --steps 2 --warmup 1 "$@" > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
tail -c 3000 gpurun_out/bench_$TAG.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/prof_$TAG.json 2> $R/gpurun_out/prof_$TAG.err
cd $R
find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/prof_${TAG}_kernel_stats.csv
# keep only the small summary (the per-dispatch trace is large)
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -size +20M -delete
ls -la gpurun_out/prof_$TAG/* | head
head -30 gpurun_out/prof_${TAG}_kernel_stats.csv
