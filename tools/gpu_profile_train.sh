#!/bin/bash
# On the GPU box: rocprofv3 --kernel-trace --stats of one training step at size (bench.py --mode train).
R=${GRAFT_REPO_ROOT:-/root/repo}
DT=${1:-fp32}
OUT=$R/gpurun_out/profile_train_$DT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --mode train --train-dtype $DT --steps 1 --warmup 1 > $OUT/bench.json 2> $OUT/prof.err
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/rocprof_kernel_stats.csv
find $OUT/prof -name "*kernel_trace.csv" -delete
rm -rf $OUT/prof
head -40 $OUT/rocprof_kernel_stats.csv | cut -c1-230
