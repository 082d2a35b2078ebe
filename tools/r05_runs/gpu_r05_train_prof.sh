#!/bin/bash
# rocprofv3 --kernel-trace --stats of the config-5 step (1 warm-up + 1 step), default switches
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/profile_train_${1:-cur}
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --mode train --steps 1 --warmup 1 > $OUT/bench.json 2> $OUT/prof.err
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/rocprof_kernel_stats.csv
rm -rf $OUT/prof
