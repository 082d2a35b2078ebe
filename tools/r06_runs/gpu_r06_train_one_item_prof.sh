#!/bin/bash
# round 6: rocprofv3 kernel statistics of the config-5 step at ONE item (16 clips) per step, and at 8 items for the same launches' times
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_train_prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -- python3 $R/bench.py --mode train --train-items 1 --steps 10 --warmup 3 > $O/bench_one_item.json 2> $O/prof1.err
find $O/prof1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/rocprof_train_one_item_kernel_stats.csv
rm -rf $O/prof1
cd $R
head -30 $O/rocprof_train_one_item_kernel_stats.csv | cut -c1-200
