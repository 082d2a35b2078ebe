#!/bin/bash
# end-of-round collection: headline bench + rocprof stats + PMC (tools/gpu_profile_round.sh), the training step's line,
# kernel table and per-layer table, per-layer inference probes
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/gpu_profile_round.sh r03 > /dev/null 2>&1
OUT=$R/gpurun_out/profile_r03
cp bench_detail.json $OUT/bench_detail.json 2>/dev/null
python bench.py --mode train --steps 3 --warmup 2 --train-profile > $OUT/train_bench.json 2> $OUT/train_kernels.log
python tools/probe_train_layers.py > $OUT/train_layers.log 2>&1
python tools/probe_x3.py f16x3 83 > $OUT/probe_x3_per_layer_b83.log 2>&1
head -c 1500 $OUT/bench.json; echo
head -c 600 $OUT/train_bench.json; echo
sed -n 1,4p $OUT/probe_x3_per_layer_b83.log
