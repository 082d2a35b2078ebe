#!/bin/bash
# round 6: the config-5 step at ONE item, after the deeper bn_pre_reduce walk: BatchNorm / conv tests, the eager and the replayed step,
# and a per-dispatch kernel trace of the REPLAYED step (tools/experimental/analyze_kernel_trace.py: busy / idle / serial parts)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_timeline
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_bn_train.py tests/test_gpu_train_conv.py tests/test_gpu_train_step.py -x -q -m gpu 2>&1 | tail -6 | tee $O/tests.log
timeout 600 python bench.py --mode train --train-items 1 --steps 12 --warmup 4 > $O/train_one_item_eager.json 2> $O/eager.err
timeout 600 python bench.py --mode train --train-items 1 --steps 12 --warmup 4 --train-graph 1 > $O/train_one_item_graph.json 2> $O/graph.err
timeout 600 python bench.py --mode train --steps 3 --warmup 2 > $O/train_eight_items.json 2> $O/eight.err
tail -c 500 $O/train_one_item_eager.json; echo; tail -c 500 $O/train_one_item_graph.json; echo; tail -c 500 $O/train_eight_items.json; echo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --mode train --train-items 1 --steps 8 --warmup 3 --train-graph 1 > $O/trace_bench.json 2> $O/trace.err
cd $R
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 tools/experimental/analyze_kernel_trace.py $T steps=4 skip_frac=0.6 | tee $O/one_item_graph_timeline.txt
rm -rf $O/trace
