#!/bin/bash
# round 6, the end-of-round collection on the final code: the whole GPU suite + smoke, the default bench + the rocprofv3 statistics of
# the headline leg alone + PMC passes (tools/gpu_profile_round.sh r06), the config-5 step's kernel statistics (8 items), and the
# roofline recomputed from the profile (tools/roofline_from_rocprof.py)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd)
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) 2>&1 | tee gpurun_out/r06_suite_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r06_smoke.log
bash tools/gpu_profile_round.sh r06 > gpurun_out/r06_profile_round.log 2>&1
python tools/roofline_from_rocprof.py gpurun_out/profile_r06/rocprof_kernel_stats.csv gpurun_out/profile_r06/bench_detail_profiled.json > gpurun_out/profile_r06/roofline_from_rocprof.txt 2>&1
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/profile_r06
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/proft -- python3 $R/bench.py --mode train --steps 1 --warmup 1 > $OUT/bench_train_profiled.json 2> $OUT/proft.err
find $OUT/proft -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/rocprof_train_kernel_stats.csv
rm -rf $OUT/proft
cd $R
tail -c 3000 gpurun_out/profile_r06/bench.json; echo; cat gpurun_out/profile_r06/roofline_from_rocprof.txt | tail -25
