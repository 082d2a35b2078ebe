#!/bin/bash
# round 5: BatchNorm forward statistics on the conv epilogue (general + XL IO32 tiles + the streaming pointwise kernel): parity, then the
# config-5 leg with and without, alternating on one box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
if [ "$1" != "ab" ]; then
timeout 900 python -m pytest tests/test_gpu_train_conv.py tests/test_gpu_bn_train.py tests/test_gpu_train_step.py -x -q 2>&1 | tail -15 | tee gpurun_out/r05_epistats_tests.log
fi
for rep in 1 2; do for epi in 0 1; do
  python bench.py --mode train --steps 6 --warmup 2 --train-epi-stats $epi 2> gpurun_out/r05_epistats_train_$epi.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('epi_stats=$epi', d['value'], d['ms_per_step'], d['loss_first_last'])" | tee -a gpurun_out/r05_epistats_ab.log
done; done
