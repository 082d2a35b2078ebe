#!/bin/bash
# round 5: BatchNorm backward statistics on the input-gradient epilogue (128-row IO32 tiles): parity, then the config-5 leg A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
if [ "$1" != "ab" ]; then
timeout 900 python -m pytest tests/test_gpu_train_conv.py tests/test_gpu_bn_train.py tests/test_gpu_train_step.py -x -q 2>&1 | tail -15 | tee gpurun_out/r05_epibwd_tests.log
fi
for rep in 1 2; do for epi in 0 1; do
  python bench.py --mode train --steps 6 --warmup 2 --train-epi-bwd $epi 2> gpurun_out/r05_epibwd_train_$epi.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('epi_bwd=$epi', d['value'], d['ms_per_step'], d['loss_first_last'], d['max_memory_allocated_gb'])" | tee -a gpurun_out/r05_epibwd_ab.log
done; done
