#!/bin/bash
# round 5: BatchNorm forward statistics on the conv epilogue (general + XL IO32 tiles): parity, then the config-5 leg
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train_conv.py tests/test_gpu_bn_train.py tests/test_gpu_train_step.py -x -q 2>&1 | tail -15 | tee gpurun_out/r05_epistats_tests.log
python bench.py --mode train --steps 6 --warmup 2 2> gpurun_out/r05_epistats_train.err | tail -c 700 | tee gpurun_out/r05_epistats_train.json
