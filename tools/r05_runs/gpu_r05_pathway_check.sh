#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_rccl.py tests/test_gpu_cli_train.py tests/test_gpu_bn_train.py -x -q 2>&1 | tail -4
python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default run: value', round(d['value'],1), 'r03', round(d['value_inputs_r03'],1), 'train', round(d['train_clips_per_s'],1), d['train_ms_per_step'])"
