#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_train_convergence.py -x -q -s 2>&1 | grep -E "CONVERGENCE40|ROUNDTRIP|passed|failed|Error|assert" | cut -c1-1500 ) 2>&1 | tee gpurun_out/r05_conv_test.log
( time timeout 1200 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_train_convergence.py 2>&1 | tail -5 ) 2>&1 | tee gpurun_out/r05_suite_rest.log
