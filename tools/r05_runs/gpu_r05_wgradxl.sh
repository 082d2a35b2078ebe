#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_train_conv.py -x -q -k "weight_gradient or conv_forward_and_gradients" 2>&1 | tail -8 | tee gpurun_out/r05_wgradxl_tests.log
timeout 600 python tools/probe_wgrad_xl.py 120 2>&1 | tail -40 | tee gpurun_out/r05_wgradxl_probe.log
