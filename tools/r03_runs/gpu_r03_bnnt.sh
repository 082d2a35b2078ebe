#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03v
mkdir -p $OUT
cd $R
for round in 1 2; do
AVT_HIP_LIB=$R/audio-video-textures_amd/libavt_hip_exp.so python bench.py --mode train --steps 3 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('exp lib: NT store of the forward output too', d['value'], d['ms_per_step'])" | tee -a $OUT/train_bn_nt_ab.log
python bench.py --mode train --steps 3 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('shipped: NT loads everywhere, NT stores in the backward', d['value'], d['ms_per_step'])" | tee -a $OUT/train_bn_nt_ab.log
done
timeout 600 python -m pytest tests/test_gpu_bn_train.py -x -q 2>&1 | tail -2
