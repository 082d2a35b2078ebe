#!/bin/bash
# round 6: the one-item config-5 step's two chains (query encoder on ONE clip, target encoder on 15), each as a replayed graph of its own
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_chains
mkdir -p $O
timeout 900 python tools/experimental/probe_one_item_chains.py 20 2>&1 | grep -v amdgpu.ids | tee $O/one_item_chains.log
timeout 600 python tools/probe_train_layers.py 1 2>&1 | grep -v amdgpu.ids > $O/train_layers_one_item.log
head -12 $O/train_layers_one_item.log
