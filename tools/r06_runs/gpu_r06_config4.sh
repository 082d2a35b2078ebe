#!/bin/bash
# round 6: BASELINE config 4's per-rank shape on one GPU (2048 windows per rank, top-8, th 0.0 + 0.3) on the final code, f32 and bf16x3 similarity;
# and the one-item step's chains as graphs of their own, for the record
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_config4
mkdir -p $O
timeout 900 python bench.py --config 4 --no-train-leg --no-cpu-baseline > $O/bench_config4_one_rank.json 2> $O/err.log
timeout 900 python bench.py --config 4 --sim-precision bf16x3 --no-train-leg --no-cpu-baseline --no-fast --no-r03-leg > $O/bench_config4_one_rank_sim_bf16x3.json 2>> $O/err.log
for f in bench_config4_one_rank bench_config4_one_rank_sim_bf16x3; do python3 -c "
import json
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', d.get('value'), d.get('ms_per_step'), d.get('nxn_build_ms'), d.get('topk_ms'), d.get('frames_lists_identical'), d.get('th0_ties'))"; done
timeout 600 python tools/experimental/probe_one_item_chains.py 20 2>&1 | grep -v "amdgpu.ids\|Warning\|warn\|return Variable" | tee $O/one_item_chains_final.log
