#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
for rep in 1 2; do for ps in 0 1; do
  python bench.py --mode train --steps 6 --warmup 2 --train-pathway-streams $ps 2> gpurun_out/r05_pathway_$ps.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pathway_streams=$ps', d['value'], d['ms_per_step'], d['loss_first_last'], d['max_memory_allocated_gb'])" | tee -a gpurun_out/r05_pathway_ab.log
done; done
tail -3 gpurun_out/r05_pathway_1.err
