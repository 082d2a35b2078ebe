#!/bin/bash
# round 6 (diagnostic): DistributedDataParallel over gloo, 2 ranks on one GPU, 1 warm-up + 2 steps: where is a rank after 100 s?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_ddp_debug
mkdir -p $O
for f in 1 0; do
AVT_DUMP_STACKS_AFTER=100 timeout 200 python bench.py --mode train --gpus 2 --dist-backend gloo --train-items 2 --steps 2 --warmup 1 --train-fused-sgd $f > $O/ddp_fused$f.json 2> $O/ddp_fused$f.err
echo "fused=$f rc=$?"; tail -c 400 $O/ddp_fused$f.json; echo
grep -n "File \"/root/repo\|Thread\|most recent" $O/ddp_fused$f.err | head -40
done
