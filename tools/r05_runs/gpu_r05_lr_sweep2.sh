#!/bin/bash
# round 5, call 2: which synthetic task / init / learning rate lets the InfoNCE loss fall within a few hundred steps
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
run() { tag=$1; shift; timeout 600 python tools/train_convergence.py --steps 150 --modes x3 --out gpurun_out/r05_sweep2_$tag.json "$@" 2> gpurun_out/r05_sweep2_$tag.err | tail -c 300; }
run A --scene-len 24 --lr 0.03 --init default
run B --scene-len 8 --lr 0.03 --init bench
run C --scene-len 8 --lr 0.03 --init default
run D --scene-len 8 --lr 0.1 --init bench
