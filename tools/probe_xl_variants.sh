#!/bin/bash
# XL kernel: default build vs rebuilt ON THE BOX with extra -D flags ($1), per-layer + encoder
cd ${GRAFT_REPO_ROOT:-/root/repo}
echo "== default build"; AVT_CONV_XL_NK=8 bash tools/probe_xl_layers.sh 2>&1 | grep "XL=1"; python tools/probe_fused.py 80 | grep ^fused
cd audio-video-textures_amd/csrc && touch conv_igemm.hip && make FLAGS="-O3 -ffp-contract=off -std=c++17 -fPIC --offload-arch=gfx950 $1" -j8 > /dev/null 2>&1; cd ../..
echo "== $1"; AVT_CONV_XL_NK=8 bash tools/probe_xl_layers.sh 2>&1 | grep "XL=1"; python tools/probe_fused.py 80 | grep ^fused
