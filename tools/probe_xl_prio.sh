#!/bin/bash
# XL kernel per-layer: default build, then rebuilt ON THE BOX with -DXL_PRIO=1 (s_setprio around the MFMA clusters)
cd ${GRAFT_REPO_ROOT:-/root/repo}
echo "== default build"; bash tools/probe_xl_layers.sh 2>&1 | grep "XL=1"
cd audio-video-textures_amd/csrc && touch conv_igemm.hip && make FLAGS="-O3 -ffp-contract=off -std=c++17 -fPIC --offload-arch=gfx950 -DXL_PRIO=1" -j8 > /dev/null 2>&1; cd ../..
echo "== -DXL_PRIO=1"; bash tools/probe_xl_layers.sh 2>&1 | grep "XL=1"
