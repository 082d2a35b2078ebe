#!/bin/bash
# round 6: where the fused slow-res2 kernel's time goes — one diagnostic library per phase-skip mask (built on the box), the layer alone
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r06_res2
mkdir -p $O
cd audio-video-textures_amd/csrc
OBJS=$(ls *.o | grep -v res2_x3.o | tr '\n' ' ')
for m in 0 1 2 3 4 8 16 32 48 63; do
  /opt/rocm/bin/hipcc -O3 -ffp-contract=off -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DR2_DBG=$m -c res2_x3.hip -o /tmp/res2_dbg_$m.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libavt_dbg_$m.so $OBJS /tmp/res2_dbg_$m.o
done
cd ../..
for m in 0 1 2 3 4 8 16 32 48 63; do
  echo -n "mask $m: "; AVT_HIP_LIB=/tmp/libavt_dbg_$m.so timeout 120 python tools/probe_res2.py 249 5 2>&1 | grep -v amdgpu.ids | tail -1
done | tee $O/res2_phase_skips.log
