#!/bin/bash
# round 6: build-time variants of the fused slow-res2 kernel (fragment read-ahead depth, scheduler fences), the layer alone
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r06_res2
mkdir -p $O
cd audio-video-textures_amd/csrc
OBJS=$(ls *.o | grep -v res2_x3.o | tr '\n' ' ')
i=0
for v in "$@"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc -O3 -ffp-contract=off -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $v -c res2_x3.hip -o /tmp/res2_var_$i.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libavt_var_$i.so $OBJS /tmp/res2_var_$i.o
done
cd ../..
i=0
for v in "$@"; do
  i=$((i+1))
  echo "[$v]: "; AVT_HIP_LIB=/tmp/libavt_var_$i.so timeout 120 python tools/probe_res2.py 249 5 2>&1 | grep -v amdgpu.ids | tail -10
done | tee $O/res2_variants.log
