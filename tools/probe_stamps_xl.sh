#!/bin/bash
# diagnostic: in-kernel stamps of the x3 XL tile (libavt_hip_stamp.so, never the shipped library)
cd ${GRAFT_REPO_ROOT:-/root/repo}
make -C audio-video-textures_amd/csrc stamp STAMP_EXTRA="${STAMP_EXTRA}" -j8 > /dev/null 2>&1
export AVT_HIP_LIB=$PWD/audio-video-textures_amd/libavt_hip_stamp.so
export PRECISION=f16x3
IFS=";" read -ra ARR <<< "${SHAPES:-1024 256 3 1 1 83 8 14 14;256 256 1 3 3 83 8 14 14;2048 512 3 1 1 83 8 7 7}"; for SHAPE in "${ARR[@]}"; do
python - $SHAPE <<'PY'
import sys, ctypes
sys.path.insert(0, ".")
import torch, avtex
from avtex import _lib
lib = _lib.lib()
stamps = lib.avt_debug_stamps_x3
buf = (ctypes.c_ulonglong * 10)()
sys.argv = ["x"] + sys.argv[1:]
import runpy
stamps(buf, 1)
runpy.run_path("tools/conv_layer_bench.py", run_name="__main__")
torch.cuda.synchronize()
stamps(buf, 1)
n = buf[7]
names = ["prologue", "fragment reads landed (lgkmcnt 0)", "48 MFMAs + 8 DMA pieces", "wait own DMA (vmcnt 0)", "barrier", "-", "epilogue"]
tot = sum(buf[i] for i in range(7))
print("  workgroups %d, cycles per workgroup %.0f; shader clock while resident %.0f MHz" % (n, tot / max(n, 1), 100.0 * buf[8] / max(buf[9], 1)))
for i, nm in enumerate(names):
    print("    %-40s %6.1f %%  (%.0f cycles/workgroup)" % (nm, 100.0 * buf[i] / tot, buf[i] / max(n, 1)))
PY
done
