#!/bin/bash
# diagnostic: build libavt_hip_stamp.so (-DAVT_CONV_STAMP, a SEPARATE library: the product .so is untouched) on the GPU
# box copy and print where a K-step's cycles go
cd ${GRAFT_REPO_ROOT:-/root/repo}
make -C audio-video-textures_amd/csrc stamp STAMP_EXTRA="${STAMP_EXTRA}" -j8 > /dev/null 2>&1
export AVT_HIP_LIB=$PWD/audio-video-textures_amd/libavt_hip_stamp.so
IFS=";" read -ra ARR <<< "${SHAPES:-256 256 1 3 3 64 8 14 14;1024 256 3 1 1 64 8 14 14;64 256 1 1 1 64 8 56 56 res;64 64 1 3 3 64 8 56 56}"; for SHAPE in "${ARR[@]}"; do
python - $SHAPE <<'PY'
import sys, ctypes, subprocess
sys.path.insert(0, ".")
import torch, avtex
from avtex import _lib
import os
lib = _lib.lib()
stamps = lib.avt_debug_stamps_x3 if os.environ.get("PRECISION") else lib.avt_debug_stamps  # PRECISION=f16x3: the split-plane tile
buf = (ctypes.c_ulonglong * 10)()
sys.argv = ["x"] + sys.argv[1:]
import runpy
stamps(buf, 1)
runpy.run_path("tools/conv_layer_bench.py", run_name="__main__")
torch.cuda.synchronize()
stamps(buf, 1)
n = buf[7]
names = ["prologue", "gload issue | XL: wait own DMA", "compute", "barrier1 | XL: barrier stage complete", "wait+ds_write | XL: DMA issue", "barrier2 | XL: barrier slot free", "epilogue"]
tot = sum(buf[i] for i in range(7))
print("  workgroups %d, cycles per workgroup %.0f; shader clock while resident %.0f MHz (s_memtime / s_memrealtime x 100 MHz)" % (n, tot / max(n, 1), 100.0 * buf[8] / max(buf[9], 1)))
for i, nm in enumerate(names):
    print("    %-40s %6.1f %%  (%.0f cycles/workgroup)" % (nm, 100.0 * buf[i] / tot, buf[i] / max(n, 1)))
PY
done
