#!/usr/bin/env python3
"""Compile one .hip file of csrc/ with -Rpass-analysis=kernel-resource-usage and print one line per kernel:
registers, scratch, occupancy, spills.  usage: kernel_resources.py <file.hip> [name filter] [-Dflag ...]"""
import re
import subprocess
import sys

src = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
extra = [a for a in sys.argv[2:] if a.startswith("-")]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-ffp-contract=off", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: (?:\s*)([A-Za-z \[\]/]+): (.+?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
        rows[cur] = {}
    elif cur:
        rows[cur][k] = v
for name, r in rows.items():
    if filt and filt not in name:
        continue
    short = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
    print("%-70s VGPR %3s AGPR %3s scratch %4s occ %s spill v%s s%s" % (short[:70], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize [bytes/lane]"),
          r.get("Occupancy [waves/SIMD]"), r.get("VGPRs Spill"), r.get("SGPRs Spill")))
