#!/usr/bin/env python3
"""Instruction mix of a gfx950 kernel's main loop (the backward branch that encloses the most MFMAs) from `hipcc -S --cuda-device-only`
output: MFMA / vector / scalar / LDS / memory instruction counts and the most frequent vector opcodes.  A wave issues one vector
instruction per 4 cycles and a 32x32x16 MFMA occupies the matrix pipe for 32: a loop with 340 vector instructions per 24 MFMAs
(the weight-gradient tile before round 5's trimming) is bound by the vector unit, whatever its LDS or HBM traffic looks like.
(Both arms of a uniform branch inside the loop are counted: read the opcode list for code that runs only on a rare path.)

    hipcc -O3 -ffp-contract=off -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -o /tmp/k.s csrc/wgrad_x3.hip
    python tools/diag/isa_loop_mix.py /tmp/k.s wgrad_x3_xl_kernelILb0ELi128E
"""
import re
import sys
from collections import Counter


def main():
    path, pat = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    i0 = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*%s\S*:" % re.escape(pat), l))
    i1 = next(i for i in range(i0, len(lines)) if "s_endpgm" in lines[i])
    body = lines[i0:i1 + 1]
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    best = None
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l) or re.search(r"s_branch (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            a, b = labels[m.group(1)], i
            n = sum(1 for x in body[a:b] if "v_mfma" in x)
            if best is None or n > best[0] or (n == best[0] and b - a > best[2] - best[1]):
                best = (n, a, b)
    if best is None:
        print("no loop found")
        return
    _, a, b = best
    kinds, valu = Counter(), Counter()
    for x in body[a:b]:
        x = x.strip()
        if not x or x.startswith(";") or x.startswith(".L"):
            continue
        op = x.split()[0]
        if op.startswith("v_mfma"):
            k = "mfma"
        elif op.startswith("v_"):
            k = "valu"
            valu[re.sub(r"_e32|_e64|_sdwa", "", op)] += 1
        elif op.startswith("s_"):
            k = "salu"
        elif op.startswith("ds_"):
            k = "lds"
        elif op.startswith(("buffer_", "global_", "scratch_", "flat_")):
            k = "vmem" + ("(scratch!)" if op.startswith("scratch_") else "")
        else:
            k = "other"
        kinds[k] += 1
    print(body[0].split(":")[0])
    print("  main loop: %d lines  %s" % (b - a, dict(kinds)))
    print("  vector opcodes: " + ", ".join("%s x%d" % kv for kv in valu.most_common(16)))


if __name__ == "__main__":
    main()
