#!/usr/bin/env python3
"""Summary of a gfx950 kernel's instruction stream from `hipcc -S --cuda-device-only` output: per run of instructions between
waits / barriers / branches / labels, the number of MFMAs, global / buffer loads, LDS reads and writes.  What it is for: a
software-pipelined loop whose s_waitcnt vmcnt(N) has a SMALLER N than the prefetch distance intends (a load issued under a
condition makes the compiler count for the path that skipped it) — found in the 256 x 128 weight-gradient tile in round 5.

    hipcc -O3 -ffp-contract=off -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -o /tmp/k.s csrc/wgrad_x3.hip
    python tools/diag/isa_loop_summary.py /tmp/k.s wgrad_x3_xl_kernelILb0 [--from-label .LBB4_53]
"""
import re
import sys


def main():
    path, pat = sys.argv[1], sys.argv[2]
    start_label = sys.argv[4] if len(sys.argv) > 4 and sys.argv[3] == "--from-label" else None
    lines = open(path).read().split("\n")
    i0 = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*%s\S*:" % re.escape(pat), l))
    i1 = next(i for i in range(i0, len(lines)) if "s_endpgm" in lines[i])
    body = lines[i0:i1 + 1]
    print(lines[i0].split(":")[0], "(%d lines)" % len(body))
    counts = {}
    on = start_label is None

    def flush():
        if counts:
            print("      " + "  ".join("%s x%d" % kv for kv in counts.items()))
            counts.clear()

    for n, l in enumerate(body):
        t = l.strip()
        if not t or t.startswith(";"):
            continue
        if start_label and t.startswith(start_label + ":"):
            on = True
        if not on:
            continue
        op = t.split()[0]
        key = None
        if op.startswith("v_mfma"):
            key = "mfma"
        elif op.startswith("global_load") or op.startswith("buffer_load"):
            key = "vmload" + ("_lds" if " lds" in t else "")
        elif op.startswith("global_store") or op.startswith("buffer_store") or op.startswith("global_atomic"):
            key = "vmstore"
        elif op.startswith("ds_read") or op.startswith("ds_load"):
            key = "ds_read"
        elif op.startswith("ds_write") or op.startswith("ds_store"):
            key = "ds_write"
        elif op.startswith("scratch_"):
            key = "scratch"
        if key:
            counts[key] = counts.get(key, 0) + 1
            continue
        if op in ("s_waitcnt", "s_barrier") or op.startswith("s_cbranch") or op == "s_branch" or t.startswith(".LBB"):
            flush()
            print("%5d  %s" % (n, t.split(";")[0].strip()))
    flush()


if __name__ == "__main__":
    main()
