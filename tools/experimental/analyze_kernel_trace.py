#!/usr/bin/env python3
"""Where the wall clock of a replayed (or eager) training step goes, from a rocprofv3 per-dispatch kernel trace.

usage: analyze_kernel_trace.py <kernel_trace.csv> [steps=N] [skip_frac=0.5]

Reads Start_Timestamp / End_Timestamp / Kernel_Name / Queue_Id of every dispatch, keeps the LAST (1 - skip_frac) of the trace (the timed
steps: warm-up and capture come first), and prints
  * the span, the time with >= 1 kernel running (busy), the idle time, and the histogram of concurrently running kernels;
  * per kernel family: launches, summed duration, and the part of it that ran ALONE (nothing else on the device — the serial part of
    the step: what a faster kernel would give back to the wall clock one to one);
  * the idle gaps: count, total, and what ran just before the longest ones.
Nothing here runs on the GPU; the trace comes from `rocprofv3 --kernel-trace --output-format csv`."""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:70]


def main():
    path = sys.argv[1]
    opts = dict(a.split("=") for a in sys.argv[2:])
    skip = float(opts.get("skip_frac", 0.5))
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "0")))
    rows.sort()
    rows = rows[int(len(rows) * skip):]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    span = (t1 - t0) / 1e6
    # sweep: events (+1 at start, -1 at end)
    ev = []
    for i, (s, e, n, q) in enumerate(rows):
        ev.append((s, 1, i))
        ev.append((e, -1, i))
    ev.sort(key=lambda x: (x[0], x[1]))
    running = set()
    hist = collections.Counter()
    alone = collections.Counter()
    last = t0
    gaps = []
    last_ended = None
    for t, d, i in ev:
        dt = t - last
        if dt > 0:
            hist[min(len(running), 4)] += dt
            if len(running) == 1:
                alone[rows[next(iter(running))][2]] += dt
            if not running and last_ended is not None:
                gaps.append((dt, rows[last_ended][2], rows[i][2] if d == 1 else "?"))
        if d == 1:
            running.add(i)
        else:
            running.discard(i)
            last_ended = i
        last = t
    steps = int(opts.get("steps", 0))
    print("%d dispatches over %.2f ms%s; queues: %s" % (len(rows), span, " (%.2f ms per step)" % (span / steps) if steps else "",
                                                      dict(collections.Counter(r[3] for r in rows))))
    tot = sum(hist.values())
    for k in sorted(hist):
        print("  %s kernels running: %7.2f ms  %5.1f %%" % (("%d" % k) if k < 4 else ">=4", hist[k] / 1e6, 100.0 * hist[k] / tot))
    fam = collections.defaultdict(lambda: [0, 0])
    for s, e, n, q in rows:
        fam[n][0] += 1
        fam[n][1] += e - s
    ksum = sum(v[1] for v in fam.values())
    print("kernel time summed %.2f ms (x%.2f of the span)" % (ksum / 1e6, ksum / 1e6 / span))
    print("%-72s %7s %9s %9s" % ("kernel", "calls", "sum ms", "alone ms"))
    for n, (c, d) in sorted(fam.items(), key=lambda kv: -alone.get(kv[0], 0))[:32]:
        print("%-72s %7d %9.2f %9.2f" % (n, c, d / 1e6, alone.get(n, 0) / 1e6))
    gaps.sort(reverse=True)
    print("idle gaps: %d, %.2f ms in all; > 20 us: %d (%.2f ms)" % (len(gaps), sum(g[0] for g in gaps) / 1e6,
                                                                 sum(1 for g in gaps if g[0] > 20000), sum(g[0] for g in gaps if g[0] > 20000) / 1e6))
    after = collections.Counter()
    for g in gaps:
        after[g[1] + "  ->  " + g[2]] += g[0]
    for k, v in after.most_common(12):
        print("  %8.2f ms idle between  %s" % (v / 1e6, k))


if __name__ == "__main__":
    main()
