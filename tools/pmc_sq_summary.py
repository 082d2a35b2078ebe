#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc counter_collection CSVs per kernel symbol: mean counter value per launch and the SQ wave-cycle shares.
usage: pmc_sq_summary.py <dir> [<dir> ...]"""
import collections
import csv
import glob
import re
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            name = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"]).split("(")[0].replace("void ", "")
            agg[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
rows = []
for k, cs in agg.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    rows.append((m.get("SQ_WAVE_CYCLES", 0.0) * len(next(iter(cs.values()))), k, m, len(next(iter(cs.values())))))
for _, k, m, n in sorted(rows, reverse=True)[:24]:
    wc = m.get("SQ_WAVE_CYCLES") or 1.0
    parts = ["%s %.3f" % (c.replace("SQ_", ""), v / wc) for c, v in sorted(m.items()) if c != "SQ_WAVE_CYCLES" and c.startswith("SQ_")]
    other = ["%s %.3g" % (c, v) for c, v in sorted(m.items()) if not c.startswith("SQ_")]
    print("%-58s x%-3d wave-cycles %.3g | as a share of them: %s %s" % (k[:58], n, wc, ", ".join(parts), " | " + ", ".join(other) if other else ""))
