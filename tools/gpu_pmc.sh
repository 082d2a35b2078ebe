#!/bin/bash
# On the GPU box: HBM traffic counters for the hand-written kernels, one counter per pass (MI355X_MICROARCH.md).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$C -- python3 $R/tools/pmc_kernels.py > $R/gpurun_out/pmc_$C.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob("gpurun_out/pmc_%s/**/*counter_collection.csv" % c, recursive=True)
    agg = collections.defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == c:
                agg[row["Kernel_Name"][:90]].append(float(row["Counter_Value"]))
    for k, v in agg.items():
        out.setdefault(k, {})[c] = {"launches": len(v), "mean": sum(v) / len(v), "total": sum(v)}
json.dump(out, open("gpurun_out/pmc_summary.json", "w"), indent=1)
for k, v in out.items():
    print(k, {c: round(x["mean"], 1) for c, x in v.items()})
PY
