#!/bin/bash
# On the GPU box: FETCH_SIZE / WRITE_SIZE (separate passes, kernel-trace only) of the training kernels (tools/pmc_train_kernels.py).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_train
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -- python3 $R/tools/pmc_train_kernels.py > $OUT/pmc_$C.log 2>&1
done
cd $R
python3 - $OUT <<'PY'
import csv, glob, collections, json, sys
out_dir = sys.argv[1]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for f in glob.glob("%s/pmc_%s/**/*counter_collection.csv" % (out_dir, c), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == c:
                agg[row["Kernel_Name"][:110]].append(float(row["Counter_Value"]))
    for k, v in agg.items():
        out.setdefault(k, {})[c] = {"launches": len(v), "per_launch_KB": v}
json.dump(out, open(out_dir + "/pmc_train_summary.json", "w"), indent=1)
for k, v in sorted(out.items()):
    if any(s in k for s in ("conv_x3", "wgrad_x3", "bn_")):
        f, w = v.get("FETCH_SIZE", {}).get("per_launch_KB", []), v.get("WRITE_SIZE", {}).get("per_launch_KB", [])
        print(k[:90], "| fetch MB (x2 for gfx950):", [round(2 * a / 1024, 1) for a in f][-8:], "| write MB:", [round(a / 1024, 1) for a in w][-8:])
PY
grep "algorithmic" $OUT/pmc_FETCH_SIZE.log
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
