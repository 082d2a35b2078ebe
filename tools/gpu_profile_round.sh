#!/bin/bash
# On the GPU box, ONE script for everything profiles/rNN/ quotes about the final code (so the files agree with each other):
#   1. python bench.py                                  -> bench.json (the headline line, both precision legs)
#   2. rocprofv3 --kernel-trace --stats of the same cmd -> rocprof_kernel_stats.csv (per-kernel time; no PMC in this pass)
#   3. rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE    -> pmc_fetch_write_summary.json (separate passes, kernel-trace only,
#      over tools/pmc_kernels.py: every hand-written kernel at the bench shapes; FETCH_SIZE is doubled by the READER
#      (bench.py attach_pmc_traffic) as MI355X_MICROARCH.md prescribes for gfx950)
# usage: bash tools/gpu_profile_round.sh r04   -> gpurun_out/profile_r04/
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
OUT=$R/gpurun_out/profile_$TAG
mkdir -p $OUT
cd $R
python bench.py > $OUT/bench.json 2> $OUT/bench.err
cp $R/bench_detail.json $OUT/bench_detail.json
cd /tmp && export TMPDIR=/tmp
# the profiled command runs NOTHING but the headline leg (no round-3-inputs leg, no precision block: both launch the same kernel symbols),
# so every launch of an encoder symbol belongs to one of its steps + warm-up steps: tools/roofline_from_rocprof.py recomputes the
# line's roofline rows from rocprof_kernel_stats.csv + bench_detail_profiled.json (tests/test_host_logic.py holds them within 5 %)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --steps 2 --warmup 1 --no-fast --no-r03-leg --no-train-leg --no-cpu-baseline --no-precision-block --no-nxn-legs > $OUT/prof_bench.json 2> $OUT/prof.err
cp $R/bench_detail.json $OUT/bench_detail_profiled.json
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/rocprof_kernel_stats.csv
find $OUT/prof -name "*kernel_trace.csv" -delete   # the per-dispatch trace is large; the stats summary is what is kept
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -- python3 $R/tools/pmc_kernels.py > $OUT/pmc_$C.log 2>&1
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
                agg[row["Kernel_Name"][:120]].append(float(row["Counter_Value"]))
    for k, v in agg.items():
        out.setdefault(k, {})[c] = {"launches": len(v), "mean": sum(v) / len(v), "total": sum(v)}
out["_note"] = "KB per launch as rocprofv3 reports them (FETCH_SIZE NOT yet doubled); tools/pmc_kernels.py shapes: N=4096, D=2304, encoder forwards at bench.py's default --enc-batch"
json.dump(out, open(out_dir + "/pmc_fetch_write_summary.json", "w"), indent=1)
for k, v in out.items():
    if not k.startswith("_"):
        pass
PY
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/prof
cut -c1-120 $OUT/rocprof_kernel_stats.csv | head -25
