#!/bin/bash
# L2 / vector-L1 counters of one conv layer (tools/conv_layer_bench.py SHAPE): hit rate and request volume.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
SHAPE=${SHAPE:-"1024 256 3 1 1 128 8 14 14"}
rm -rf $R/gpurun_out/convl2_*
for SET in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_BUSY_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TA_BUSY_sum TA_TA_BUSY_sum TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE"; do
  TAG=$(echo $SET | cut -d' ' -f1)
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $R/gpurun_out/convl2_$TAG -- python3 $R/tools/conv_layer_bench.py $SHAPE > $R/gpurun_out/convl2_$TAG.log 2>&1
  tail -2 $R/gpurun_out/convl2_$TAG.log | cut -c1-200
done
cd $R
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/convl2_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "conv_igemm" in row["Kernel_Name"] or "conv_xl" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(agg.items()):
    print("%-36s mean %.4g (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
