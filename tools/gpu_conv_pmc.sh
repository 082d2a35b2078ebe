#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
SHAPE=${SHAPE:-"256 256 1 3 3 64 8 14 14"}
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SALU SQ_LDS_UNALIGNED_STALL SQ_VALU_MFMA_COEXEC_CYCLES"; do
  TAG=$(echo $SET | cut -d' ' -f1)
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $R/gpurun_out/convpmc_$TAG -- python3 $R/tools/conv_layer_bench.py $SHAPE > $R/gpurun_out/convpmc_$TAG.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/convpmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if any(k in row["Kernel_Name"] for k in ("conv_igemm", "conv_xl", "conv_xb", "stem_kernel", "conv_x3", "pw_x3")):
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(agg.items()):
    print("%-28s mean %.4g (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
