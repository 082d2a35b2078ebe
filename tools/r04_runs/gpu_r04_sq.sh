# SQ counters of the encoder kernels (separate --pmc passes, kernel-trace only): where do the wave-cycles go
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/sq; mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/p1 -- python3 $R/tools/pmc_x3_forward.py 166 > $O/p1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $O/p2 -- python3 $R/tools/pmc_x3_forward.py 166 > $O/p2.log 2>&1
cd $R
python3 tools/pmc_sq_summary.py $O/p1 > $O/sq_pass1.txt 2>&1; python3 tools/pmc_sq_summary.py $O/p2 > $O/sq_pass2.txt 2>&1
rm -rf $O/p1 $O/p2
cat $O/sq_pass1.txt | cut -c1-330; echo; cat $O/sq_pass2.txt | cut -c1-330
