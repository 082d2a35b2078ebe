# end-of-round collection on the final code: smoke(), the whole GPU suite, then tools/gpu_profile_round.sh r04 (bench + rocprofv3 stats +
# PMC passes), the per-layer probes, the bench on round 3's inputs, the training profile
mkdir -p gpurun_out/profile_r04
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/profile_r04/smoke.log 2>&1; tail -1 gpurun_out/profile_r04/smoke.log
python -m pytest tests -q -m gpu > gpurun_out/profile_r04/tests_gpu.log 2>&1
tail -3 gpurun_out/profile_r04/tests_gpu.log
bash tools/gpu_profile_round.sh r04 > gpurun_out/profile_round.log 2>&1
O=gpurun_out/profile_r04
cp bench_detail.json $O/bench_detail.json
python tools/probe_x3.py f16x3 249 table > $O/probe_x3_per_layer_b249.log 2>&1
python tools/probe_x3.py f16x3 166 table > $O/probe_x3_per_layer_b166.log 2>&1
python tools/probe_x3.py f16x3 83 table > $O/probe_x3_per_layer_b83.log 2>&1
python bench.py --inputs r03 --no-train-leg --no-cpu-baseline --no-nxn-legs > $O/bench_inputs_r03.json 2> /dev/null
python bench.py --config 4 --no-train-leg --no-cpu-baseline --no-nxn-legs --no-fast --no-precision-block > $O/bench_config4_one_rank.json 2> /dev/null
bash tools/gpu_profile_train.sh fp32 > /dev/null 2>&1
cp gpurun_out/profile_train_fp32/rocprof_kernel_stats.csv $O/rocprof_train_kernel_stats.csv
tail -1 $O/bench.json | cut -c1-700; tail -1 $O/bench_inputs_r03.json | cut -c1-200; tail -1 $O/bench_config4_one_rank.json | cut -c1-300
