# end-of-round collection: bench + rocprofv3 stats + PMC passes (tools/gpu_profile_round.sh), per-layer probes, the r03-inputs A/B
bash tools/gpu_profile_round.sh r04 > gpurun_out/profile_round.log 2>&1
O=gpurun_out/profile_r04
python tools/probe_x3.py f16x3 166 table > $O/probe_x3_per_layer_b166.log 2>&1
python tools/probe_x3.py f16x3 83 table > $O/probe_x3_per_layer_b83.log 2>&1
python bench.py --inputs r03 --no-train-leg --no-cpu-baseline --no-nxn-legs > $O/bench_inputs_r03.json 2> /dev/null
cp bench_detail.json $O/bench_detail_inputs_r03.json
tail -1 $O/bench.json | cut -c1-600; tail -1 $O/bench_inputs_r03.json | cut -c1-300
